"""eventad_amd -- MI355X-native implementation of EventAD's data-parallel hot path."""
__version__ = "0.1.0"
