"""Test/validation loop of EventAD (reference utils/test.py, utils/evaluation.py, utils/utils.py:70-93) on top of
the MI355X path: predictions leave the device once per batch instead of once per object."""
