"""`dagr.asynchronous` (reference src/dagr/asynchronous/): event-by-event inference.

The reference converts a model layer by layer (make_model_asynchronous, __init__.py:41-110): every converted layer keeps
the graph it last saw and, when called with a graph that has more nodes, computes only what the new events change
(conv.py:96-190, max_pool.py:245-252), over the incremental graph builder (graph/ev_graph.py:45-103).  Here the whole
backbone is native, so the conversion switches the backbone as a whole: its first call (or one with reset=True) is a dense
pass that reserves room for later events; a call whose batch holds the events already seen plus new ones behind them
appends the new events at O(k) cost at the event level -- their ELL rows, their layer-1 rows, the pool1 voxels they touch
(csrc/incr.hip) -- and re-runs the pooled stage (at most 2240 nodes per clip) on the updated level-0 graph.  The CNN
feature maps are those of the dense pass (one frame per clip window, as in the reference's asynchronous evaluation).

Also here: the native index-masked row kernels of `asy_tools` (asy_tools.py)."""


def _backbone_of(module):
    for path in ((), ("backbone",), ("dagr_model", "backbone"), ("ema", "backbone")):
        m = module
        try:
            for name in path:
                m = getattr(m, name)
        except AttributeError:
            continue
        if hasattr(m, "async_forward") and hasattr(m, "sync_forward"):
            return m
    raise TypeError("make_model_asynchronous: expected a dagr Net, a DAGR model or an EventADModel (got %s)" % type(module).__name__)


def make_model_asynchronous(module, log_flops=False, capacity=65536):
    """Switch the model's graph backbone to event-by-event processing (reference __init__.py:41-110).  capacity: events that can
    be appended behind a dense pass before the next dense pass is needed.  log_flops is accepted for signature compatibility;
    FLOP accounting (reference asynchronous/flops/) is not implemented.  Returns `module`."""
    net = _backbone_of(module)
    net.enable_asynchronous(int(capacity))
    return module


def make_model_synchronous(module):
    """Back to one dense pass per call (reference __init__.py:30-39)."""
    _backbone_of(module).disable_asynchronous()
    return module


__all__ = ["make_model_asynchronous", "make_model_synchronous"]
