"""Drop-in without editing the reference: run its entry points as

    PYTHONPATH=/path/to/this/repo/shim python train.py ...        (or test.py)

Python imports `sitecustomize` at start-up; this one registers the MI355X implementations under the module names the
reference imports (config.eventad_config, models.EventAD, dagr.*, ev_graph_cuda, asy_tools, and the sync-free
utils.test / utils.evaluation), so they win over the same-named packages next to train.py.  Set EVENTAD_SHIM_EVAL=0 to
keep the reference's own utils.test / utils.evaluation.  If the interpreter already has a sitecustomize further down
sys.path (some distributions ship one), it is run afterwards."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_root = os.path.dirname(_here)
if _root not in sys.path:
    sys.path.insert(0, _root)

import eventad_amd.compat as _compat  # noqa: E402

_compat.install(eval_loop=os.environ.get("EVENTAD_SHIM_EVAL", "1") == "1")

for _p in sys.path:                                         # chain to a sitecustomize this one shadows
    _f = os.path.join(_p, "sitecustomize.py") if _p else None
    if _f and os.path.isfile(_f) and os.path.abspath(_p) != _here:
        _spec = importlib.util.spec_from_file_location("_shadowed_sitecustomize", _f)
        try:
            _spec.loader.exec_module(importlib.util.module_from_spec(_spec))
        except Exception:
            pass
        break
