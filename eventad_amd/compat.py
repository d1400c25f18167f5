"""Drop-in aliases: after `eventad_amd.compat.install()` the reference's import statements resolve
to this package --
    from config.eventad_config import parse_eventad_args            (train.py / test.py)
    from models.EventAD import EventADModel                         (utils/model.py:3)
    from dagr.model.networks.dagr import DAGR                       (utils/model.py:4)
    from dagr.model.networks.ema import ModelEMA                    (utils/model.py:5)
    from dagr.utils.buffers import format_data                      (utils/train.py)
    import ev_graph_cuda                                            (src/dagr/graph/utils.py:2)
    import asy_tools                                                (src/dagr/asynchronous/*.py)
    from dagr.asynchronous import make_model_asynchronous           (src/dagr/asynchronous/__init__.py)
so the reference's train.py / test.py / utils run unchanged on top of the HIP path."""
import importlib
import sys
import types

import torch

from ._lib import call, ptr

_ALIASES = {
    "config": "eventad_amd.config",
    "config.eventad_config": "eventad_amd.config.eventad_config",
    "models": "eventad_amd.models",
    "models.EventAD": "eventad_amd.models.EventAD",
    "dagr": "eventad_amd.dagr",
    "dagr.utils": "eventad_amd.dagr.utils",
    "dagr.utils.buffers": "eventad_amd.dagr.utils.buffers",
    "dagr.model": "eventad_amd.dagr.model",
    "dagr.model.networks": "eventad_amd.dagr.model.networks",
    "dagr.model.networks.dagr": "eventad_amd.dagr.model.networks.dagr",
    "dagr.model.networks.net": "eventad_amd.dagr.model.networks.net",
    "dagr.model.networks.net_img": "eventad_amd.dagr.model.networks.net_img",
    "dagr.model.networks.ema": "eventad_amd.dagr.model.networks.ema",
    "dagr.graph": "eventad_amd.dagr.graph",
    "dagr.graph.ev_graph": "eventad_amd.dagr.graph.ev_graph",
    "dagr.asynchronous": "eventad_amd.dagr.asynchronous",          # make_model_asynchronous / make_model_synchronous
    "asy_tools": "eventad_amd.dagr.asynchronous.asy_tools",        # src/dagr/asynchronous/asy_tools (CUDA extension)
}


def _check(t, name):
    # same failure mode as the reference's AT_ASSERTM checks (ev_graph.cu:9-12): RuntimeError
    if not t.is_cuda:
        raise RuntimeError(name + " must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(name + " must be contiguous")


def _ev_graph_module():
    """`ev_graph_cuda` with the reference's three pybind signatures (ev_graph.cu:279-283) over the C ABI."""
    m = types.ModuleType("ev_graph_cuda")
    stream = lambda: torch.cuda.current_stream().cuda_stream

    def fill_edges_cuda(batch, pos, all_timestamps, event_queue, indices, max_num_neighbors, radius, delta_t_us,
                        edges, min_index):
        for t, n in ((batch, "batch"), (pos, "pos"), (event_queue, "event_queue"), (all_timestamps, "all_timestamps"),
                     (edges, "edges"), (indices, "indices")):
            _check(t, n)
        B, Q, H, W = event_queue.shape
        call("eventad_fill_edges", ptr(batch), ptr(pos), ptr(all_timestamps), ptr(event_queue), ptr(indices),
             int(max_num_neighbors), float(radius), float(delta_t_us), ptr(edges), int(min_index), B, Q, H, W,
             batch.shape[0], edges.shape[1], stream())

    def insert_in_queue_cuda(indices, unique_coords, cumsum_counts, queue):
        for t, n in ((indices, "indices"), (unique_coords, "unique_coords"), (cumsum_counts, "cumsum_counts"),
                     (queue, "queue")):
            _check(t, n)
        B, Q, H, W = queue.shape
        coords = unique_coords.int()                         # held until after the launch (may be a fresh tensor)
        call("eventad_insert_in_queue", ptr(indices), ptr(coords), ptr(cumsum_counts), ptr(queue),
             B, Q, H, W, unique_coords.shape[0], stream())
        return queue

    def insert_in_queue_single_cuda(indices, events, queue):
        B, Q, H, W = queue.shape
        ids, ev = indices.int().contiguous(), events.int().contiguous()
        call("eventad_insert_in_queue_single", ptr(ids), ptr(ev), ptr(queue), B, Q, H, W, stream())
        return queue

    m.fill_edges_cuda, m.insert_in_queue_cuda, m.insert_in_queue_single_cuda = \
        fill_edges_cuda, insert_in_queue_cuda, insert_in_queue_single_cuda
    return m


# the test/validation loop: only these modules of the reference's `utils` package are replaced (test, evaluation, fps) --
# the package itself (data loaders, result files, ...) stays the reference's, so they are registered as submodules
_EVAL_ALIASES = {"utils.test": "eventad_amd.utils.test", "utils.evaluation": "eventad_amd.utils.evaluation",
                 "utils.fps": "eventad_amd.utils.fps"}


def install(eval_loop=False):
    """eval_loop=True additionally routes `utils.test.collect_predictions` and `utils.evaluation.*` (test.py:8-9) to
    the sync-free versions in eventad_amd/utils/."""
    for alias, target in _ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)
    sys.modules["ev_graph_cuda"] = _ev_graph_module()
    if eval_loop:
        for alias, target in _EVAL_ALIASES.items():
            sys.modules[alias] = importlib.import_module(target)
