"""oracle/graph.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's graph-builder host logic around the native
kernels (paths relative to /root/reference):

  insert_events_into_queue   src/dagr/graph/utils.py:6-18
  search_for_edges           src/dagr/graph/utils.py:20-23
  SlidingWindowGraphRef      src/dagr/graph/ev_graph.py:18-103,106-166 (forward with
                             delete_nodes=False, as EV_TGN calls it, ev_tgn.py:56)
  denormalize_pos            src/dagr/model/layers/ev_tgn.py:11-16
  graph_params               src/dagr/model/layers/ev_tgn.py:27-37

The kernels themselves are in oracle/graph_ref.c (loaded with ctypes).
PARITY UNPINNED w.r.t. reference outputs (no reference fixtures; the CUDA extension cannot be built here): pinned by
the hand-derived known answers of tests/test_oracle_graph.py only.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgraph_ref.so")
_lib = None


def build():
    """Compile oracle/graph_ref.c with gcc (idempotent)."""
    src = os.path.join(_HERE, "graph_ref.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def spiral(n):
    xy = torch.zeros(n, 2, dtype=torch.int32)
    lib().ora_spiral(ctypes.c_int(n), _p(xy))
    return xy


def insert_events_into_queue(batch, pos, indices, queue):
    """graph/utils.py:6-18.  batch i32[N], pos i32[N,3], indices i32[N], queue i32[B,Q,H,W]."""
    B, Q, H, W = queue.shape
    if len(batch) > 1:
        lin = pos[:, 0] + W * pos[:, 1] + W * H * batch
        sorted_lin, sort_index = torch.sort(lin, stable=True, descending=False)
        sorted_indices = indices[sort_index].int().contiguous()
        uniq, counts = torch.unique_consecutive(sorted_lin, return_counts=True)
        uniq = uniq.int().contiguous()
        cumsum = torch.cumsum(counts, dim=0).int().contiguous()
        lib().ora_insert_in_queue(_p(sorted_indices), _p(uniq), _p(cumsum), _p(queue),
                                  B, Q, H, W, ctypes.c_int(len(uniq)))
    else:
        ev = pos.reshape(-1).int().contiguous()
        lib().ora_insert_in_queue_single(_p(indices.int().contiguous()), _p(ev), _p(queue), B, Q, H, W)
    return queue


def search_for_edges(batch, pos, all_timestamps, queue, indices, max_num_neighbors, radius,
                     delta_t_us, edges, min_index):
    """graph/utils.py:20-23 (kernel call + drop of the -1 columns)."""
    B, Q, H, W = queue.shape
    N = len(batch)
    K = edges.shape[1]
    lib().ora_fill_edges(_p(batch), _p(pos), _p(all_timestamps), _p(indices), _p(queue), _p(edges),
                         B, Q, H, W, N, ctypes.c_int64(K), ctypes.c_float(radius),
                         ctypes.c_float(delta_t_us), max_num_neighbors, min_index)
    return edges[:, edges[1] >= 0]


class SlidingWindowGraphRef:
    """ev_graph.py:18-103 + 138-166 restated for the way EV_TGN drives it
    (one forward per batch, delete_nodes=False, reset() between batches)."""

    def __init__(self, width, height, batch_size, max_num_neighbors=16, max_queue_size=128,
                 radius=7, delta_t_us=600000):
        self.W, self.H, self.B = width, height, batch_size
        self.K, self.Q = max_num_neighbors, max_queue_size
        self.radius, self.delta_t_us = radius, delta_t_us
        self.event_queue = None
        self.reset()

    def reset(self):                                   # ev_graph.py:52-60
        self.all_timestamps = torch.zeros((0,), dtype=torch.int32)
        self.max_index = 0
        self.min_index = 0
        if self.event_queue is not None:
            self.event_queue.fill_(-1)

    def forward(self, batch, pos):                     # ev_graph.py:62-103
        n_ev = len(batch)
        if self.event_queue is None:                   # initialize(), ev_graph.py:45-50
            self.event_queue = torch.full((self.B, self.Q, self.H, self.W), -1, dtype=torch.int32)
        if n_ev == 0:
            return torch.zeros((2, 0), dtype=torch.int64)
        assert batch.dtype == torch.int32 and pos.dtype == torch.int32
        batch = batch.contiguous()
        pos = pos.contiguous()
        self.all_timestamps = torch.cat([self.all_timestamps, pos[:, 2]]).contiguous()
        indices = (self.max_index + torch.arange(n_ev, dtype=torch.int32)).contiguous()
        self.max_index += n_ev
        insert_events_into_queue(batch, pos, indices, self.event_queue)
        edge_buffer = torch.full((2, self.K * n_ev), -1, dtype=torch.int64)
        return search_for_edges(batch, pos, self.all_timestamps, self.event_queue, indices,
                                self.K, float(self.radius), float(self.delta_t_us), edge_buffer,
                                self.min_index)


def denormalize_pos(pos, width, height, time_window):
    """ev_tgn.py:11-16: (int64[W,H,T] * f32 pos + 1e-3).int() -- two fp32 roundings, then trunc."""
    denorm = torch.tensor([int(width), int(height), int(time_window)])
    return (denorm.view(1, -1) * pos + 1e-3).int()


def graph_params(radius, width, time_window):
    """ev_tgn.py:27-29: integer pixel radius and time window of the radius graph."""
    delta_t_us = int(radius * time_window)
    r = int(radius * width + 1)
    return r, delta_t_us


def build_edges(pos_norm, batch, width, height, time_window, num_graphs, radius=0.01,
                max_neighbors=16, max_queue_size=128):
    """EV_TGN.forward (ev_tgn.py:39-58) for one fresh batch -> edge_index i64[2,E]."""
    r, dt = graph_params(radius, width, time_window)
    g = SlidingWindowGraphRef(width, height, num_graphs, max_neighbors, max_queue_size, r, dt)
    pos_i = denormalize_pos(pos_norm, width, height, time_window)
    return g.forward(batch.int(), pos_i).long()
