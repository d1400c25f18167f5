"""Hand-derived known answers for the graph-builder oracle.
The reference has no tests for src/dagr/graph; these pin oracle/graph_ref.c."""
import torch

from oracle import graph as G


def _edges(xs, ys, ts, bs, W=8, H=8, B=1, r=1, dt=100, K=16, Q=128):
    g = G.SlidingWindowGraphRef(W, H, B, K, Q, r, dt)
    pos = torch.tensor([xs, ys, ts], dtype=torch.int32).t().contiguous()
    return g.forward(torch.tensor(bs, dtype=torch.int32), pos)


def test_spiral_order_r1_r2():
    # spiral.h: (0,0),(1,0),(1,1),(0,1),(-1,1),(-1,0),(-1,-1),(0,-1),(1,-1),(2,-1)...
    xy = G.spiral(25).tolist()
    assert xy[:10] == [[0, 0], [1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, -1], [1, -1], [2, -1]]
    assert sorted(map(tuple, xy[:9])) == sorted((x, y) for x in (-1, 0, 1) for y in (-1, 0, 1))
    assert sorted(map(tuple, xy)) == sorted((x, y) for x in range(-2, 3) for y in range(-2, 3))


def test_self_edge_first_and_older_to_newer():
    e = _edges([3, 4, 3], [3, 3, 3], [0, 10, 20], [0, 0, 0])
    # event0: self only; event1: self, then 0 (pixel (-1,0)); event2: self, same pixel 0 first
    # (spiral step 0 = own pixel), then event1 at (+1,0).
    assert e.t().tolist() == [[0, 0], [1, 1], [0, 1], [2, 2], [0, 2], [1, 2]]


def test_dt_boundary_kept_and_dropped():
    e = _edges([2, 2, 2], [2, 2, 2], [0, 1, 101], [0, 0, 0], dt=100)
    # event2: dt to event1 = 100 (kept, `dt > delta` is strict), dt to event0 = 101 (dropped)
    assert e.t().tolist() == [[0, 0], [1, 1], [0, 1], [2, 2], [1, 2]]


def test_newest_first_within_pixel_and_cap_at_16():
    n = 20
    e = _edges([1] * n, [1] * n, list(range(n)), [0] * n)
    last = e[:, e[1] == n - 1]
    assert last.shape[1] == 16                              # 17th.. neighbours dropped
    assert last[0].tolist() == [19] + list(range(18, 3, -1))  # self, then newest-first


def test_queue_keeps_only_q_newest():
    n, Q = 12, 4
    e = _edges([1] * n, [1] * n, list(range(n)), [0] * n, Q=Q)
    # all events inserted before the search: FIFO of pixel holds ids 11,10,9,8 only.
    assert e[:, e[1] == 11][0].tolist() == [11, 10, 9, 8]
    assert e[:, e[1] == 9][0].tolist() == [9, 8]          # ids 11,10 skipped (newer), 9 is self
    assert e[:, e[1] == 5][0].tolist() == [5]             # older ids fell out of the FIFO


def test_fov_border_and_radius():
    # corner pixel; neighbour two pixels away is outside r=1, inside r=2
    a = _edges([0, 2], [0, 0], [0, 1], [0, 0], r=1)
    b = _edges([0, 2], [0, 0], [0, 1], [0, 0], r=2)
    assert a.t().tolist() == [[0, 0], [1, 1]]
    assert b.t().tolist() == [[0, 0], [1, 1], [0, 1]]


def test_clips_have_independent_queue_planes():
    e = _edges([3, 3, 3, 3], [3, 3, 3, 3], [0, 1, 0, 1], [0, 0, 1, 1], B=2)
    assert e.t().tolist() == [[0, 0], [1, 1], [0, 1], [2, 2], [3, 3], [2, 3]]


def test_spiral_visit_order_decides_neighbours():
    # three older events at (+1,0), (0,+1), (-1,0) of the last one: spiral order is
    # (1,0) -> (0,1) -> (-1,0)
    e = _edges([5, 4, 3, 4], [4, 5, 4, 4], [0, 1, 2, 3], [0, 0, 0, 0])
    assert e[:, e[1] == 3][0].tolist() == [3, 0, 1, 2]


def test_single_event_path():
    e = _edges([3], [2], [7], [0])
    assert e.t().tolist() == [[0, 0]]


def test_denormalize_and_params():
    pos = torch.tensor([[5 / 256, 7 / 256, 999999 / 1e6]], dtype=torch.float32)
    d = G.denormalize_pos(pos, 256, 256, 1000000)
    assert d[0, :2].tolist() == [5, 7] and d.dtype == torch.int32
    assert G.graph_params(0.01, 256, 1000000) == (3, 10000)
    assert G.graph_params(0.01, 1280, 1000000) == (13, 10000)
    assert G.graph_params(0.01, 64, 1000000) == (1, 10000)
