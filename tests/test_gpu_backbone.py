"""GPU parity: native HIP backbone (through the C ABI) vs the CPU oracle on the same seeded input."""
import pytest
import torch

from eventad_amd import synthetic, weights
from eventad_amd._lib import call, ptr

pytestmark = pytest.mark.gpu


def _setup(W, H, n, B, t_span_us=50000, mode="gather", ell_variant=None):
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    data = synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True, t_span_us=t_span_us)
    data = dagr_ref.format_data(data)
    sd = weights.random_graph_state()
    cfg = dagr_ref.make_config(W, H, B)
    (o3, o4), stages = dagr_ref.backbone_forward(sd, data, cfg, data.image_feats, return_all=True, mode=mode)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(sd, W, H, dev)
    if ell_variant is not None:
        nb.ell_variant = ell_variant
    res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    assert nb.check_flags() & ~1 == 0
    return data, cfg, (o3, o4), stages, nb, res


def _level(L):
    n = int(L.n_dev.item())
    e = int(L.e_dev.item())
    rowptr = L.rowptr[: n + 1].cpu().long()
    src = L.csr_src[:e].cpu().long()
    dst = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
    return n, L.x[:n].cpu(), L.npos[:n].cpu(), L.nbatch[:n].cpu().long(), torch.stack([src, dst])


# the third case is dense (72 events per 3x3 search window: the spiral-walk edge search), the others sparse (row scan)
# ... and 12 clips make the level-0 slot table (56*40*13 = 29120 slots) longer than one pass of the single-workgroup scans
CASES = [(64, 64, 3000, 2, 50000), (256, 256, 20000, 3, 50000), (64, 64, 33000, 2, 50000), (256, 256, 6000, 2, 400000),
         (64, 64, 1500, 12, 50000)]


@pytest.mark.parametrize("W,H,n,B,span", CASES)
def test_backbone_matches_oracle(W, H, n, B, span):
    _check(*_setup(W, H, n, B, span))


# Sensor geometries of the DoTA-shape (1280x720, search radius 13 px) and dense-stress (640x480, radius 7 px)
# configurations: widths that are not powers of two, the layer-1 LUT too large for LDS (global-memory variant of the
# convolution), the spiral walk at large radii.  A 12 ms time span puts almost every event of a window inside delta_t, so
# most rows hit the 16-neighbour cap; the 50 ms cases leave rows partly filled.  The oracle evaluates LUT cells on
# demand (the full layer-5 LUT would take 14 GB at 1280x720).
BIG_CASES = [(640, 480, 60000, 2, 12000), (640, 480, 20000, 2, 50000), (1280, 720, 80000, 2, 12000), (1280, 720, 12000, 3, 50000)]


def test_large_radius_convolution_with_a_ragged_last_group():
    """Radius 13, 150 events crowded into a 40 x 40 corner: rows with neighbours in the outer rings (cells that are not
    LDS-resident) and a last group of 64 destinations that is mostly empty -- with the rows ranked by their outer tail, the
    empty rows form whole passes IN FRONT of the ones that hold rows."""
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    W, H, B = 1280, 720, 1
    raw = synthetic.make_batch(W, H, 150, B, with_image=False, precomputed_feats=True, t_span_us=12000)
    g = torch.Generator().manual_seed(9)
    raw.pos[:, 0] = torch.randint(0, 40, (150,), generator=g).to(raw.pos.dtype)
    raw.pos[:, 1] = torch.randint(0, 40, (150,), generator=g).to(raw.pos.dtype)
    data = dagr_ref.format_data(raw)
    sd = weights.random_graph_state()
    cfg = dagr_ref.make_config(W, H, B)
    (o3, o4), stages = dagr_ref.backbone_forward(sd, data, cfg, data.image_feats, return_all=True, mode="ondemand")
    dev = torch.device("cuda:0")
    nb = NativeBackbone(sd, W, H, dev)
    res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    assert nb.check_flags() & ~1 == 0
    N = data.pos.shape[0]
    steps = res["ws"]["ell_step"][:N].long() & 0xffff
    deg = res["ws"]["deg"][:N].long()
    live = torch.arange(16, device=dev)[None, :] < deg[:, None]
    assert int((live & (steps >= 117)).sum()) > 50               # outer-ring cells are really used
    _check(data, cfg, (o3, o4), stages, nb, res)


# ell_variant 2 forces the wave-per-pixel edge search (the default from 0.4 events per pixel) on these sparse
# inputs: 4 and 12 passes of 64 spiral steps, mostly empty neighbour pixels; 4 forces its ring-band form (row-major lane
# order inside bands of whole rings, source counts brought into spiral order through an LDS table; on request only)
@pytest.mark.parametrize("ell_variant", [None, 2, 4])
@pytest.mark.parametrize("W,H,n,B,span", BIG_CASES)
def test_backbone_matches_oracle_large_sensors(W, H, n, B, span, ell_variant):
    data, cfg, outs, stages, nb, res = _setup(W, H, n, B, span, mode="ondemand", ell_variant=ell_variant)
    deg = res["ws"]["deg"][: data.pos.shape[0]]
    if span <= 12000:
        assert float((deg == 16).float().mean()) > 0.3          # the neighbour cap really binds
    _check(data, cfg, outs, stages, nb, res)


def _check(data, cfg, outs, stages, nb, res):
    (o3, o4) = outs
    ws = res["ws"]
    N = data.pos.shape[0]
    # --- edges: bit exact against the oracle's edge list
    K = 16 * N
    edges = torch.empty((2, K), dtype=torch.int64, device="cuda")
    call("eventad_ell_to_edges", ptr(ws["ell_src"]), ptr(ws["deg"]), ptr(ws["sid"]), N, 16, K, ptr(edges),
         torch.cuda.current_stream().cuda_stream)
    edges = edges.cpu()
    edges = edges[:, edges[1] >= 0]
    assert torch.equal(edges, stages["edge_index0"])
    # --- layer 1 (sorted order -> event order)
    sid = ws["sid"].cpu().long()
    y2 = torch.empty(N, 16)
    y2[sid] = ws["y2"].cpu()
    ref = stages["layer1"].x
    assert torch.allclose(y2, ref, atol=2e-4, rtol=1e-4), float((y2 - ref).abs().max())
    # --- pooled levels
    for lv in range(4):
        n_l, x, pos, batch, ei = _level(ws["levels"][lv])
        p = stages["pool%d" % (lv + 1)]
        assert n_l == p.x.shape[0], (lv, n_l, p.x.shape)
        assert torch.equal(batch, p.batch)
        C = p.x.shape[1]
        assert torch.allclose(x[:, :C], p.x, atol=5e-4, rtol=1e-4), (lv, float((x[:, :C] - p.x).abs().max()))
        assert torch.allclose(pos[:, :2], p.pos[:, :2], atol=1e-7) and torch.allclose(pos[:, 2], p.pos[:, 2], atol=1e-5)
        assert torch.allclose(x[:, C:C + 2], p.pos[:, :2], atol=1e-7)
        # oracle edges are sorted by (src,dst); native CSR is grouped by dst with ascending src
        pe = p.edge_index
        order = torch.argsort(pe[1] * (n_l + 1) + pe[0])
        assert torch.equal(ei, pe[:, order]), lv
    for name, ref_o in (("out3", o3), ("out4", o4)):
        o = res[name]
        n_o = int(o["n_dev"].item())
        assert n_o == ref_o.x.shape[0]
        got = o["x"][:n_o].cpu()
        assert torch.allclose(got, ref_o.x, atol=1e-3, rtol=1e-3), (name, float((got - ref_o.x).abs().max()))
        assert torch.equal(o["batch"][:n_o].cpu().long(), ref_o.batch)


def test_reference_compatible_graph_entry_points():
    """eventad_insert_in_queue / eventad_fill_edges (drop-in for ev_graph_cuda) == oracle, bit for bit."""
    from oracle import graph as G
    W = H = 32; B = 2; Q = 8; K = 16; r = 2; dt = 3000
    data = synthetic.make_batch(W, H, 4000, B, with_image=False, t_span_us=20000)
    pos_i = torch.stack([data.pos[:, 0].int(), data.pos[:, 1].int(), data.t.int()], 1).contiguous()
    batch = data.batch.int()
    g = G.SlidingWindowGraphRef(W, H, B, K, Q, r, dt)
    want = g.forward(batch, pos_i)
    N = len(batch)
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    lin = pos_i[:, 0] + W * pos_i[:, 1] + W * H * batch
    s_lin, s_idx = torch.sort(lin, stable=True)
    uniq, counts = torch.unique_consecutive(s_lin, return_counts=True)
    indices = torch.arange(N, dtype=torch.int32)
    d = lambda t: t.contiguous().to(dev)
    queue = torch.full((B, Q, H, W), -1, dtype=torch.int32, device=dev)
    si, uq, cs = d(indices[s_idx]), d(uniq.int()), d(torch.cumsum(counts, 0).int())
    call("eventad_insert_in_queue", ptr(si), ptr(uq), ptr(cs), ptr(queue), B, Q, H, W, len(uniq), st)
    assert torch.equal(queue.cpu(), g.event_queue)
    edges = torch.full((2, K * N), -1, dtype=torch.int64, device=dev)
    b_d, p_d, ts_d, id_d = d(batch), d(pos_i), d(pos_i[:, 2]), d(indices)
    call("eventad_fill_edges", ptr(b_d), ptr(p_d), ptr(ts_d), ptr(queue), ptr(id_d), K, float(r), float(dt), ptr(edges),
         0, B, Q, H, W, N, K * N, st)
    e = edges.cpu()
    assert torch.equal(e[:, e[1] >= 0], want)
    # single-event variant
    q1 = torch.full((1, Q, H, W), -1, dtype=torch.int32, device=dev)
    one, ev = d(torch.tensor([7], dtype=torch.int32)), d(torch.tensor([3, 5], dtype=torch.int32))
    call("eventad_insert_in_queue_single", ptr(one), ptr(ev), ptr(q1), 1, Q, H, W, st)
    q1 = q1.cpu()
    assert int(q1[0, 0, 5, 3]) == 7 and int((q1 != -1).sum()) == 1


def test_gemm_matches_torch():
    g = torch.Generator().manual_seed(0)
    for M, N, K in [(1000, 1728, 84), (130, 64, 64), (257, 260, 132)]:
        A = torch.randn(M, K, generator=g).cuda()
        Bm = torch.randn(K, N, generator=g).cuda()
        C = torch.empty(M, N, device="cuda")
        call("eventad_gemm_f32", ptr(A), K, ptr(Bm), N, ptr(C), N, M, None, N, K, None, None, None, None, 0, 0,
             torch.cuda.current_stream().cuda_stream)
        ref = (A.double() @ Bm.double()).float()
        assert torch.allclose(C, ref, atol=1e-4, rtol=1e-4)


def _ell_rows(nb, data, B, dev):
    res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    ws = res["ws"]
    N = data.pos.shape[0]
    live = torch.arange(16, device=dev).view(1, -1) < ws["deg"][:N].view(-1, 1)
    return ws["deg"][:N].clone(), torch.where(live, ws["ell_src"][:N], -1), torch.where(live, ws["ell_step"][:N].int(), -1)


def test_wave_per_pixel_edge_search_equals_the_walk_on_dense_input():
    """640x480 (search radius 7: 225 spiral steps = 4 passes of 64) at 4.9 events per pixel, a third of the pixels deeper
    than the 8 entries a lane keeps in registers, plus one pixel with 300 events (more than 64 destinations, deeper than
    the FIFO): the wave-per-pixel search (what this density gets by default) and the per-event walk (compared with the
    oracle in the tests above) give identical rows."""
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    W, H, B = 640, 480, 1
    raw = synthetic.make_batch(W, H, 1500000, B, with_image=False, precomputed_feats=True)
    raw.pos[1000:1300] = torch.tensor([100, 50], dtype=raw.pos.dtype)
    data = dagr_ref.format_data(raw)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
    outs = []
    for variant in (3, None, 2, 4, 5, 6):
        nb.ell_variant = variant
        outs.append(_ell_rows(nb, data, B, dev))
    assert int((outs[0][0] < 16).sum()) > 1000 and float((outs[0][0] == 16).float().mean()) > 0.5
    assert int(outs[0][2].max()) > 64                            # sources beyond the first pass of 64 steps
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_deep_pixel_edge_search_equals_the_walk():
    """Tens of events per pixel (the 32 M-event configuration's density, here 64x64 x 2 clips): 24 events per pixel on
    average, one pixel with 300 events (more than the FIFO keeps), a time window shorter than the clip so that the window's
    lower end moves through the segments too, and a window beyond float precision.  The deep-pixel first pass (what this
    density gets by default, and forced as variant 5) gives the rows of the per-event walk."""
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    W = H = 64
    B = 2
    raw = synthetic.make_batch(W, H, 100000, B, with_image=False, precomputed_feats=True)
    raw.pos[5000:5300] = torch.tensor([10, 20], dtype=raw.pos.dtype)
    data = dagr_ref.format_data(raw)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
    for delta_t in (nb.geom.delta_t_us, 3000, (1 << 24) + 5):
        nb.geom.delta_t_us = delta_t
        outs = []
        for variant in (3, None, 5, 2, 6):
            nb.ell_variant = variant
            outs.append(_ell_rows(nb, data, B, dev))
        if delta_t == 3000:
            assert int((outs[0][0] < 16).sum()) > 1000           # the short window leaves rows unfilled
        else:
            assert float((outs[0][0] == 16).float().mean()) > 0.5
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert torch.equal(a, b)


def test_wave_per_pixel_edge_search_with_a_time_window_beyond_float_precision():
    """delta_t >= 2^24 us: (float)(ts - t) > (float)delta_t is no longer the integer comparison, so the wave-per-pixel
    search takes its float-compare instantiation; every older event of the window then qualifies.  Same rows as the walk."""
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    W = H = 64
    data = dagr_ref.format_data(synthetic.make_batch(W, H, 20000, 2, with_image=False, precomputed_feats=True))
    dev = torch.device("cuda:0")
    nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
    nb.geom.delta_t_us = (1 << 24) + 5
    outs = []
    for variant in (3, 2, 4, 5):
        nb.ell_variant = variant
        outs.append(_ell_rows(nb, data, 2, dev))
    assert float((outs[0][0] == 16).float().mean()) > 0.5
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_both_edge_search_kernels_give_the_same_rows():
    """Row scan (sparse windows), register-row spiral walk (dense windows) and wave-per-pixel search (dense input) on the
    same batch: identical ELL rows, degrees, pooled graph -- whichever the occupancy heuristic would pick."""
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    W = H = 256
    data = synthetic.make_batch(W, H, 30000, 3, with_image=False, precomputed_feats=True)
    data = dagr_ref.format_data(data)
    dev = torch.device("cuda:0")
    outs = []
    for variant in (0, 1, 2, 3, 4, 5, 6):
        nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
        nb.ell_variant = variant
        res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), 3, [m.to(dev) for m in data.image_feats])
        torch.cuda.synchronize()
        ws = res["ws"]
        live = torch.arange(16, device=dev).view(1, -1) < ws["deg"].view(-1, 1)
        L = ws["levels"][0]
        n, e = int(L.n_dev.item()), int(L.e_dev.item())
        outs.append((ws["deg"].clone(), torch.where(live, ws["ell_src"], -1), torch.where(live, ws["ell_step"].int(), -1),
                     L.csr_src[:e].clone(), L.rowptr[: n + 1].clone(), res["out4"]["x"][: int(res["out4"]["n_dev"].item())].clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_pooled_positions_at_full_density_and_non_power_of_two_width():
    """pool1 node positions at 1280x720 with 500k events per clip (~220 events per voxel).  The reference averages
    normalised fp32 positions (in atomic order on its GPU, sequentially in the oracle) and floors (mean + 1e-5) * W;
    the HIP path averages exact integer pixel sums.  For a width that is not a power of two the two can only differ
    where the exact value (sum/n + 1e-5 W) lies within float error of an integer -- possible only for voxels with
    >= 77 members -- and then by exactly one pixel.  Everything else must be equal."""
    from fractions import Fraction
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    W, H, n, B = 1280, 720, 500000, 2
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True))
    cfg = dagr_ref.make_config(W, H, B)
    _, pos, batch, _, _ = dagr_ref.pooling(torch.zeros(n * B, 1), data.pos, data.batch, torch.zeros((2, 0), dtype=torch.long),
                                           cfg.voxel[0], cfg.cart_max[1], "max", W, H, B)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
    res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    assert nb.check_flags() & ~1 == 0
    L = res["ws"]["levels"][0]
    n_l = int(L.n_dev.item())
    assert n_l == pos.shape[0]
    got = L.npos[:n_l].cpu()
    assert torch.equal(L.nbatch[:n_l].cpu().long(), batch)
    assert torch.allclose(got[:, 2], pos[:, 2], atol=1e-5)
    bad = torch.nonzero((got[:, :2] != pos[:, :2]).any(1)).flatten().tolist()
    if bad:
        # exact per-voxel pixel sums from the integer inputs
        px = torch.round(data.pos[:, 0] * W).long(); py = torch.round(data.pos[:, 1] * H).long()
        vs = cfg.voxel[0]
        start, end = torch.Tensor([0, 0, 0, 0]), torch.Tensor([0.9999999, 0.9999999, 0.9999999, B - 1])
        cl = dagr_ref.grid_cluster(torch.cat([data.pos, data.batch.float().view(-1, 1)], 1), vs, start, end)
        _, inv = torch.unique(cl, sorted=True, return_inverse=True)
        for j in bad:
            m = inv == j
            cnt = int(m.sum())
            assert cnt >= 77, cnt
            for d, (pix, size) in enumerate(((px, W), (py, H))):
                if got[j, d] != pos[j, d]:
                    q = Fraction(int(pix[m].sum()), cnt) + Fraction(size, 100000)
                    assert abs(q - round(q)) < Fraction(1, 1000), (j, d, float(q))
                    assert abs(round(float(got[j, d]) * size) - round(float(pos[j, d]) * size)) == 1
    print("pool1 nodes %d, positions off by one pixel at a rounding boundary: %d" % (n_l, len(bad)))


@pytest.mark.parametrize("W,H,n,B,span", [(256, 256, 20000, 3, 50000), (640, 480, 20000, 2, 50000)])
def test_fused_blend_and_pooling_equal_the_separate_launches(W, H, n, B, span):
    """A pooled Layer's last blend that also scatters into the next level's voxel table (eventad_tail_blend_pool, + the mean
    gather for pool4) leaves the same levels, bit for bit, as eventad_tail_blend followed by eventad_pool_scatter."""
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True, t_span_us=span))
    sd = weights.random_graph_state()
    dev = torch.device("cuda:0")
    levels = []
    for fused in (True, False):
        nb = NativeBackbone(sd, W, H, dev)
        nb.fused_blend_pool = fused
        nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
        torch.cuda.synchronize()
        assert nb.check_flags() & ~1 == 0
        levels.append([(_level(L), L.y[: int(L.n_dev.item())].cpu(), L.csr_dxy[: int(L.e_dev.item())].cpu()) for L in nb._tail_view])
    for (la, ya, da), (lb, yb, db) in zip(*levels):
        assert la[0] == lb[0] and all(torch.equal(p, q) for p, q in zip(la[1:], lb[1:]))
        assert torch.equal(ya, yb) and torch.equal(da, db)


@pytest.mark.parametrize("W,H,n,B,span", [(256, 256, 20000, 3, 50000), (64, 64, 3000, 2, 50000), (1280, 720, 30000, 1, 50000)])
def test_pooled_products_only_for_the_cells_in_use_give_the_same_levels(W, H, n, B, span):
    """The pooled levels' Z = X.[W_0..W_24|root|skip] without the 64x64 tiles of kernel cells that no edge's 4-tap blend
    reads (eventad_tail_live_cells + eventad_tail_gemm) leaves every level bit for bit as the full products do -- also when
    the skipped parts of Z hold NaN -- and the masks name exactly the cells the edges use."""
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True, t_span_us=span))
    sd = weights.random_graph_state()
    dev = torch.device("cuda:0")
    args = (data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    levels, skipped = [], 0
    for live in (True, False):
        nb = NativeBackbone(sd, W, H, dev)
        nb.tail_live_cells = live
        nb.forward(*args)
        for L in nb.ws["levels"]:
            L.z1.fill_(float("nan")); L.z2.fill_(float("nan"))
        nb.forward(*args)
        torch.cuda.synchronize()
        assert nb.check_flags() & ~1 == 0
        levels.append([(_level(L), L.y[: int(L.n_dev.item())].cpu(), L.y1[: int(L.n_dev.item())].cpu()) for L in nb._tail_view])
        if live:
            for lv, L in enumerate(nb._tail_view):
                if L.n_max < 4096:                           # small levels keep the full products
                    continue
                nn, ne = int(L.n_dev.item()), int(L.e_dev.item())
                src, pk = L.csr_src[:ne].long().cpu(), L.csr_dxy[:ne].cpu()
                dx, dy = ((pk & 0xffff) ^ 0x8000) - 0x8000, pk >> 16
                dvx, dvy = nb.geom.lut_div[lv + 1]
                want = torch.zeros((nn + 63) // 64, dtype=torch.int64)
                tx, ty = ((dx.float() / dvx + 0.5) * 4.0).long(), ((dy.float() / dvy + 0.5) * 4.0).long()
                for cx in (tx % 5, (tx + 1) % 5):
                    for cy in (ty % 5, (ty + 1) % 5):
                        for t in range(want.numel()):
                            m = (src >> 6) == t
                            want[t] |= int((1 << (cx[m] + 5 * cy[m])).unique().sum())     # distinct powers of two: sum == OR
                got = L.live[: want.numel()].cpu().long() & 0xFFFFFFFF
                assert torch.equal(got, want), lv
                skipped += sum(25 - bin(int(w)).count("1") for w in want)
                z = L.z1[:nn].view(nn, 27, 64)
                assert bool(torch.isfinite(z[:, 25:]).all())             # root and skip blocks always computed
    assert skipped > 0
    for (la, ya, y1a), (lb, yb, y1b) in zip(*levels):
        assert la[0] == lb[0] and all(torch.equal(p, q) for p, q in zip(la[1:], lb[1:]))
        assert bool(torch.isfinite(ya).all()) and torch.equal(ya, yb) and torch.equal(y1a, y1b)
