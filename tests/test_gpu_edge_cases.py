"""GPU parity on awkward inputs: ragged clips, an empty clip, a hot pixel with more events than the FIFO
depth (exercises the long-segment ordering kernel), non-monotone timestamps (slow scan path), a
non-square sensor, a single-event batch, and size-independent properties at the full C2 size."""
import pytest
import torch

from eventad_amd import synthetic, weights
from eventad_amd.batch import EventBatch
from eventad_amd._lib import call, ptr

pytestmark = pytest.mark.gpu


def _run(raw, W, H, B, ell_variant=None):
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    data = dagr_ref.format_data(raw.clone())
    sd = weights.random_graph_state()
    (o3, o4), stages = dagr_ref.backbone_forward(sd, data, dagr_ref.make_config(W, H, B), data.image_feats, return_all=True)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(sd, W, H, dev)
    if ell_variant is not None:
        nb.ell_variant = ell_variant
    res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    flags = nb.check_flags()
    ws = res["ws"]
    N = data.pos.shape[0]
    edges = torch.empty((2, 16 * max(N, 1)), dtype=torch.int64, device="cuda")
    call("eventad_ell_to_edges", ptr(ws["ell_src"]), ptr(ws["deg"]), ptr(ws["sid"]), N, 16, 16 * max(N, 1), ptr(edges),
         torch.cuda.current_stream().cuda_stream)
    edges = edges.cpu()[:, : 16 * N]
    edges = edges[:, edges[1] >= 0]
    assert torch.equal(edges, stages["edge_index0"])
    for name, ref in (("out3", o3), ("out4", o4)):
        o = res[name]
        n = int(o["n_dev"].item())
        assert n == ref.x.shape[0], (name, n, ref.x.shape)
        assert torch.equal(o["batch"][:n].cpu().long(), ref.batch)
        assert torch.allclose(o["x"][:n].cpu(), ref.x, atol=1e-3, rtol=1e-3), name
        assert torch.allclose(o["pos"][:n, :2].cpu(), ref.pos[:, :2], atol=1e-7)
    return flags, res, stages


def _clips(W, H, ns, **kw):
    return [synthetic.make_clip(i, W, H, n, with_image=False, precomputed_feats=True, **kw) for i, n in enumerate(ns)]


# ell_variant: None = the edge search the density heuristic picks, 2 = the wave-per-pixel search forced, 4 = its ring-band
# form forced (row-major lane order, non-empty pixels only; non-monotone batches: the exact walk launched behind it),
# 5 = its deep-pixel first pass forced (two positions per neighbour segment), 6 = its 16-step first pass forced
VARIANTS = [None, 2, 4, 5, 6]


@pytest.mark.parametrize("ell_variant", VARIANTS)
def test_ragged_clips_and_an_empty_clip(ell_variant):
    W = H = 64
    raw = EventBatch.from_clips(_clips(W, H, [1500, 0, 4000, 7]))
    _run(raw, W, H, 4, ell_variant)


@pytest.mark.parametrize("ell_variant", VARIANTS)
def test_hot_pixel_exceeds_fifo_depth(ell_variant):
    W = H = 64
    clips = _clips(W, H, [3000, 2500])
    clips[0]["pos"][500:900] = torch.tensor([10, 20], dtype=torch.int16)        # 400 events in one pixel (> Q = 128)
    clips[1]["pos"][::3] = torch.tensor([5, 5], dtype=torch.int16)              # ~830 events in one pixel
    flags, res, stages = _run(EventBatch.from_clips(clips), W, H, 2, ell_variant)
    assert int(res["ws"]["counters"][3].item()) >= 800                          # longest pixel segment, far beyond the FIFO depth


@pytest.mark.parametrize("depth", [20, 40])
def test_row_scan_with_long_runs_in_one_pixel(depth):
    """The row scan keeps a pixel's survivors as one (first offset, length) run with a 5-bit length: 20 events in one pixel
    stay on that path (runs longer than a row's 15 free slots are cut while the row is read off), 40 -- still inside the
    FIFO -- send the batch to the exact walk.  Rows against the oracle either way."""
    W = H = 64
    clips = _clips(W, H, [3000, 2500])
    clips[0]["pos"][1000:1000 + depth] = torch.tensor([30, 31], dtype=torch.int16)
    clips[1]["pos"][200:200 + 2 * depth:2] = torch.tensor([7, 50], dtype=torch.int16)
    flags, res, stages = _run(EventBatch.from_clips(clips), W, H, 2, 1)
    assert depth <= int(res["ws"]["counters"][3].item()) <= depth + 4            # longest pixel segment: no deeper pixel around


@pytest.mark.parametrize("ell_variant", VARIANTS)
def test_non_monotone_timestamps_take_the_exact_scan_path(ell_variant):
    W = H = 64
    clips = _clips(W, H, [2500, 2500], t_span_us=30000)
    g = torch.Generator().manual_seed(9)
    for c in clips:
        perm = torch.randperm(2500, generator=g)
        c["t"] = c["t"][perm].contiguous()                                      # ids no longer ordered by time
    flags, _, _ = _run(EventBatch.from_clips(clips), W, H, 2, ell_variant)
    assert flags & 1                                                            # EVENTAD_FLAG_T_NOT_MONOTONE raised


def test_non_square_sensor():
    W, H = 96, 64
    _run(EventBatch.from_clips(_clips(W, H, [3000, 2000])), W, H, 2)


def test_single_event_batch():
    W = H = 64
    _run(EventBatch.from_clips(_clips(W, H, [1])), W, H, 1)


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_full_size_properties(cfg):
    """At the full sizes of the benchmark configurations (C2: 400k events; C3: 6.4M events, spiral-walk edge search; C4: one GPU's 8 clips x 500k
    events at 1280x720, search radius 13, two-entry wave-per-pixel search and the tail-ranked convolutions; C5: 32M events at 640x480,
    search radius 7, rank-dealt spiral walk and the partly-in-LDS LUT) the oracle is too slow for the CI budget: check
    invariants of the graph and of every pooled level instead."""
    from eventad_amd.backbone import NativeBackbone
    from eventad_amd.dagr.utils.buffers import format_data
    W, H, n, B, live = synthetic.CONFIGS[cfg]
    data = format_data(synthetic.make_config_batch(cfg).to("cuda"))
    if live:                                                 # the trunk has its own tests: random maps of its output shapes
        g = torch.Generator().manual_seed(0)
        data.image_feats = [torch.randn(B, c, h, w, generator=g).to("cuda")
                            for c, (h, w) in zip(synthetic.FEATURE_CHANNELS, synthetic.feature_map_sizes(W, H))]
    nb = NativeBackbone(weights.random_graph_state(), W, H, torch.device("cuda:0"))
    res = nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
    torch.cuda.synchronize()
    assert nb.check_flags() == 0
    ws = res["ws"]
    N = B * n
    sid = ws["sid"].long()
    assert torch.equal(torch.sort(sid).values, torch.arange(N, device="cuda"))              # a permutation
    key_s = ws["key_s"].long()
    assert bool((key_s[1:] >= key_s[:-1]).all())                                            # sorted by pixel
    same = key_s[1:] == key_s[:-1]
    assert bool((sid[1:][same] > sid[:-1][same]).all())                                     # stable inside a pixel
    deg = ws["deg"].long()
    assert int(deg.min()) >= 1 and int(deg.max()) <= 16
    src = ws["ell_src"].long()
    rows = torch.arange(N, device="cuda")
    assert torch.equal(src[:, 0], rows)                                                     # self edge first
    k = torch.arange(16, device="cuda").view(1, -1)
    live = k < deg.view(-1, 1)
    s_ids, d_ids = sid[src.clamp(min=0)], sid.view(-1, 1).expand(-1, 16)
    assert bool((s_ids[live] <= d_ids[live]).all())                                         # older -> newer only
    assert bool((data.batch[s_ids[live]] == data.batch[d_ids[live]]).all())                 # never across clips
    t_us = (data.pos[:, 2] * 1e6)
    assert float((t_us[d_ids[live]] - t_us[s_ids[live]]).max()) <= 10000 + 1               # inside the time window
    px, py = key_s % W, (key_s // W) % H
    R = nb.geom.radius_px
    assert int((px[src.clamp(min=0)] - px.view(-1, 1)).abs()[live].max()) <= R              # inside the radius
    assert int((py[src.clamp(min=0)] - py.view(-1, 1)).abs()[live].max()) <= R
    del s_ids, d_ids, live, src
    caps = [2240, 560, 140, 35]
    at_window_end = int((data.pos[:, 2] >= 1.0).sum())       # events at t == time_window open voxels in the plane after their clip's
    for lv, L in enumerate(ws["levels"]):
        nl = int(L.n_dev.item())
        cnt = torch.bincount(L.nbatch[:nl].long(), minlength=B)
        assert int(cnt.max()) <= caps[lv] + at_window_end and nl > 0                        # voxel grid bound (+ the t==T quirk)
        e = int(L.e_dev.item())
        rp = L.rowptr[: nl + 1].long()
        assert int(rp[-1]) == e and bool((rp[1:] >= rp[:-1]).all())
        srcs = L.csr_src[:e].long()
        dst = torch.repeat_interleave(torch.arange(nl, device="cuda"), rp[1:] - rp[:-1])
        assert bool((srcs != dst).all()) and int(srcs.max()) < nl                           # no self loops after pooling
        key = dst * nl + srcs
        assert bool((key[1:] > key[:-1]).all())                                             # unique, sorted by (dst, src)
    out4 = res["out4"]
    n4 = int(out4["n_dev"].item())
    assert bool(torch.isfinite(out4["x"][:n4]).all()) and float(out4["x"][:n4].min()) >= 0  # ReLU output


def test_workspace_of_a_larger_batch_is_reused_for_a_smaller_one():
    """A loader's batches vary in size: after a large batch, a smaller one (same number of clips) must run in the same
    workspace and give exactly what a fresh backbone gives."""
    W = H = 64
    from eventad_amd.backbone import NativeBackbone
    from eventad_amd import weights
    from oracle import dagr_ref
    dev = torch.device("cuda:0")
    big = dagr_ref.format_data(EventBatch.from_clips(_clips(W, H, [5000, 4000, 6000])))
    small = dagr_ref.format_data(EventBatch.from_clips(_clips(W, H, [1200, 900, 2000])))
    run = lambda nb, d: nb.forward(d.pos.to(dev), d.x.to(dev), d.batch.to(dev), 3, [m.to(dev) for m in d.image_feats])
    nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
    run(nb, big)
    ws_id = id(nb.ws)
    got = run(nb, small)
    assert id(nb.ws) == ws_id                                                    # no re-allocation
    n = int(got["out4"]["n_dev"].item())
    x_reused = got["out4"]["x"][:n].clone()
    fresh = run(NativeBackbone(weights.random_graph_state(), W, H, dev), small)
    assert n == int(fresh["out4"]["n_dev"].item()) and torch.equal(x_reused, fresh["out4"]["x"][:n])
    assert nb.check_flags() & ~1 == 0


def test_rarely_taken_kernel_paths_give_the_same_results():
    """Code paths the default configuration never takes, each against the default path or the oracle:
    (1) slot tables beyond one pass of the single-workgroup scan kernel (nx*ny*(B+1) > 131072: B >= 58 clips) go through
        eventad_pool_finalize + eventad_pool_edges_emit; (2) pool1's neighbour bitmap marked by the edge search kernels (fused_mark = "edge_search",
        both kernels) or in a pass of its own over the ELL (False, eventad_pool_edges_mark_ell) instead of by the pooling kernel; (3) a neighbour cap other than 16 takes the generic edge-search
        kernel of eventad_build_ell."""
    import torch
    from oracle import dagr_ref, graph as G
    from eventad_amd import synthetic, weights
    from eventad_amd._lib import call, ptr
    from eventad_amd.backbone import NativeBackbone
    dev = torch.device("cuda:0")
    sd = weights.random_graph_state()
    # (1) 60 clips: level-0 table 56*40*61 = 136,640 slots
    W = H = 64; B = 60; n = 300
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True))
    (o3, o4) = dagr_ref.backbone_forward(sd, data, dagr_ref.make_config(W, H, B), data.image_feats)
    nb = NativeBackbone(sd, W, H, dev)
    res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    assert nb.ws["levels"][0].S > 131072 and nb.check_flags() & ~1 == 0
    n4 = int(res["out4"]["n_dev"].item())
    assert n4 == o4.x.shape[0] and torch.equal(res["out4"]["batch"][:n4].cpu().long(), o4.batch)
    assert torch.allclose(res["out4"]["x"][:n4].cpu(), o4.x, atol=1e-3, rtol=1e-3)
    # (2) the three places that can mark the bitmap give the same pooled graph
    W = H = 64; B = 3; n = 3000
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True))
    outs = []
    for fused, variant in ((True, None), (False, None), ("edge_search", 0), ("edge_search", 1)):
        nb = NativeBackbone(sd, W, H, dev)
        nb.fused_mark, nb.ell_variant = fused, variant
        res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
        torch.cuda.synchronize()
        L = nb.ws["levels"][0]
        e = int(L.e_dev.item())
        outs.append((L.csr_src[:e].clone(), L.rowptr[: int(L.n_dev.item()) + 1].clone(),
                     res["out4"]["x"][: int(res["out4"]["n_dev"].item())].clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    # (3) neighbour cap 8: generic kernel vs the oracle's builder with K = 8
    ws, N, g = nb.ws, data.pos.shape[0], nb.geom
    K = 8
    ell_src = torch.empty((N, K), dtype=torch.int32, device=dev)
    ell_step = torch.empty((N, K), dtype=torch.int16, device=dev)
    deg = torch.empty(N, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call("eventad_build_ell", ptr(ws["sid"]), ptr(ws["ts_s"]), ptr(ws["ev2"]), ptr(ws["key_s"]), ptr(ws["pix_start"]), ptr(ws["counters"]),
         N, B, W, H, g.radius_px, g.delta_t_us, K, g.Q, ptr(ell_src), ptr(ell_step), ptr(deg), None, None, 0, 0, None, 0, 1, st)
    edges = torch.empty((2, K * N), dtype=torch.int64, device=dev)
    call("eventad_ell_to_edges", ptr(ell_src), ptr(deg), ptr(ws["sid"]), N, K, K * N, ptr(edges), st)
    e = edges.cpu()
    want = G.build_edges(data.pos, data.batch, W, H, 1000000, B, 0.01, K)
    assert torch.equal(e[:, e[1] >= 0], want)


def test_channels_last_feature_maps_are_used_in_place():
    """[B,C,h,w] maps in torch's channels_last memory format are NHWC in memory: the backbone reads them without the
    layout-change launch and must give exactly what it gives for plain NCHW maps of the same values."""
    import torch
    from oracle import dagr_ref
    from eventad_amd import synthetic, weights
    from eventad_amd.backbone import NativeBackbone
    dev = torch.device("cuda:0")
    W = H = 64; B = 3; n = 4000
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True))
    sd = weights.random_graph_state()
    outs = []
    for fmt in ("nchw", "channels_last"):
        nb = NativeBackbone(sd, W, H, dev)
        maps = [m.to(dev) for m in data.image_feats]
        if fmt == "channels_last":
            maps = [m.contiguous(memory_format=torch.channels_last) for m in maps]
        nb.timing = {}
        res = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, maps)
        torch.cuda.synchronize()
        assert ("eventad_nchw_to_nhwc_group#1" in nb.timing) == (fmt == "nchw")
        n4, n3 = int(res["out4"]["n_dev"].item()), int(res["out3"]["n_dev"].item())
        outs.append((res["out4"]["x"][:n4].clone(), res["out3"]["x"][:n3].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
