"""GPU: backward of the event-level stage (csrc/bwd.hip) against torch autograd through the CPU oracle on the same
graph: dL/dx, dL/dweight, dL/droot of both convolutions of the first Layer (oracle `spline_conv`), the transposed
adjacency against a CPU transpose of the edge list, pool1's voxel max (oracle `pooling`) and the bilinear feature
samplers (oracle `sample_features`).  Tolerance: fp32 sums of <= a few hundred terms."""
import pytest
import torch

from eventad_amd import synthetic, weights

pytestmark = pytest.mark.gpu


# 320x240: search radius 4 px, 81 spiral steps -- both kernels then ask for more than 64 KB of dynamic LDS; 640x480: radius 7,
# 225 steps -- the data gradient reads the table from memory, the weight gradient sorts 226 bins in 140 KB of LDS
@pytest.mark.parametrize("W,H,n,B,span", [(64, 64, 3000, 2, 50000), (256, 256, 20000, 2, 50000), (64, 64, 20000, 1, 20000), (320, 240, 12000, 2, 30000),
                                          (640, 480, 30000, 2, 12000)])
def test_event_conv_backward_matches_autograd(W, H, n, B, span):
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    from eventad_amd.graph_backward import EventConvBackward
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True, t_span_us=span))
    sd = weights.random_graph_state()
    cfg = dagr_ref.make_config(W, H, B)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(sd, W, H, dev)
    nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    ws, N = nb.ws, data.pos.shape[0]
    sid = ws["sid"][:N].cpu().long()                                         # event id at sorted position
    # the oracle's graph and edge attributes, event numbering
    ei = dagr_ref.G.build_edges(data.pos, data.batch, W, H, cfg.time_window, B, cfg.radius, cfg.max_neighbors)
    attr = torch.clamp(dagr_ref.cartesian(data.pos, ei, cfg.cart_max[0]), min=0, max=1)
    bw = EventConvBackward(nb.geom, dev)
    bw.transpose(ws, N)
    torch.cuda.synchronize()
    # ---- transposed adjacency == CPU transpose (self edges included: slot 0 of every row)
    deg = ws["deg"][:N].cpu().long()
    E = int(deg.sum())
    assert int(bw.out_start[N].item()) == E == ei.shape[1]
    start = bw.out_start[: N + 1].cpu().long()
    keys = bw.out_edge[:E].cpu().long()
    src_of_key = ws["ell_src"][:N].cpu().long().view(-1)[keys]
    owner = torch.repeat_interleave(torch.arange(N), start[1:] - start[:-1])
    assert torch.equal(src_of_key, owner)                                    # every key sits in its source's list
    assert bool(((keys[1:] > keys[:-1]) | (owner[1:] != owner[:-1])).all())  # ascending inside a list
    assert len(torch.unique(keys)) == E
    g = torch.Generator().manual_seed(3)
    for blk, cin, cin_pad in (("conv_block1.", 19, 20), ("conv_block2.", 16, 16)):
        p = "backbone.conv_block1." + blk
        weight, root = sd[p + "conv.weight"], sd[p + "conv.lin.weight"]
        x_ev = torch.randn(N, cin, generator=g)                              # event order
        G_ev = torch.randn(N, 16, generator=g)
        xo = x_ev.clone().requires_grad_(True)
        wo, ro = weight.clone().requires_grad_(True), root.clone().requires_grad_(True)
        out = dagr_ref.spline_conv(xo, ei, attr, wo, ro, cfg.lut[0], H, W, mode="gather")
        (out * G_ev).sum().backward()
        x_s = torch.zeros(N, cin_pad)
        x_s[:, :cin] = x_ev[sid]
        dx, dw, dr = bw.backward(ws, N, x_s.to(dev), G_ev[sid].contiguous().to(dev), weight, root)
        torch.cuda.synchronize()
        dx_ev = torch.empty(N, cin)
        dx_ev[sid] = dx[:, :cin].cpu()
        scale = lambda t: float(t.abs().max())
        assert float((dx_ev - xo.grad).abs().max()) < 2e-5 * max(1.0, scale(xo.grad)), blk
        assert float((dw - wo.grad).abs().max()) < 1e-4 * max(1.0, scale(wo.grad)), (blk, float((dw - wo.grad).abs().max()), scale(wo.grad))
        assert float((dr - ro.grad).abs().max()) < 1e-4 * max(1.0, scale(ro.grad)), blk
        dx2, dw2, dr2 = bw.backward(ws, N, x_s.to(dev), G_ev[sid].contiguous().to(dev), weight, root)
        assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(dr, dr2)     # no float atomics: reproducible


@pytest.mark.parametrize("W,H,n,B,span", [(64, 64, 3000, 2, 50000), (256, 256, 20000, 3, 50000), (96, 64, 9000, 1, 20000)])
def test_pool1_and_sampler_backward_match_autograd(W, H, n, B, span):
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    from eventad_amd.graph_backward import EventPoolBackward
    batch = synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True, t_span_us=span)
    data = dagr_ref.format_data(batch)
    sd = weights.random_graph_state()
    cfg = dagr_ref.make_config(W, H, B)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(sd, W, H, dev)
    nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    ws, N = nb.ws, data.pos.shape[0]
    sid = ws["sid"][:N].cpu().long()
    g = torch.Generator().manual_seed(11)
    bw = EventPoolBackward(nb)
    # ---- pool1: random (tie-free) layer-1 output and 64-channel map, random cotangent on the occupied slots
    L = ws["levels"][0]
    y2_ev = torch.randn(N, 16, generator=g)
    map1 = torch.randn(tuple(data.image_feats[1].shape), generator=g)         # NCHW like the reference's maps
    yo, mo = y2_ev.clone().requires_grad_(True), map1.clone().requires_grad_(True)
    x = torch.cat((yo, dagr_ref.sample_features(data.pos, data.batch, mo, W, H)), dim=1)
    ei = dagr_ref.G.build_edges(data.pos, data.batch, W, H, cfg.time_window, B, cfg.radius, cfg.max_neighbors)
    new_x = dagr_ref.pooling(x, data.pos, data.batch, ei, cfg.voxel[0], cfg.cart_max[1], "max", W, H, cfg.batch_size)[0]
    slots = torch.unique(ws["node_slot0"][:N].cpu().long())                   # occupied slots ascending = the oracle's cluster order
    assert new_x.shape == (len(slots), 80)
    G_nodes = torch.randn(len(slots), 80, generator=g)
    (new_x * G_nodes).sum().backward()
    dfeat = torch.zeros(L.S, 80)
    dfeat[slots] = G_nodes
    y2_s = y2_ev[sid].contiguous().to(dev)
    map1_nhwc = map1.permute(0, 2, 3, 1).contiguous().to(dev)
    dy2, dsamp = bw.pool1(N, B, dfeat.to(dev), y2=y2_s, fmap=map1_nhwc)
    dmap1 = bw.sampler(N, B, dsamp, 64, map1.shape[2], map1.shape[3])
    torch.cuda.synchronize()
    dy2_ev = torch.empty(N, 16)
    dy2_ev[sid] = dy2.cpu()
    assert torch.equal(dy2_ev, yo.grad)                                       # pure routing: exact
    ref = mo.grad.permute(0, 2, 3, 1)
    assert float((dmap1.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    # every (slot, channel) gradient lands on exactly one member
    assert torch.equal((dsamp != 0).sum(0).cpu(), (G_nodes[:, 16:] != 0).sum(0))
    # ---- sampler of Layer 1's input: 16-channel map, gradient rows laid out like x0s (stride 20, channels 1..16)
    map0 = torch.randn(tuple(data.image_feats[0].shape), generator=g).requires_grad_(True)
    G0 = torch.randn(N, 16, generator=g)
    (dagr_ref.sample_features(data.pos, data.batch, map0, W, H) * G0).sum().backward()
    rows = torch.zeros(N, 20)
    rows[:, 1:17] = G0[sid]
    dmap0 = bw.sampler(N, B, rows.to(dev), 16, map0.shape[2], map0.shape[3], column=1)
    ref0 = map0.grad.permute(0, 2, 3, 1)
    assert float((dmap0.cpu() - ref0).abs().max()) < 2e-5 * max(1.0, float(ref0.abs().max()))
    dmap0b = bw.sampler(N, B, rows.to(dev), 16, map0.shape[2], map0.shape[3], column=1)
    assert torch.equal(dmap0, dmap0b)                                         # fixed summation order: reproducible
