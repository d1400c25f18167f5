"""GPU: backward of the event-level spline convolution (csrc/bwd.hip) against torch autograd through the CPU oracle's
`spline_conv` on the same graph -- dL/dx, dL/dweight, dL/droot for both convolutions of the first Layer -- and the
transposed adjacency against a CPU transpose of the edge list.  Tolerance: fp32 sums of <= a few hundred terms."""
import pytest
import torch

from eventad_amd import synthetic, weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,n,B,span", [(64, 64, 3000, 2, 50000), (256, 256, 20000, 2, 50000), (64, 64, 20000, 1, 20000)])
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
