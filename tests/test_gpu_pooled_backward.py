"""GPU: backward of the pooled levels (csrc/bwd_pooled.hip: Layers 2..5, pools 2..4, samplers at pooled nodes) against
torch autograd through the CPU oracle (`layer`, `sample_features`, `pooling`) started from the same level-0 graph.

Compared: dL/d(level-0 features), dL/d(CNN maps 2..4) and dL/d of all 20 weight tensors of Layers 2..5, for a random
cotangent on out3 and out4.  The native forward and the oracle's differ by fp32 rounding; a ReLU whose pre-activation
lies within that rounding of zero could come out on different sides in the two, which changes a gradient by far more
than rounding, so the test first asserts that every ReLU mask of the two forwards agrees (a property of the seed).
Tolerance: fp32 sums of up to a few 10^5 terms, relative to the largest entry of each gradient."""
import pytest
import torch

from eventad_amd import synthetic, weights

pytestmark = pytest.mark.gpu

NAMES = ["conv_block1.", "layer2.", "layer3.", "layer4.", "layer5."]


def _oracle_pooled_stage(dagr_ref, sd, cfg, p1, maps, W, H):
    """The part of Net.forward (net.py:137-197) behind pool1, differentiable in p1.x, maps and sd."""
    x, pos, batch, ei, ea = p1.x, p1.pos, p1.batch, p1.edge_index, p1.edge_attr
    x = torch.cat((x, pos[:, :2]), dim=1)
    ys, o3 = [], None
    for li in range(1, 5):
        x = dagr_ref.layer(sd, "backbone." + NAMES[li], x, ei, ea, cfg.lut[li], H, W, "ondemand")
        ys.append(x)
        if li == 3:
            o3 = x
        if li == 4:
            break
        x = torch.cat((x, dagr_ref.sample_features(pos, batch, maps[li - 1], W, H)), dim=1)
        x, pos, batch, ei, ea = dagr_ref.pooling(x, pos, batch, ei, cfg.voxel[li], cfg.cart_max[li + 1], cfg.pool_aggr[li],
                                                 W, H, cfg.batch_size)
        x = torch.cat((x, pos[:, :2]), dim=1)
    return o3, x, ys


@pytest.mark.parametrize("W,H,n,B,span,seed", [(64, 64, 3000, 2, 50000, 5), (256, 256, 20000, 2, 50000, 6), (96, 64, 6000, 3, 20000, 7)])
def test_pooled_levels_backward_matches_autograd(W, H, n, B, span, seed):
    from types import SimpleNamespace
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    from eventad_amd.graph_backward import PooledBackward
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True, t_span_us=span))
    sd = weights.random_graph_state()
    cfg = dagr_ref.make_config(W, H, B)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(sd, W, H, dev)
    nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    assert nb.check_flags() & ~1 == 0
    with torch.no_grad():
        _, stages = dagr_ref.backbone_forward(sd, data, cfg, data.image_feats, return_all=True, mode="ondemand")
    s1 = stages["pool1"]
    tail_keys = [k for k in sd if k.startswith("backbone.layer") and (k.endswith("conv.weight") or k.endswith("lin.weight")
                                                                      or k.endswith("mlp.weight"))]
    assert len(tail_keys) == 20
    sdg = {k: (v.clone().requires_grad_(True) if k in tail_keys else v) for k, v in sd.items()}
    x0 = s1.x.clone().requires_grad_(True)
    maps = [m.clone().requires_grad_(True) for m in data.image_feats[2:]]
    p1 = SimpleNamespace(x=x0, pos=s1.pos, batch=s1.batch, edge_index=s1.edge_index, edge_attr=s1.edge_attr)
    o3, o4, ys = _oracle_pooled_stage(dagr_ref, sdg, cfg, p1, maps, W, H)
    bw = PooledBackward(nb)
    cnt_n, cnt_e = bw.counts()
    assert cnt_n == [y.shape[0] for y in ys]
    view = nb._tail_view
    for lv in range(4):                                     # precondition: the two forwards agree on every ReLU mask
        for mine, ref in ((view[lv].y[: cnt_n[lv]].cpu(), ys[lv].detach()),):
            assert float((mine - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
            assert torch.equal(mine > 0, ref > 0), "seed precondition: a ReLU mask differs between the two forwards"
    g = torch.Generator().manual_seed(seed)
    G3, G4 = torch.randn(o3.shape, generator=g), torch.randn(o4.shape, generator=g)
    ((o3 * G3).sum() + (o4 * G4).sum()).backward()
    res = bw.backward(G3.to(dev), G4.to(dev))
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.cpu() - b).abs().max()) / max(1e-6, float(b.abs().max()))
    assert rel(res["dx0"], x0.grad) < 2e-4, rel(res["dx0"], x0.grad)
    for k in range(3):
        assert rel(res["dmaps"][k], maps[k].grad.permute(0, 2, 3, 1)) < 2e-4, (k, rel(res["dmaps"][k], maps[k].grad.permute(0, 2, 3, 1)))
    assert sorted(res["weights"]) == sorted(tail_keys)
    for k in tail_keys:
        assert res["weights"][k].shape == sd[k].shape, k
        assert rel(res["weights"][k], sdg[k].grad) < 5e-4, (k, rel(res["weights"][k], sdg[k].grad))
    res2 = bw.backward(G3.to(dev), G4.to(dev))            # no float atomics, fixed summation orders: reproducible
    assert torch.equal(res["dx0"], res2["dx0"]) and all(torch.equal(a, b) for a, b in zip(res["dmaps"], res2["dmaps"]))
    assert all(torch.equal(res["weights"][k], res2["weights"][k]) for k in tail_keys)


def test_csr_transpose_lists_every_edge_once_in_order():
    from eventad_amd._lib import call, ptr
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    n = 5000
    deg = torch.randint(0, 40, (n,), generator=g)
    deg[7] = 0
    rowptr = torch.zeros(n + 1, dtype=torch.long)
    rowptr[1:] = deg.cumsum(0)
    E = int(rowptr[-1])
    src = torch.randint(0, n, (E,), generator=g)
    src[:600] = 11                                          # one source with a long list
    rowptr_d, src_d = rowptr.to(torch.int32).to(dev), src.to(torch.int32).to(dev)     # named: alive until the call is issued
    out_start, out_edge, edge_dst = (torch.empty(k, dtype=torch.int32, device=dev) for k in (n + 1, E, E))
    scratch = [torch.empty(k, dtype=torch.int32, device=dev) for k in (n + 1, n // 2048 + 4, E)]
    call("eventad_csr_transpose", ptr(rowptr_d), ptr(src_d), n, E, ptr(out_start), ptr(out_edge), ptr(edge_dst),
         ptr(scratch[0]), ptr(scratch[1]), ptr(scratch[2]), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    start, edges = out_start.cpu().long(), out_edge.cpu().long()
    assert torch.equal(start[1:] - start[:-1], torch.bincount(src, minlength=n)) and int(start[0]) == 0
    owner = torch.repeat_interleave(torch.arange(n), start[1:] - start[:-1])
    assert torch.equal(src[edges], owner)
    assert bool(((edges[1:] > edges[:-1]) | (owner[1:] != owner[:-1])).all())
    assert torch.equal(edge_dst.cpu().long(), torch.repeat_interleave(torch.arange(n), deg))


@pytest.mark.parametrize("W,H,n,B,span,seed", [(64, 64, 3000, 2, 50000, 15), (96, 64, 6000, 3, 20000, 17), (256, 256, 20000, 2, 50000, 16),
                                               (640, 480, 20000, 2, 12000, 18),       # search radius 7 px: table of the data gradient in memory
                                               (1280, 720, 12000, 2, 12000, 19)])     # radius 13: 730 LUT cells, the weight gradient's LDS sort in 4 bands
def test_whole_graph_branch_backward_matches_autograd(W, H, n, B, span, seed):
    """out3 / out4 cotangents -> gradients of all 25 weight tensors of the five Layers and of the five CNN maps, against
    autograd through the oracle's whole Net.forward (same mask precondition as above, here also for Layer 1)."""
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    from eventad_amd.graph_backward import BackboneBackward
    data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True, t_span_us=span))
    sd = weights.random_graph_state()
    cfg = dagr_ref.make_config(W, H, B)
    dev = torch.device("cuda:0")
    nb = NativeBackbone(sd, W, H, dev)
    out = nb.forward(data.pos.to(dev), data.x.to(dev), data.batch.to(dev), B, [m.to(dev) for m in data.image_feats])
    torch.cuda.synchronize()
    assert nb.check_flags() & ~1 == 0
    keys = [k for k in sd if k.endswith("conv.weight") or k.endswith("lin.weight") or k.endswith("mlp.weight")]
    assert len(keys) == 25
    sdg = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    maps = [m.clone().requires_grad_(True) for m in data.image_feats]
    (o3, o4), stages = dagr_ref.backbone_forward(sdg, data, cfg, maps, return_all=True, mode="ondemand")
    N = data.pos.shape[0]
    sid = nb.ws["sid"][:N].cpu().long()
    y2 = torch.empty(N, 16)
    y2[sid] = nb.ws["y2"][:N].cpu()
    assert torch.equal(y2 > 0, stages["layer1"].x.detach() > 0), "seed precondition: a ReLU mask of Layer 1 differs"
    for lv in range(4):
        ref = stages["layer%d" % (lv + 2)].x.detach()
        assert torch.equal(nb._tail_view[lv].y[: ref.shape[0]].cpu() > 0, ref > 0), "seed precondition: a ReLU mask differs"
    g = torch.Generator().manual_seed(seed)
    G3, G4 = torch.randn(o3.x.shape, generator=g), torch.randn(o4.x.shape, generator=g)
    ((o3.x * G3).sum() + (o4.x * G4).sum()).backward()
    bw = BackboneBackward(nb, sd)
    res = bw.backward(G3.to(dev), G4.to(dev))
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.cpu() - b).abs().max()) / max(1e-6, float(b.abs().max()))
    for k in range(5):
        assert rel(res["dmaps"][k], maps[k].grad.permute(0, 2, 3, 1)) < 5e-4, (k, rel(res["dmaps"][k], maps[k].grad.permute(0, 2, 3, 1)))
    assert sorted(res["weights"]) == sorted(keys)
    for k in keys:
        assert res["weights"][k].shape == sd[k].shape, k
        assert rel(res["weights"][k], sdg[k].grad) < 1e-3, (k, rel(res["weights"][k], sdg[k].grad))
    res2 = bw.backward(G3.to(dev), G4.to(dev))
    assert all(torch.equal(res["weights"][k], res2["weights"][k]) for k in keys)
    assert all(torch.equal(a, b) for a, b in zip(res["dmaps"], res2["dmaps"]))


@pytest.mark.parametrize("W,H,n,B", [(64, 64, 3000, 2), (1280, 720, 6000, 1)])
def test_loss_backward_reaches_the_backbone_weights(W, H, n, B):
    """Net.forward_with_grad: out3.x / out4.x carry autograd history, so a loss on them fills .grad of the 25 weight tensors
    of the five Layers and of precomputed CNN maps -- with exactly what BackboneBackward returns (checked against autograd
    through the oracle above) -- and an optimiser step on those weights changes the next pass."""
    import eventad_amd.compat as compat
    compat.install()
    from config.eventad_config import parse_eventad_args
    from dagr.model.networks.dagr import DAGR
    from dagr.utils.buffers import format_data
    from eventad_amd.graph_backward import BackboneBackward
    dev = torch.device("cuda:0")
    args = parse_eventad_args([])
    args.batch_size = B
    sd = weights.random_graph_state()
    dagr = DAGR(args, height=H, width=W).to(dev)
    res = dagr.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    dagr.cache_luts(radius=args.radius, height=H, width=W)
    net = dagr.backbone
    data = format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True, t_span_us=12000).to(dev))
    data.image_feats = [m.clone().requires_grad_(True) for m in data.image_feats]
    (o3, o4), _ = net.forward_with_grad(data)
    assert o3.x.requires_grad and o4.x.requires_grad and o4.x.shape[0] == int(o4.n_dev.item())
    g = torch.Generator().manual_seed(5)
    G3, G4 = torch.randn(o3.x.shape, generator=g).to(dev), torch.randn(o4.x.shape, generator=g).to(dev)
    want = BackboneBackward(net._native, {"backbone." + k: v for k, v in net.state_dict().items()}).backward(G3, G4)
    ((o3.x * G3).sum() + (o4.x * G4).sum()).backward()
    named = dict(net.named_parameters())
    keys = net.grad_weight_keys()
    assert len(keys) == 25
    for k in keys:
        assert named[k].grad is not None and torch.equal(named[k].grad, want["weights"]["backbone." + k]), k
        assert float(named[k].grad.abs().max()) > 0
    for m, dm in zip(data.image_feats, want["dmaps"]):
        assert torch.equal(m.grad, dm.permute(0, 3, 1, 2))
    assert all(p.grad is None for k, p in named.items() if k not in keys)           # BatchNorm / CNN parameters: no gradient
    before = o4.x.detach().clone()
    torch.optim.SGD([named[k] for k in keys], lr=1e-2).step()
    (p3, p4), _ = net.forward_with_grad(data)                                       # rebuilds the native weight images
    assert p4.x.shape == before.shape and not torch.equal(p4.x.detach(), before)
    with pytest.raises(RuntimeError, match="next pass"):
        (q3, q4), _ = net.forward_with_grad(data)
        net(data)                                                                   # another pass in between
        q4.x.sum().backward()
