"""GPU: the HIP head (csrc/head.hip + gemm.hip through the C ABI) against (a) the reference's golden
vectors and (b) torch autograd on the same masked formulation; fused clip+AdamW against torch.optim.AdamW."""
import os
from types import SimpleNamespace

import pytest
import torch

from tests.helpers import seeded_head_state, make_head_case, grad_sample_index

pytestmark = pytest.mark.gpu
CASES = ["head_case_small", "head_case_rol", "head_case_dense"]


def _model(seed):
    from eventad_amd.models.EventAD import EventADModel
    stub = torch.nn.Module()
    stub.width, stub.height = 256, 256
    m = EventADModel(stub).cuda()
    m.load_state_dict(seeded_head_state(seed), strict=False)
    return m.eval()


def _data(case):
    d = SimpleNamespace(**{k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in case.items()
                           if k in ("num_graphs", "bbox", "bbox_batch", "bbox0", "bbox0_batch", "width", "height")})
    return d


@pytest.mark.parametrize("name", CASES)
def test_native_head_matches_reference_golden_and_autograd(name, golden_dir):
    gold = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    kw = gold["kw"]
    m = _model(kw["seed"])
    data = _data(make_head_case(**kw))
    feats = gold["feats"].cuda()
    labels = data.bbox[:, 4]
    nh = m.native_head()
    loss, logits, valid, targets = nh.forward(feats, data.bbox, data.bbox_batch, labels, data.num_graphs, 256, 256, False)
    got = logits[valid].cpu()
    assert got.shape == gold["logits"].shape
    assert torch.equal(targets[valid].cpu(), gold["targets"])
    assert float((got - gold["logits"]).abs().max()) < 1e-4
    assert torch.equal(got.argmax(1), gold["logits"].argmax(1))
    assert abs(float(loss) - float(gold["loss"])) < 1e-4 * max(1.0, float(gold["loss"]))
    nh.backward()
    for k, g in gold["grads"].items():
        if g is None:
            continue
        mine = nh.G[k].flatten().cpu()
        assert abs(float(mine.norm()) - g["norm"]) <= 2e-4 * max(1.0, g["norm"]), k
        assert torch.allclose(mine[grad_sample_index(mine.numel(), kw["seed"])], g["sample"], atol=2e-5, rtol=2e-4), k
    # torch autograd over the same masked formulation (all entries, not a sample)
    native = {k: v.clone() for k, v in nh.G.items()}
    for p in m.parameters():
        p.grad = None
    loss_t, logits_t, valid_t, _ = m.head(feats, data, labels)
    loss_t.backward()
    assert torch.equal(valid_t, valid[:, :31])
    for k, p in m.named_parameters():
        if k in native:
            assert torch.allclose(native[k], p.grad, atol=2e-5, rtol=2e-4), k


def test_autograd_bridge_and_fused_adamw():
    kw = dict(seed=12, num_graphs=8, tracks=10, nodes_per_clip=35, drop_prob=0.2, empty_clip=None, far_box=True)
    m = _model(12)
    data = _data(make_head_case(**kw))
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(8, 2, 31, 64, generator=g).cuda()
    labels = data.bbox[:, 4]
    # reference-style loop: loss.backward(); clip; torch AdamW  -- on the autograd bridge
    ref = _model(12)
    loss_r, _, _, _ = ref._head_native(feats, data, labels)
    loss_r.backward()
    params_r = [p for n, p in ref.named_parameters() if p.grad is not None]
    norm_r = torch.nn.utils.clip_grad_norm_(params_r, 1.0)
    opt = torch.optim.AdamW(params_r, lr=1e-3, weight_decay=1e-5)
    opt.step()
    # native fused path
    nh = m.native_head(lr=1e-3, weight_decay=1e-5, grad_clip=1.0)
    with torch.no_grad():
        loss_n, _, _, _ = m._head_native(feats, data, labels)
    nh.backward()
    norm_n = nh.optimizer_step()
    assert abs(float(loss_n) - float(loss_r)) < 1e-5 * max(1.0, float(loss_r))
    assert abs(float(norm_n) - float(norm_r)) < 1e-4 * max(1.0, float(norm_r))
    pr = dict(ref.named_parameters())
    for k, v in nh.P.items():
        assert torch.allclose(v, pr[k].detach(), atol=1e-6, rtol=1e-5), k
    # second step exercises the bias-correction / moment state
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        l2, _, _, _ = ref._head_native(feats, data, labels)
        l2.backward()
        torch.nn.utils.clip_grad_norm_(params_r, 1.0)
        opt.step()
        with torch.no_grad():
            m._head_native(feats, data, labels)
        nh.backward()
        nh.optimizer_step()
    for k, v in nh.P.items():
        assert torch.allclose(v, pr[k].detach(), atol=2e-6, rtol=1e-4), k
    assert nh.nonfinite_steps() == 0


def test_gemm_variants():
    from eventad_amd.head_native import NativeHead
    g = torch.Generator().manual_seed(1)
    for (ta, tb, M, N, K) in [(0, 1, 32, 768, 256), (0, 0, 32, 256, 768), (1, 0, 768, 64, 256), (0, 1, 256, 96, 4),
                              (1, 0, 96, 4, 256), (0, 0, 256, 32, 256), (0, 1, 20, 70, 36), (1, 1, 100, 52, 40)]:
        A = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
        Bm = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
        C0 = torch.randn(M, N, generator=g).cuda()
        bias = torch.randn(N, generator=g).cuda()
        C = C0.clone()
        NativeHead._gemm(ta, tb, A, A.shape[1], Bm, Bm.shape[1], C, N, M, N, K, bias=bias, relu=0, acc=1)
        ref = ((A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double() + bias.double() + C0.double()).float()
        assert torch.allclose(C, ref, atol=2e-4, rtol=1e-4), (ta, tb, M, N, K, float((C - ref).abs().max()))


def test_grouped_tile_gemm():
    import ctypes
    from eventad_amd._lib import call
    g = torch.Generator().manual_seed(2)
    shapes = [(0, 1, 32, 768, 256), (0, 0, 32, 256, 768), (1, 0, 768, 64, 256), (0, 1, 256, 96, 4), (1, 0, 96, 4, 256),
              (0, 0, 256, 32, 256), (0, 1, 20, 70, 36), (1, 1, 100, 52, 40)]
    probs, refs = [], []
    for (ta, tb, M, N, K) in shapes:
        A = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
        Bm = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
        C0 = torch.randn(M, N, generator=g).cuda()
        bias = torch.randn(N, generator=g).cuda()
        C = C0.clone()
        part = torch.empty(((K + 255) // 256) * M * N, device="cuda")
        probs.append((A, Bm, part, C, bias, A.shape[1], Bm.shape[1], N, M, N, K, ta, tb, 1, 1))
        refs.append(torch.relu((A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double() + bias.double()).float() + C0)
    d = (ctypes.c_int64 * (16 * len(probs)))()
    for i, pr in enumerate(probs):
        d[16 * i:16 * i + 15] = [v.data_ptr() if isinstance(v, torch.Tensor) else v for v in pr]
    call("eventad_gemm_group", len(probs), ctypes.cast(d, ctypes.c_void_p), torch.cuda.current_stream().cuda_stream)
    for pr, ref, shp in zip(probs, refs, shapes):
        assert torch.allclose(pr[3], ref, atol=2e-4, rtol=1e-4), (shp, float((pr[3] - ref).abs().max()))


def test_native_head_large_batch_matches_autograd():
    """B = 32 (R = 1024 rows): the weight-gradient products need 4 split-K slabs -- guards the scratch sizing."""
    kw = dict(seed=31, num_graphs=32, tracks=10, nodes_per_clip=10, drop_prob=0.1, empty_clip=None, far_box=False)
    m = _model(31)
    data = _data(make_head_case(**kw))
    feats = torch.randn(32, 2, 31, 64, generator=torch.Generator().manual_seed(5)).cuda()
    labels = data.bbox[:, 4]
    nh = m.native_head()
    loss, logits, valid, targets = nh.forward(feats, data.bbox, data.bbox_batch, labels, 32, 256, 256, False)
    nh.backward()
    native = {k: v.clone() for k, v in nh.G.items()}
    loss_t, logits_t, valid_t, _ = m.head(feats, data, labels)
    loss_t.backward()
    assert abs(float(loss) - float(loss_t)) < 1e-4 * max(1.0, float(loss_t))
    for k, p in m.named_parameters():
        if k in native:
            assert torch.allclose(native[k], p.grad, atol=5e-5, rtol=5e-4), k
