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


# launch form of the clip recurrence: "stage" = one fused launch per clip stage forward and backward (the default),
# "stage_fwd" = fused forward, three launches per stage backward, "launches" = two / three launches per stage
FORMS = {"stage": (True, True), "stage_fwd": (True, False), "launches": (False, False)}


@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("name", CASES)
def test_native_head_matches_reference_golden_and_autograd(name, form, golden_dir):
    gold = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    kw = gold["kw"]
    m = _model(kw["seed"])
    data = _data(make_head_case(**kw))
    feats = gold["feats"].cuda()
    labels = data.bbox[:, 4]
    nh = m.native_head()
    nh.stage_fwd, nh.stage_bwd = FORMS[form]
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


@pytest.mark.parametrize("entry", ["eventad_gemm_group", "eventad_gemm_group64"])
def test_grouped_tile_gemm(entry):
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
    call(entry, len(probs), ctypes.cast(d, ctypes.c_void_p), torch.cuda.current_stream().cuda_stream)
    for pr, ref, shp in zip(probs, refs, shapes):
        assert torch.allclose(pr[3], ref, atol=2e-4, rtol=1e-4), (shp, float((pr[3] - ref).abs().max()))


def test_launch_forms_agree_with_dropout_active():
    """Training mode (GRU inter-layer dropout + fusion dropout, same seeded masks): the fused stage kernels and the
    separate launches give the same loss and gradients up to the order of the sums."""
    kw = dict(seed=17, num_graphs=8, tracks=10, nodes_per_clip=35, drop_prob=0.2, empty_clip=3, far_box=True)
    m = _model(17).train()
    data = _data(make_head_case(**kw))
    feats = torch.randn(8, 2, 31, 64, generator=torch.Generator().manual_seed(6)).cuda()
    feats[3] = 0                                                  # a clip without any valid track: its state passes through
    labels = data.bbox[:, 4]
    nh = m.native_head()
    out = {}
    for form, (f, b) in FORMS.items():
        nh.stage_fwd, nh.stage_bwd = f, b
        nh.mask_step = 123                                        # the masks are a function of (seed, step): the same in every form
        loss, logits, valid, _ = nh.forward(feats, data.bbox, data.bbox_batch, labels, 8, 256, 256, True)
        nh.backward()
        torch.cuda.synchronize()
        out[form] = (float(loss), logits.clone(), valid.clone(), {k: v.clone() for k, v in nh.G.items()})
    ref = out["launches"]
    assert int(ref[2].sum()) > 20 and ref[0] > 0
    for form in ("stage", "stage_fwd"):
        got = out[form]
        assert abs(got[0] - ref[0]) < 1e-5 * max(1.0, ref[0]) and torch.equal(got[2], ref[2])
        assert torch.allclose(got[1], ref[1], atol=1e-5, rtol=1e-5)
        for k in ref[3]:
            assert torch.allclose(got[3][k], ref[3][k], atol=2e-5, rtol=2e-4), (form, k)


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


@pytest.mark.parametrize("name", CASES)
def test_roi_pool_kernel_matches_reference_golden_features(name, golden_dir):
    """eventad_roi_pool on the golden cases' out4 nodes == the batch_features the reference's _process_bbox loop
    produced (models/EventAD.py:382-397, 406-499): boxes with nodes inside (inverse-distance weights), a far box with
    none (mean of the 5 nearest), a clip without nodes (zero global feature), dropped rows, 30 tracks per clip."""
    gold = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    kw = gold["kw"]
    case = make_head_case(**kw)
    m = _model(kw["seed"])
    data = _data(case)
    n = case["out4_x"].shape[0]
    pad = 7                                                     # padded rows past n_dev must be ignored
    out4 = SimpleNamespace(
        x_padded=torch.cat([case["out4_x"], torch.full((pad, 64), 1e9)]).cuda().contiguous(),
        pos_padded=torch.cat([case["out4_pos"], torch.full((pad, 3), 0.5)]).cuda().contiguous(),
        batch_padded=torch.cat([case["out4_batch"], torch.zeros(pad, dtype=torch.long)]).int().cuda().contiguous(),
        n_dev=torch.tensor([n], dtype=torch.int32).cuda())
    feats = m.roi_features(out4, data, frames=(0, 1)).cpu()
    want = gold["feats"]
    assert feats.shape == want.shape
    assert torch.equal(feats == 0, want == 0)                   # same (clip, frame, track) slots filled
    assert torch.allclose(feats, want, atol=2e-5, rtol=1e-4), float((feats - want).abs().max())


def test_sliding_window_graph_mirror_and_ev_graph_cuda_module():
    """The reference-compatible graph builder as the reference drives it: `SlidingWindowGraph.forward/reset`
    (src/dagr/graph/ev_graph.py:45-103) and the pybind-shaped `ev_graph_cuda` module, both over the C ABI, against the
    CPU restatement -- two batches with a reset between them, then two batches WITHOUT a reset (growing ids)."""
    import eventad_amd.compat as compat
    compat.install()
    import ev_graph_cuda
    from dagr.graph.ev_graph import SlidingWindowGraph
    from eventad_amd import synthetic
    from oracle import graph as G
    W = H = 48; B = 2; Q = 16; K = 16; r = 2; dt = 4000
    mine = SlidingWindowGraph(width=W, height=H, batch_size=B, max_num_neighbors=K, max_queue_size=Q, radius=r, delta_t_us=dt)
    ref = G.SlidingWindowGraphRef(W, H, B, K, Q, r, dt)

    def batch(first, n):
        d = synthetic.make_batch(W, H, n, B, first_clip=first, with_image=False, t_span_us=20000)
        pos_i = torch.stack([d.pos[:, 0].int(), d.pos[:, 1].int(), d.t.int()], 1).contiguous()
        return d.batch.int().contiguous(), pos_i

    for step, (first, n, reset) in enumerate(((0, 3000, True), (2, 2500, True), (4, 2000, False))):
        b, p = batch(first, n)
        if reset:
            mine.reset(); ref.reset()
        got = mine.forward(b.cuda(), p.cuda(), delete_nodes=False, collect_edges=True)
        want = ref.forward(b, p)
        assert torch.equal(got.cpu(), want), step
        assert torch.equal(mine.event_queue.cpu(), ref.event_queue), step
    # the raw module functions with the pybind signatures (ev_graph.cu:279-283)
    b, p = batch(6, 1500)
    ref.reset()
    want = ref.forward(b, p)
    N = len(b)
    queue = torch.full((B, Q, H, W), -1, dtype=torch.int32, device="cuda")
    lin = p[:, 0] + W * p[:, 1] + W * H * b
    s_lin, s_idx = torch.sort(lin, stable=True)
    uniq, counts = torch.unique_consecutive(s_lin, return_counts=True)
    idx = torch.arange(N, dtype=torch.int32)
    q2 = ev_graph_cuda.insert_in_queue_cuda(idx[s_idx].contiguous().cuda(), uniq.int().cuda(), torch.cumsum(counts, 0).int().cuda(), queue)
    assert q2 is queue and torch.equal(queue.cpu(), ref.event_queue)
    edges = torch.full((2, K * N), -1, dtype=torch.int64, device="cuda")
    ev_graph_cuda.fill_edges_cuda(b.cuda(), p.cuda(), p[:, 2].contiguous().cuda(), queue, idx.cuda(), K, float(r), float(dt), edges, 0)
    e = edges.cpu()
    assert torch.equal(e[:, e[1] >= 0], want)
    with pytest.raises(RuntimeError):                          # AT_ASSERTM equivalents (ev_graph.cu:9-12)
        ev_graph_cuda.fill_edges_cuda(b, p.cuda(), p[:, 2].contiguous().cuda(), queue, idx.cuda(), K, float(r), float(dt), edges, 0)


def test_optimiser_skips_non_finite_and_all_zero_gradients():
    """The reference raises on a non-finite gradient before optimizer.step() (utils/train.py:144-149) and AdamW skips
    parameters whose grad is None (a batch without valid objects): in both cases weights, moments and the step count
    must stay exactly as they were; the non-finite case is counted for the host's lazy check."""
    m = _model(12)
    nh = m.native_head(lr=1e-3, weight_decay=1e-5, grad_clip=1.0)
    g = torch.Generator().manual_seed(3)
    nh.grads.copy_(torch.randn(nh.n, generator=g).cuda())
    nh.optimizer_step()                                          # one ordinary step so the moments are non-zero
    torch.cuda.synchronize()
    before = [t.clone() for t in (nh.flat, nh.exp_avg, nh.exp_avg_sq)]
    assert float(nh.opt_state[0]) == 1.0 and float(nh.opt_state[6]) == 0.0
    for poison, counted in ((float("nan"), 1), (float("inf"), 2), (None, 2)):
        nh.grads.zero_()
        if poison is not None:
            nh.grads[12345] = poison
        nh.optimizer_step()
        torch.cuda.synchronize()
        for a, b in zip((nh.flat, nh.exp_avg, nh.exp_avg_sq), before):
            assert torch.equal(a, b)
        assert float(nh.opt_state[0]) == 1.0 and float(nh.opt_state[6]) == 1.0 and nh.nonfinite_steps() == counted
    nh.grads.copy_(torch.randn(nh.n, generator=g).cuda())        # and the next ordinary step goes through
    nh.optimizer_step()
    torch.cuda.synchronize()
    assert float(nh.opt_state[0]) == 2.0 and float(nh.opt_state[6]) == 0.0 and not torch.equal(nh.flat, before[0])


def test_dropout_masks_are_a_function_of_seed_step_and_index():
    """eventad_dropout_masks: values in {0, 1/keep}, keep rate as asked, a step's masks are reproducible whether the step
    number comes as an argument or from a device word, different steps and seeds give different masks, and a prefix of a
    longer draw equals the shorter draw (value i depends on i only)."""
    import ctypes
    from eventad_amd._lib import call, ptr
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    n = 2 * 8 * 32 * 256

    def draw(n, seed, step, step_dev=None):
        out = torch.full((n + 4,), -1.0, device=dev)
        call("eventad_dropout_masks", ptr(out), n, 0.7, seed, step, ptr(step_dev), st)
        assert bool((out[n:] == -1).all())
        return out[:n]

    a = draw(n, 42, 5)
    vals = torch.unique(a)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / 0.7) < 1e-6
    assert abs(float((a > 0).float().mean()) - 0.7) < 5e-3
    assert torch.equal(a, draw(n, 42, 5))
    assert torch.equal(a, draw(n, 42, 0, torch.tensor([5], dtype=torch.int64, device=dev)))
    assert torch.equal(a[:1001], draw(1001, 42, 5))
    for other in (draw(n, 42, 6), draw(n, 43, 5)):
        agree = float((other == a).float().mean())
        assert abs(agree - (0.7 * 0.7 + 0.3 * 0.3)) < 5e-3      # independent masks agree on p^2 + (1-p)^2 of the entries
    # neighbouring entries are uncorrelated
    k = (a > 0).float()
    assert abs(float((k[:-1] * k[1:]).mean()) - 0.49) < 5e-3
