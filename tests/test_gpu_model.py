"""GPU parity of the whole hot path through the reference's module API: DAGR.extract_features ->
EventADModel.forward / one training step, against the CPU oracle on the same seeded input.
Tolerance (the build's stated bar): logits within 1e-3, argmax labels identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(W, H, B, n, live_cnn=False, seed=7):
    import eventad_amd.compat as compat
    compat.install()
    from config.eventad_config import parse_eventad_args
    from dagr.model.networks.dagr import DAGR
    from dagr.model.networks.ema import ModelEMA
    from dagr.utils.buffers import format_data
    from models.EventAD import EventADModel
    from eventad_amd import synthetic, weights
    from tests.helpers import seeded_head_state
    args = parse_eventad_args([])
    args.batch_size = B
    dev = torch.device("cuda:0")
    sd = weights.random_graph_state()
    if live_cnn:
        sd.update(weights.random_cnn_state())
    dagr = DAGR(args, height=H, width=W).to(dev)
    ema = ModelEMA(dagr)
    res = ema.ema.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and (live_cnn or all(k.startswith("backbone.net.") for k in res.missing_keys))
    ema.ema.cache_luts(radius=args.radius, height=H, width=W)
    model = EventADModel(ema.ema, x_dim=args.x_dim, h_dim=args.h_dim).to(dev)
    head_sd = seeded_head_state(seed)
    model.load_state_dict(head_sd, strict=False)
    raw = synthetic.make_batch(W, H, n, B, with_image=live_cnn, precomputed_feats=not live_cnn)
    data = format_data(raw.clone().to(dev))
    return args, model, sd, head_sd, raw, data


def _oracle(sd, head_sd, raw, W, H, B, live_cnn=False, grads=False, mode="gather"):
    from oracle import dagr_ref, head_ref, resnet_ref
    d = dagr_ref.format_data(raw.clone())
    feats_img = resnet_ref.hook_module_forward(sd, d.image) if live_cnn else d.image_feats
    o3, o4 = dagr_ref.backbone_forward(sd, d, dagr_ref.make_config(W, H, B), feats_img, mode=mode)
    p = {k: v.clone().requires_grad_(grads) for k, v in head_sd.items()}
    feats = head_ref.roi_features(o4, d)
    loss, outs = head_ref.head_forward(p, feats, d, d.bbox[:, 4])
    return o4, feats, loss, outs, p


@pytest.mark.parametrize("W,H,B,n", [(64, 64, 3, 3000), (256, 256, 4, 20000)])
def test_forward_logits_match_oracle(W, H, B, n):
    args, model, sd, head_sd, raw, data = _build(W, H, B, n)
    model.eval()
    with torch.no_grad():
        losses, outputs, labels = model(data, data.bbox[:, 4])
    feats_native = model.batch_feature_cache.cpu()
    o4, feats, loss, outs, _ = _oracle(sd, head_sd, raw, W, H, B)
    assert torch.allclose(feats_native, feats, atol=1e-3, rtol=1e-3), float((feats_native - feats).abs().max())
    got = torch.cat([o for frame in outputs for o in frame], 0).cpu()
    want = torch.cat([o[2] for o in outs], 0).detach()
    assert got.shape == want.shape and len(want) > 0
    assert float((got - want).abs().max()) < 1e-3
    assert torch.equal(got.argmax(1), want.argmax(1))
    assert abs(float(losses["cross_entropy"]) - float(loss)) < 1e-3 * max(1.0, float(loss))
    tg = torch.cat([t for frame in labels for t in frame], 0).cpu()
    assert torch.equal(tg, torch.cat([o[3] for o in outs], 0))


def test_train_step_grads_and_update_match_oracle():
    from eventad_amd.train_step import TrainStep
    W = H = 64; B = 3; n = 3000
    args, model, sd, head_sd, raw, data = _build(W, H, B, n)
    model.eval()                                     # dropout off so the step is comparable
    ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
    loss, norm = ts.step(data)
    ts.check()
    o4, feats, oloss, outs, p = _oracle(sd, head_sd, raw, W, H, B, grads=True)
    oloss.backward()
    used = [k for k in p if p[k].grad is not None]
    onorm = torch.sqrt(sum((p[k].grad ** 2).sum() for k in used))
    assert abs(float(loss) - float(oloss)) < 1e-3 * max(1.0, float(oloss))
    assert abs(float(norm) - float(onorm)) < 2e-3 * max(1.0, float(onorm))
    # oracle's own AdamW step on the same (clipped) grads
    params = [p[k] for k in used]
    torch.nn.utils.clip_grad_norm_(params, args.grad_clip)
    opt = torch.optim.AdamW(params, lr=args.learning_rate, weight_decay=1e-5)
    opt.step()
    mine = dict(model.named_parameters())
    for k in used:
        assert torch.allclose(mine[k].detach().cpu(), p[k].detach(), atol=2e-5, rtol=1e-4), k


def test_live_cnn_branch_matches_oracle():
    W = H = 64; B = 2; n = 2000
    args, model, sd, head_sd, raw, data = _build(W, H, B, n, live_cnn=True)
    from oracle import dagr_ref, resnet_ref
    d = dagr_ref.format_data(raw.clone())
    want = resnet_ref.hook_module_forward(sd, d.image)
    got = model.dagr_model.backbone.net(data.image)
    for g, w in zip(got, want):
        g = g.permute(0, 3, 1, 2).cpu()
        assert g.shape == w.shape
        assert float((g - w).abs().max()) < 1e-3 * max(1.0, float(w.abs().max()))
    model.eval()
    with torch.no_grad():
        losses, outputs, labels = model(data, data.bbox[:, 4])
    o4, feats, loss, outs, _ = _oracle(sd, head_sd, raw, W, H, B, live_cnn=True)
    got = torch.cat([o for frame in outputs for o in frame], 0).cpu()
    want = torch.cat([o[2] for o in outs], 0).detach()
    assert float((got - want).abs().max()) < 1e-3 and torch.equal(got.argmax(1), want.argmax(1))
    # the trunk's native weight images are built once: eval()/train() (extract_features calls eval() on every step) keep
    # them, a state-dict load or a device move drops them
    hook = model.dagr_model.backbone.net
    native = hook._native
    assert native is not None
    model.train(); model.eval()
    with torch.no_grad():
        model(data, data.bbox[:, 4])
    assert hook._native is native
    model.load_state_dict(model.state_dict())
    assert hook._native is None
    got2 = hook(data.image)
    assert all(torch.equal(a, b) for a, b in zip(got2, model.dagr_model.backbone.net(data.image)))


@pytest.mark.parametrize("W,H,B,n", [(64, 64, 1, 5000),          # the single-clip configuration: B = 1 takes the sampler's B' = max(B, 2) branch
                                     (256, 256, 2, 200000),      # 200k events per clip (dense windows, spiral-walk edge search) + the trunk at 256x256
                                     (640, 480, 1, 20000),       # tap heights 240/120/60/30/15
                                     (1280, 720, 1, 20000)])     # the DoTA geometry: tap heights 360/180/90/45/23 (odd sizes: (n+1)//2), r = 13
def test_live_cnn_configurations_match_oracle(W, H, B, n):
    args, model, sd, head_sd, raw, data = _build(W, H, B, n, live_cnn=True)
    from oracle import dagr_ref, resnet_ref
    d = dagr_ref.format_data(raw.clone())
    want = resnet_ref.hook_module_forward(sd, d.image)
    got = model.dagr_model.backbone.net(data.image)
    for g, w in zip(got, want):
        g = g.permute(0, 3, 1, 2).cpu()
        assert g.shape == w.shape
        assert float((g - w).abs().max()) < 1e-3 * max(1.0, float(w.abs().max()))
    model.eval()
    with torch.no_grad():
        losses, outputs, labels = model(data, data.bbox[:, 4])
    o4, feats, loss, outs, _ = _oracle(sd, head_sd, raw, W, H, B, live_cnn=True, mode="ondemand")
    got = torch.cat([o for frame in outputs for o in frame], 0).cpu()
    want = torch.cat([o[2] for o in outs], 0).detach()
    assert got.shape == want.shape and len(want) == 10 * B
    assert float((got - want).abs().max()) < 1e-3 and torch.equal(got.argmax(1), want.argmax(1))


def _pinned(d):
    """The batch with every tensor the step loads in pinned host memory."""
    h = d.clone()
    for k in ("pos", "x", "batch", "bbox", "bbox_batch", "bbox0", "bbox0_batch", "image"):
        v = getattr(d, k, None)
        if isinstance(v, torch.Tensor):
            setattr(h, k, v.cpu().pin_memory())
    if getattr(d, "image_feats", None) is not None:
        h.image_feats = [m.cpu().pin_memory() for m in d.image_feats]
    return h


def _run_mode(mode, ts, batches):
    """Run one optimisation step per batch in the given launch form; -> per-step losses the form can report (floats are
    read after a synchronisation: for the no-join forms only each chain's last result is still there)."""
    from eventad_amd.train_step import TrainStep
    losses = []
    if mode == "sequential":
        for b in batches:
            losses.append(ts.step(b)[0].clone())
    elif mode == "pipelined":
        ts.prime(batches[0])
        for k, b in enumerate(batches):
            losses.append(ts.pipelined_step(b, None, batches[min(k + 1, len(batches) - 1)])[0].clone())
    elif mode == "graph":
        buf = batches[0].clone()
        ts.capture(buf, warmup=2)                        # capturing leaves the weights untouched
        for b in batches:
            TrainStep.load_into(buf, b)
            losses.append(ts.replay()[0].clone())
    else:
        join = not mode.endswith("nojoin")
        if "masked" in mode:                             # force / forbid the CU partition (default: decided by timing)
            ts.head_cus = 0 if "unmasked" in mode else 32
        if "chains" in mode:                             # backbone chains (default: 3 with the CU partition, else 2)
            ts.backbone_chains = int(mode.split("chains")[0][-1])
        prefetch = int(mode.split("prefetch")[1][0]) if "prefetch" in mode else 1
        ts.capture_dual(batches[0], warmup=2, prefetch=prefetch)
        n = ts._dual["n"]
        assert ts._dual["masked"] == ("unmasked" not in mode) or "masked" not in mode
        assert n == (int(mode.split("chains")[0][-1]) if "chains" in mode else (3 if ts._dual["masked"] else 2))
        held = []
        import contextlib
        feed = batches[1:]
        if "hostbatches" in mode:                        # batches handed over in pinned host memory
            feed = [_pinned(b) for b in feed]
        # a loop that hands over batches runs under a stream of its own (work on the legacy default stream is a barrier
        # across the pipeline's CU-masked streams)
        own = "hostbatches" in mode or "callerstream" in mode
        torch.cuda.synchronize()
        with (torch.cuda.stream(ts.caller_stream()) if own else contextlib.nullcontext()):
            for i, b in enumerate(feed):
                r = ts.replay_dual(b, join=join)
                assert (r is None) == (i < prefetch - 1)     # the first prefetch - 1 calls only start backbone passes
                if r is None:
                    continue
                held.append(r[0])
                if join:
                    losses.append(r[0].clone())
            held.append(ts.flush()[0])                       # (runs the head updates of all batches still in the pipeline)
        torch.cuda.synchronize()
        if prefetch > 1:
            losses = [held[-1].clone()]                  # flush() reports the last of its head updates only
        elif join:
            losses.append(held[-1].clone())
        else:                                            # a chain's two result floats survive until its next turn, n steps on
            assert len({h.data_ptr() for h in held}) == min(n, len(held))
            losses = [h.clone() for h in held[-min(n, len(held)):]]
    torch.cuda.synchronize()
    ts.check()
    return [float(l) for l in losses]


def _assert_same_training(results):
    ref = results[0]
    steps = len(ref[1])
    assert len(set(ref[1])) == steps                          # different batches, different losses
    for i, got in enumerate(results[1:]):
        assert got[2] == ref[2] == float(steps), i            # optimiser step count
        assert got[1] == ref[1][-len(got[1]):], (i, got[1], ref[1])   # every loss the mode reported, bit for bit
        for k in ref[0]:
            assert torch.equal(got[0][k], ref[0][k]), (i, k)  # the whole path is free of float atomics: bitwise reproducible


def test_pipelined_and_graph_steps_equal_sequential_steps():
    """backbone(k+1) || head(k) on two streams, eagerly and as replayed hipGraphs, with a DIFFERENT batch every step,
    must leave exactly the weights that plain sequential steps over the same batches leave: the head has to see the
    boxes and labels of the batch that produced its features, whatever the launch form."""
    from eventad_amd import synthetic
    from eventad_amd.train_step import TrainStep
    W = H = 64; B = 3; n = 3000
    results = []
    for mode in ("sequential", "pipelined", "graph", "dual", "dual_nojoin", "dual_masked", "dual_unmasked_nojoin",
                 "dual_masked_nojoin_3chains", "dual_masked_nojoin_4chains", "dual_2chains", "dual_masked_nojoin_3chains_prefetch2",
                 "dual_masked_4chains_prefetch3", "dual_masked_nojoin_3chains_callerstream", "dual_masked_nojoin_3chains_hostbatches"):
        args, model, sd, head_sd, raw, data = _build(W, H, B, n)
        from dagr.utils.buffers import format_data
        batches = [data] + [format_data(synthetic.make_batch(W, H, n, B, first_clip=3 * k, with_image=False,
                                                              precomputed_feats=True).to("cuda:0")) for k in (1, 2, 3)]
        assert not torch.equal(batches[0].bbox, batches[1].bbox) and batches[0].bbox.shape == batches[1].bbox.shape
        model.eval()
        ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
        losses = _run_mode(mode, ts, batches)
        results.append(({k: v.clone() for k, v in ts.nh.P.items()}, losses, float(ts.nh.opt_state[0])))
        ts.close()
    _assert_same_training(results)


def test_more_than_two_chains_without_the_cu_partition_are_refused():
    """Three or more backbone passes in flight on whole-chip streams ended in a GPU memory fault on a 32 M-event batch and
    the cause is not known: the mode is refused at every size (with the CU-partitioned head it runs, see above)."""
    from eventad_amd.train_step import TrainStep
    args, model, sd, head_sd, raw, data = _build(64, 64, 2, 1000)
    model.eval()
    ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
    ts.head_cus, ts.backbone_chains = 0, 3
    with pytest.raises(RuntimeError, match="CU-partitioned"):
        ts.capture_dual(data, warmup=1)
    torch.cuda.synchronize()


@pytest.mark.skip(reason="more than two backbone chains on unpartitioned (whole-chip) streams: a run of this mode at 32 M events "
                         "once faulted the GPU; a cause that fits (mask buffers freed while their graphs wrote them) is "
                         "fixed, but the mode has not been run since and TrainStep.capture_dual still refuses it")
def test_three_and_four_unpartitioned_chains_equal_sequential_steps():
    import eventad_amd.compat as compat
    compat.install()
    from eventad_amd import synthetic
    from dagr.utils.buffers import format_data
    from eventad_amd.train_step import TrainStep
    W = H = 64; B = 3; n = 3000
    results = []
    for mode in ("sequential", "dual_unmasked_nojoin_3chains", "dual_unmasked_nojoin_4chains"):
        args, model, sd, head_sd, raw, data = _build(W, H, B, n)
        batches = [data] + [format_data(synthetic.make_batch(W, H, n, B, first_clip=3 * k, with_image=False,
                                                              precomputed_feats=True).to("cuda:0")) for k in (1, 2, 3)]
        model.eval()
        ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
        losses = _run_mode(mode, ts, batches)
        results.append(({k: v.clone() for k, v in ts.nh.P.items()}, losses, float(ts.nh.opt_state[0])))
    _assert_same_training(results)


@pytest.mark.parametrize("train_mode", [False, True])
def test_pipelined_steps_with_the_live_trunk_equal_sequential_steps(train_mode):
    """The same with the ResNet-50 trunk inside the backbone pass (its split-K scratch is written by one launch and read by
    the next: two passes in flight need a scratch each) on six different batches with different images: two chains without a
    per-step join (the bench's form for live-trunk configurations) and three chains beside the CU-partitioned head.
    train_mode: dropout active, masks drawn inside the chain's backbone graph -- the generator is consumed in step order in
    every form, so the weights still agree bit for bit."""
    import eventad_amd.compat as compat
    compat.install()
    from eventad_amd import synthetic
    from dagr.utils.buffers import format_data
    from eventad_amd.train_step import TrainStep
    W = H = 256; B = 2; n = 6000
    results = []
    for mode in ("sequential", "dual_nojoin", "dual_masked_nojoin_3chains", "dual_unmasked_nojoin", "graph",
                 "dual_masked_nojoin_3chains_prefetch2"):
        args, model, sd, head_sd, raw, data = _build(W, H, B, n, live_cnn=True)
        batches = [data] + [format_data(synthetic.make_batch(W, H, n, B, first_clip=B * k, with_image=True).to("cuda:0"))
                            for k in range(1, 6)]
        assert not torch.equal(batches[0].image, batches[1].image)
        model.train(train_mode)
        torch.manual_seed(11); torch.cuda.manual_seed(11)
        ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
        losses = _run_mode(mode, ts, batches)
        results.append(({k: v.clone() for k, v in ts.nh.P.items()}, losses, float(ts.nh.opt_state[0])))
        ts.close()
    _assert_same_training(results)


def test_earlier_capture_survives_a_workspace_reallocation():
    """A captured graph holds raw pointers into the native workspaces.  Capturing a larger batch afterwards makes the
    event-level workspace grow; the superseded one must stay alive so that the first graph still replays correctly."""
    from dagr.utils.buffers import format_data
    from eventad_amd import synthetic
    from eventad_amd.train_step import TrainStep
    W = H = 64; B = 2
    args, model, sd, head_sd, raw, small = _build(W, H, B, 2000)
    big = format_data(synthetic.make_batch(W, H, 6000, B, with_image=False, precomputed_feats=True).to("cuda:0"))
    model.eval()
    ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
    g1 = ts.capture(small, warmup=1)
    loss1, norm1 = ts._g_loss, ts._g_norm
    nb = model.dagr_model.backbone._native
    ws_before = nb.ws
    ts.capture(big, warmup=1)                                 # N grows: new workspace
    assert nb.ws is not ws_before and any(w is ws_before for w in nb._retired_ws)
    junk = [torch.full((1 << 22,), 7.0, device="cuda") for _ in range(8)]   # would land in freed blocks, were they freed
    snap = ts.snapshot()
    g1.replay()
    torch.cuda.synchronize()
    got = (float(loss1), float(norm1), {k: v.clone() for k, v in ts.nh.P.items()})
    ts.restore(snap)
    loss, norm = ts.step(small)
    torch.cuda.synchronize()
    assert (float(loss), float(norm)) == got[:2]
    for k, v in ts.nh.P.items():
        assert torch.equal(v, got[2][k]), k
    del junk


def test_two_graph_pipeline_restarts_after_a_flush_and_closes_its_streams():
    """flush() drains the pipeline; the next replay_dual(batch) only starts the backbone of `batch` (returns None), the
    call after it runs that batch's head update.  Two epochs of two batches == four sequential steps, bit for bit.
    close() destroys the CU-masked streams and may be called twice."""
    from eventad_amd import synthetic
    from eventad_amd.train_step import TrainStep
    from dagr.utils.buffers import format_data
    W = H = 64; B = 3; n = 3000
    out = []
    for mode in ("sequential", "dual"):
        args, model, sd, head_sd, raw, data = _build(W, H, B, n)
        other = format_data(synthetic.make_batch(W, H, n, B, first_clip=3, with_image=False, precomputed_feats=True).to("cuda:0"))
        model.eval()
        ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
        if mode == "sequential":
            for b in (data, other, data, other):
                ts.step(b)
        else:
            ts.head_cus = 32
            ts.capture_dual(data, warmup=1)
            ts.replay_dual(other)                            # head(data) || backbone(other)
            ts.flush()                                       # head(other)
            assert ts.replay_dual(data) is None              # epoch 2: backbone(data) only
            ts.replay_dual(other)                            # head(data) || backbone(other)
            ts.flush()
            ts.close(); ts.close()
        torch.cuda.synchronize()
        out.append(({k: v.clone() for k, v in ts.nh.P.items()}, float(ts.nh.opt_state[0])))
    assert out[0][1] == out[1][1] == 4.0
    for k in out[0][0]:
        assert torch.equal(out[0][0][k], out[1][0][k]), k


def test_captured_evaluation_forward_equals_the_eager_one():
    """InferStep: the evaluation forward as one replayed hipGraph, with the live trunk, fed three different batches of one
    shape -- logits, validity and targets bitwise equal to the eager predict_tensors of the same batches."""
    from eventad_amd.infer_step import InferStep
    from eventad_amd.dagr.utils.buffers import format_data
    W = H = 64
    B, n = 3, 4000
    from eventad_amd import synthetic
    args, model, sd, head_sd, raw, data = _build(W, H, B, n, live_cnn=True)
    model.eval()
    batches = [format_data(synthetic.make_batch(W, H, n, B, first_clip=10 * i, with_image=True).to("cuda")) for i in range(4)]
    with torch.no_grad():
        eager = [[t.clone() for t in model.predict_tensors(d, d.bbox[:, 4])] for d in batches]
    st = InferStep(model)
    st.capture(batches[0])
    for i in (1, 2, 3, 0, 2):
        out = st.replay(batches[i])
        torch.cuda.synchronize()
        for a, b in zip(out, eager[i]):
            assert torch.equal(a, b), i
    model.train()
    with pytest.raises(RuntimeError):
        st.replay()

