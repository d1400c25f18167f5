"""The pipelined C2 step fed a DIFFERENT batch every step: (a) batches resident on the device (the one-launch load into the
chain's static buffers is then inside the step), (b) formatted batches in pinned host memory (the host-to-device copies run on the
chain's backbone stream, in front of its graph).  bench.py's `value` replays the batch each chain holds; these are the rates a
training loop sees.  usage: python tools/exp_host_batches.py [steps=300] [distinct_batches=6]"""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from eventad_amd import synthetic
from eventad_amd.train_step import TrainStep
from eventad_amd.dagr.utils.buffers import format_data
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, n, B, live = synthetic.CONFIGS["C2"]
dev = torch.device("cuda:0")
args, model, sd = bench.build_model(W, H, B, dev, live_cnn=live)
model.train()
def make(i):
    d = format_data(synthetic.make_config_batch("C2", first_clip=i * B).to(dev))
    d.image_feats = [m.contiguous(memory_format=torch.channels_last) for m in d.image_feats]
    return d
pool = [make(i) for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6)]
ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
ts.capture_dual(pool[0].clone(), prefetch="max")
def pinned(d):
    h = d.clone()
    for k in ("pos", "x", "batch", "bbox", "bbox_batch", "bbox0", "bbox0_batch"):
        setattr(h, k, getattr(d, k).cpu().pin_memory())
    # (channels-last maps: keep the memory format, the static buffers have it)
    h.image_feats = [torch.empty_like(m, device="cpu", memory_format=torch.preserve_format).pin_memory().copy_(m) for m in d.image_feats]
    return h
host_pool = [pinned(d) for d in pool]
nbytes = sum(getattr(pool[0], k).numel() * getattr(pool[0], k).element_size() for k in ("pos", "x", "batch", "bbox", "bbox_batch", "bbox0", "bbox0_batch")) \
    + sum(m.numel() * 4 for m in pool[0].image_feats)
out = {"batch_bytes": nbytes}
side = ts.caller_stream()                                     # a non-blocking stream for the caller's loop, see below
import contextlib
for name, src, ctx in (("same_batch_replayed", None, None), ("device_batches_default_stream", pool, None), ("device_batches", pool, side),
                       ("pinned_host_batches_default_stream", host_pool, None), ("pinned_host_batches", host_pool, side)):
    # The CU-masked streams of the pipeline are BLOCKING streams (hipExtStreamCreateWithCUMask has no flags argument): any
    # operation on the legacy default stream -- the event that makes a chain wait for the caller's stream included -- is a
    # barrier across all of them.  A caller that hands over batches runs its loop under a stream of its own.
    with (torch.cuda.stream(ctx) if ctx is not None else contextlib.nullcontext()):
        for i in range(40):
            ts.replay_dual(None if src is None else src[i % len(src)], join=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ts.replay_dual(None if src is None else src[i % len(src)], join=False)
        t_issue = time.perf_counter() - t0                     # host time to enqueue the steps (the GPU may still be behind)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out[name] = {"clips_per_s": round(B * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4), "host_issue_ms_per_step": round(1e3 * t_issue / steps, 4)}
ts.flush(); ts.check()
out["host_to_device_GBps_needed_at_device_rate"] = round(nbytes / (out["device_batches"]["ms_per_step"] * 1e-3) / 1e9, 1)
print(json.dumps(out), flush=True)
del host_pool, pool
torch.cuda.synchronize()
ts.close()
