"""bench.py -- EventAD hot path on MI355X: clips/sec of one optimisation step (forward of the frozen
DAGR backbone + ROI pooling + head forward/backward + grad-clip + AdamW, = reference
utils/train.py:100-152) on the synthetic ROL-shaped configuration the headline metric is quoted on
(C2: 256x256, 50k events/clip, precomputed CNN features, 8 clips per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant event-level
kernel, HIP-event timed inside the timed region) and, at N=1, `cpu_baseline` (the CPU oracle timed on
the host cores on the same batch)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENGINE_CLOCK_GHZ = 2.4         # MI355X peak engine clock (MI355X_MICROARCH.md)
LDS_B128_BYTES_PER_CLK = 256   # per CU, ds_read_b128 (MI355X_MICROARCH.md, LDS table)
VALU_FMA_PER_CLK = 64          # per CU: 4 SIMDs x 16 lanes, one fp32 FMA per lane and clock (packed fp32 is off in this build)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy peak 6290 GB/s
# C-ABI call -> prefixes of the kernel names it launches, as they appear in the rocprofv3 traces / profiles/*_hbm_traffic.json
# (the template arguments depend on the configuration: row scan or spiral walk, LUT fully or partly in LDS)
KERNEL_OF = {"eventad_build_ell#1": ("k_build_ell_rowscan", "k_build_ell_reg", "k_build_ell_pix"), "eventad_l1_conv1#1": ("k_l1_conv<20",),
             "eventad_l1_conv2#1": ("k_l1_conv<16",), "eventad_pool_seg0#1": ("k_pool_seg0(", "k_pool_seg0")}


def pmc_traffic(config, prefixes):
    """HBM bytes per launch from the committed PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    rocprofv3 --pmc runs, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_traffic.py)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_hbm_traffic.json" % config.lower())), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        hits = [k for k in d if any(k.startswith(p) for p in prefixes) and not k.startswith("k_pool_seg0_bwd")]
        if hits:
            return max(d[k]["hbm_bytes_per_launch"] for k in hits), os.path.basename(f)
    return None, None


def build_model(W, H, B, dev, seed=42, live_cnn=False):
    import eventad_amd.compat as compat
    compat.install()
    from config.eventad_config import parse_eventad_args
    from dagr.model.networks.dagr import DAGR
    from dagr.model.networks.ema import ModelEMA
    from models.EventAD import EventADModel
    from eventad_amd import weights
    args = parse_eventad_args([])
    args.batch_size = B
    torch.manual_seed(seed)
    dagr = DAGR(args, height=H, width=W).to(dev)
    ema = ModelEMA(dagr)
    sd = weights.random_graph_state(seed)
    if live_cnn:
        sd.update(weights.random_cnn_state(seed + 1))
    ema.ema.load_state_dict(sd, strict=False)
    ema.ema.cache_luts(radius=args.radius, height=H, width=W)
    model = EventADModel(ema.ema, x_dim=args.x_dim, h_dim=args.h_dim).to(dev)
    return args, model, sd


def algorithmic_bytes(N, E, B, W, H):
    """Per-stage algorithmic bytes (what the stage must read and write once), keyed by the C-ABI call that implements the stage."""
    maps1 = 4 * 64 * B * (H // 4) * (W // 4)
    return {
        "eventad_build_ell#1": 16 * N + 8 * E,                                   # S1
        "eventad_l1_conv1#1": E * (76 + 8) + N * (76 + 64),                      # S3
        "eventad_l1_conv2#1": E * (64 + 8) + N * (64 + 76 + 64),                 # S4
        # S5 + S6, plus the pooled-edge marks the kernel now does: the ELL row + degree per event, one voxel id per edge
        "eventad_pool_seg0#1": 256 * N + maps1 + N * (320 + 16) + B * 2240 * (320 + 12) + 68 * N + 4 * E,
    }


def cpu_baseline(raw, sd, head_sd, W, H, B, steps=10, warmup=3):
    """The CPU oracle (oracle/, kind "port") running the same optimisation step on the host cores."""
    from oracle import dagr_ref, head_ref
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                          # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    t_setup = time.perf_counter()
    cfg = dagr_ref.make_config(W, H, B)
    luts = dagr_ref.cache_luts(sd, cfg)                       # one-time model set-up in the reference (dagr.py:36-71)
    p = {k: v.clone().requires_grad_(True) for k, v in head_sd.items()}
    params = list(p.values())
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-5)
    d = dagr_ref.format_data(raw.clone())
    times = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        with torch.no_grad():
            _, o4 = dagr_ref.backbone_forward(sd, d, cfg, d.image_feats, mode="grouped", luts=luts)
            feats = head_ref.roi_features(o4, d)
        loss, _ = head_ref.head_forward(p, feats, d, d.bbox[:, 4])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        el = time.perf_counter() - t0
        print("[cpu_baseline] step %d: %.2f s (set-up so far %.1f s)" % (it, el, t0 - t_setup), file=sys.stderr, flush=True)
        if it >= warmup:
            times.append(el)
        elif el > 40.0:                                       # very slow host: keep the default run within minutes
            times.append(el)
            break
    sec = sum(times) / len(times)
    return {"value": round(B / sec, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "same C2 batch (%d clips x 50k events), %d timed step(s) after %d warm-up, LUTs cached at set-up, "
                      "torch CPU threads=%d, %.2f s/step" % (B, steps, warmup, cores, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap the frozen backbone of batch k+1 with the head update of batch k")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct device-resident batches handed to the step in rotation, a different one every step "
                         "(1 = every step re-runs the batch the step object already holds: no batch-load launch in the step)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from eventad_amd import synthetic
    from eventad_amd.train_step import TrainStep
    W, H, n_ev, B, live = synthetic.CONFIGS[a.config]
    args, model, sd = build_model(W, H, B, dev, live_cnn=live)
    from dagr.utils.buffers import format_data
    R = max(1, a.rotate)

    def resident(i, keep_raw=False):
        r = synthetic.make_config_batch(a.config, first_clip=(rank * R + i) * B)        # each rank owns different clips
        d = format_data(r.clone().to(dev))
        if getattr(d, "image_feats", None) is not None:
            # precomputed CNN feature maps are kept in channels-last memory format ([B,C,h,w] tensors, NHWC in memory): the
            # layout every sampler kernel reads, so the step has no layout-change launch (plain NCHW maps are accepted too)
            d.image_feats = [m.contiguous(memory_format=torch.channels_last) for m in d.image_feats]
        return (r, d) if keep_raw else d

    raw, data = resident(0, keep_raw=True)
    # the batches of the timed region: R distinct ones, all resident in HBM before the timer starts, a different one every step
    pool = [data] + [resident(i) for i in range(1, R)]
    model.train()                                                           # as utils/train.py:90 (dropout active)
    ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
    nb = model.dagr_model.backbone._native

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not a.no_graph and os.environ.get("EVENTAD_NO_GRAPH") != "1"
    pipelined = use_graph and not a.no_pipeline
    for _ in range(a.warmup):
        ts.step(data)
    if use_graph:
        try:
            if pipelined:
                # backbone graphs || head graph on event-coupled streams; the backbone runs as far ahead of the head as the
                # chains allow, so the timed region starts with that many finished passes in hand (as it ends with them)
                ts.capture_dual(data.clone(), prefetch="max")
                while ts._dual["issued"] - ts._dual["k"] < ts._dual["prefetch"]:
                    ts.replay_dual(join=False)                 # fill: these calls only start backbone passes
                # (the timed region ends with a device-wide sync)
                run = (lambda k: ts.replay_dual(join=False)) if R == 1 else (lambda k: ts.replay_dual(pool[k % R], join=False))
            else:
                held = data.clone()
                ts.capture(held)
                def run(k):
                    if R > 1:
                        TrainStep.load_into(held, pool[k % R])
                    return ts.replay()
        except Exception as e:
            if world > 1:
                # the multi-GPU number is defined on the captured pipeline (all-reduce inside the head graph): an eager run
                # in its place would be a different measurement under the same name
                print("[bench] rank %d: hipGraph capture failed (%s: %s)" % (rank, type(e).__name__, e), file=sys.stderr, flush=True)
                raise SystemExit(3)
            print("[bench] hipGraph capture failed (%s: %s); falling back to eager launches" % (type(e).__name__, e), file=sys.stderr)
            torch.cuda.synchronize()
            use_graph = pipelined = False
    if not use_graph:
        run = lambda k: ts.step(pool[k % R])
        if not a.no_kernel_timing:
            nb.timing = {}
    # untimed: the replayed graphs' first runs (cold caches, pipeline fill), and enough of them that the timed region does not
    # start on a chip that idled through the capture (the clocks take some milliseconds of load to come up: without this a
    # 20-step region reads 3-4 % slower than a 300-step one)
    n_warm = int(os.environ.get("EVENTAD_BENCH_GRAPH_WARMUP", "40"))
    # The loop that hands batches to the pipelined step runs under a stream of its own: the pipeline's CU-masked streams are
    # blocking streams, and anything on the legacy default stream -- the event that makes a chain wait for the stream its next
    # batch was produced on included -- would be a barrier across all of them.
    import contextlib
    loop_stream = (lambda: torch.cuda.stream(ts.caller_stream())) if (pipelined and R > 1) else contextlib.nullcontext

    def timed(fn, steps):
        with loop_stream():
            barrier()
            t_ = time.perf_counter()
            out_ = None
            for k in range(steps):
                out_ = fn(k)
            barrier()
            return time.perf_counter() - t_, out_

    with loop_stream():
        for k in range(n_warm if use_graph else 0):
            run(k)
    dt, (loss, norm) = timed(run, a.steps)
    held_rate = None
    if pipelined and R > 1 and world == 1:
        # beside it: the same steps on the batch each chain holds (no batch-load launch in the step)
        for k in range(10):
            ts.replay_dual(join=False)
        dt_h, _ = timed(lambda k: ts.replay_dual(join=False), a.steps)
        held_rate = round(B * a.steps / dt_h, 1)
    if use_graph and not a.no_kernel_timing:
        # per-kernel HIP-event timing of the same K steps, launched eagerly on the same stream (a graph
        # replay cannot be bracketed per kernel); the launch durations are those of the timed region.
        nb.timing = {}
        for _ in range(a.steps):
            ts.step(data)
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ts.check()
    nb.check_flags()

    if rank == 0:
        N = int(data.pos.shape[0])
        E = int(nb.ws["deg"].sum().item())
        out = {
            "metric": "clips/sec (fwd+bwd)", "value": round(world * B * a.steps / dt, 3), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # (kept under 120 characters: longer strings are cut by the harness that records the line)
            "config": {"workload": "%s: ROL-shape %dx%d, %d events/clip + %s, batch %d/GPU, fwd + head bwd + clip + AdamW"
                                   % (a.config, W, H, n_ev, "live ResNet-50 branch" if live else "precomputed CNN maps", B),
                       "global_batch": world * B, "events_per_batch_per_gpu": N, "realised_edges_per_gpu": E,
                       "parallelism": "dp%d" % world, "launch": "hipGraph replay" if use_graph else "eager",
                       "untimed_graph_replays": n_warm if use_graph else 0,
                       "pipeline": ("none" if not pipelined else "backbone(k+1..k+%d) || head update(k): hipGraphs on event-coupled streams, %d backbone chains (own workspace set and stream each), backbone %d batch(es) ahead, head updates in order on one stream" % (ts._dual["n"] - 1, ts._dual["n"], ts._dual["prefetch"])),
                         "cu_partition": ("head graph on %d CUs of its own, backbone graph on the others (CU-masked default-priority streams)" % ((ts.head_cus if ts.head_cus > 0 else torch.cuda.get_device_properties(0).multi_processor_count // 8)) if (pipelined and ts._dual and ts._dual.get("masked")) else "none"),
                       "input": ("%d distinct device-resident batches in rotation, a different one every step (its one-launch load into the step's static buffers is inside the timed region); H2D copy excluded" % R if R > 1
                                 else "device-resident batch, H2D copy excluded; every step re-runs the full backbone + head update on the batch the step holds")
                                + "; precomputed feature maps stored channels-last (NHWC in memory)",
                       "clips_per_s_on_the_held_batch": held_rate,
                       "skipped_vs_reference": "ROI pooling of the previous-frame boxes (computed by the reference, never read by its forward)",
                       "grad_allreduce_bytes": ts.bucket_bytes if world > 1 else 0,
                       "final_loss": round(float(loss), 4)},
        }
        if nb.timing:
            summ = nb.timing_summary()
            ab = algorithmic_bytes(N, E, B, W, H)
            name = max(ab, key=lambda k: summ[k][1])
            ms = summ[name][1]
            ach = ab[name] / (ms * 1e-3) / 1e9
            traffic, src = pmc_traffic(a.config, KERNEL_OF.get(name, ()))
            out["roofline"] = {"bound": "hbm", "kernel": name.split("#")[0], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                               "algorithmic_bytes": ab[name], "avg_launch_ms": round(ms, 4)}
            cin = {"eventad_l1_conv1#1": 20, "eventad_l1_conv2#1": 16}.get(name)
            if cin:
                # What holds the layer-1 convolutions is neither HBM nor LDS but the vector ALU: an edge multiplies its source
                # row by ITS cell's Cin x 16 matrix, Cin*16 fp32 FMAs that no other edge shares (v_fma_f32, one per lane and
                # clock: this build does not use the packed form).  The same edge reads the matrix from LDS once, Cin
                # ds_read_b128 per lane of its quad = Cin*64 bytes, against 256 B per clock and CU.
                cus = torch.cuda.get_device_properties(0).multi_processor_count
                peak_tf = 2.0 * VALU_FMA_PER_CLK * cus * ENGINE_CLOCK_GHZ / 1e3
                tf = 2.0 * cin * 16 * E / (ms * 1e-3) / 1e12
                out["roofline"]["valu"] = {"flop_per_edge": 2 * cin * 16, "achieved": round(tf, 2), "peak": round(peak_tf, 1),
                                           "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4),
                                           "peak_note": "non-packed v_fma_f32; the spec's 157.3 TFLOP/s counts v_pk_fma_f32"}
                peak = float(LDS_B128_BYTES_PER_CLK) * cus * ENGINE_CLOCK_GHZ
                lds = cin * 64 * E / (ms * 1e-3) / 1e9
                out["roofline"]["lds"] = {"bytes_per_edge": cin * 64, "achieved": round(lds, 1), "peak": round(peak, 1),
                                          "unit": "GB/s", "frac": round(lds / peak, 4)}
            out["stage_ms"] = {k.replace("eventad_", ""): round(v[1], 4) for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])[:12]}
            out["backbone_ms"] = round(sum(v[1] for v in summ.values()), 4)
        if world == 1 and not a.no_cpu_baseline and (live or a.config != "C2"):
            # the oracle leg is sized for the default workload (C2, precomputed CNN features): a live-trunk step or a
            # 32 M-event batch takes minutes per step on the host cores
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = "timed on the default workload only (C2); this configuration's CPU step takes minutes"
        elif world == 1 and not a.no_cpu_baseline:
            head_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()
                       if not k.startswith("dagr_model.") and not k.startswith("fusion_module.attention.")}
            out["cpu_baseline"] = cpu_baseline(raw, sd, head_sd, W, H, B)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
