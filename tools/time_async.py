"""Latency of an asynchronous update (NativeBackbone.append_events) against a dense pass.
usage: time_async.py [W H events_per_clip B]   (default: one 256x256 clip of 50k events)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
W, H, n, B = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (256, 256, 50000, 1)
dev = torch.device("cuda:0")
data = format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True).to(dev))
N = data.pos.shape[0]
nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
nb.async_capacity = 4096
E = lambda: torch.cuda.Event(enable_timing=True)


def timed(fn, reps):
    torch.cuda.synchronize()
    a, b = E(), E()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


maps = data.image_feats
dense = lambda m: nb.forward(data.pos[:m], data.x[:m], data.batch[:m], B, maps)
for _ in range(3):
    dense(N)
print("dense pass on %d events: %.3f ms" % (N, timed(lambda: dense(N), 20)))
# the last events of the LAST clip arrive one update at a time (newer than everything of their clip so far)
for k in (1, 10, 100, 1000):
    n0 = N - 4000
    dense(n0)
    upd = lambda i: nb.append_events(data.pos[n0 + i * k: n0 + (i + 1) * k], data.x[n0 + i * k: n0 + (i + 1) * k],
                                     data.batch[n0 + i * k: n0 + (i + 1) * k])
    upd(0)
    reps = min(20, 4000 // k - 1)
    torch.cuda.synchronize()
    a, b = E(), E()
    a.record()
    for i in range(1, reps + 1):
        upd(i)
    b.record()
    torch.cuda.synchronize()
    print("update of %4d events: %.3f ms (%.1f us per event)" % (k, a.elapsed_time(b) / reps, 1e3 * a.elapsed_time(b) / reps / k))
nb.timing = {}
dense(N - 4000)
nb.timing = {}
nb.append_events(data.pos[N - 4000: N - 3900], data.x[N - 4000: N - 3900], data.batch[N - 4000: N - 3900])
for name, (cnt, ms) in sorted(nb.timing_summary().items(), key=lambda kv: -kv[1][1]):
    print("    %-36s %.1f us" % (name, ms * 1e3))
