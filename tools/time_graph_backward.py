"""Forward and backward of the whole graph branch (NativeBackbone + graph_backward.BackboneBackward) on a synthetic
configuration: wall time per pass, and the HIP-event time of every native call of the backward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights, _lib
from eventad_amd.backbone import NativeBackbone
from eventad_amd.graph_backward import BackboneBackward
from eventad_amd.dagr.utils.buffers import format_data
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
W, H, n, B, live = synthetic.CONFIGS[cfg]
dev = torch.device("cuda:0")
data = format_data(synthetic.make_config_batch(cfg).to(dev))
sd = weights.random_graph_state()
nb = NativeBackbone(sd, W, H, dev)
feats = [m.contiguous(memory_format=torch.channels_last) for m in data.image_feats]
run_f = lambda: nb.forward(data.pos, data.x, data.batch, B, feats)
out = run_f()
bw = BackboneBackward(nb, sd)
n3, n4 = int(out["out3"]["n_dev"].item()), int(out["out4"]["n_dev"].item())
G3, G4 = torch.randn(n3, 64, device=dev), torch.randn(n4, 64, device=dev)
def wall(f, reps=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
print("%s: N = %d events, forward %.3f ms, backward %.3f ms (wall, eager launches)" % (cfg, data.pos.shape[0], wall(run_f), wall(lambda: bw.backward(G3, G4))))
# per-call HIP-event times of one backward
orig, rec = _lib.call, []
def timed(name, *a):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(name, *a); e.record(); rec.append((name, s, e))
import eventad_amd.graph_backward as gb
gb.call = timed
bw.backward(G3, G4); torch.cuda.synchronize()
gb.call = orig
tot = {}
for name, s, e in rec:
    c, t = tot.get(name, (0, 0.0)); tot[name] = (c + 1, t + s.elapsed_time(e))
for name, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-32s x%-3d %8.3f ms" % (name, c, t))
print("  sum of native calls %.3f ms" % sum(t for _, t in tot.values()))
