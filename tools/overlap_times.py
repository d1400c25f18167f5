"""How long do the backbone graph and the head graph take when they overlap (two-graph pipeline), and alone?
HIP events on each stream around its graph replay, averaged over steady-state steps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from eventad_amd import synthetic
from eventad_amd.train_step import TrainStep
from eventad_amd.dagr.utils.buffers import format_data
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
W, H, n, B, live = synthetic.CONFIGS[cfg]
dev = torch.device("cuda:0")
args, model, sd = bench.build_model(W, H, B, dev, live_cnn=live)
model.train()
data = format_data(synthetic.make_config_batch(cfg).to(dev))
ts = TrainStep(model, lr=args.lr, weight_decay=args.weight_decay)
ts.capture_dual(data)
d = ts._dual
sg, sh = ts._streams(d["masked"])
E = lambda: torch.cuda.Event(enable_timing=True)
def timed(stream, graph, reps=30):
    a, b = E(), E()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        a.record(stream)
        for _ in range(reps):
            graph.replay()
        b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("alone: backbone graph %.3f ms, head graph %.3f ms" % (timed(sg, d["gb"][0]), timed(sh, d["gh"][0])))
torch.cuda.synchronize()
ts.capture_dual(data)
d = ts._dual
for _ in range(10):
    ts.replay_dual(join=False)
torch.cuda.synchronize()
evs = []
for _ in range(30):
    p = d["k"] % 2
    hb, he, gb_, ge = E(), E(), E(), E()
    with torch.cuda.stream(sh):
        sh.wait_event(d["ev_b"][p])
        hb.record(sh); d["gh"][p].replay(); he.record(sh); d["ev_h"][p].record(sh)
    with torch.cuda.stream(sg):
        sg.wait_event(d["ev_h"][1 - p])
        gb_.record(sg); d["gb"][1 - p].replay(); ge.record(sg); d["ev_b"][1 - p].record(sg)
    d["k"] += 1
    evs.append((hb, he, gb_, ge))
torch.cuda.synchronize()
h = sum(a.elapsed_time(b) for a, b, _, _ in evs[5:]) / len(evs[5:])
g = sum(a.elapsed_time(b) for _, _, a, b in evs[5:]) / len(evs[5:])
step = evs[5][2].elapsed_time(evs[-1][2]) / (len(evs) - 6)
print("overlapped: backbone graph %.3f ms, head graph %.3f ms, step %.3f ms (head CUs: %d)" % (g, h, step, (ts.head_cus if ts.head_cus > 0 else 32) if d["masked"] else 0))
