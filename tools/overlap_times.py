"""How long do the backbone graphs and the head graph take when they overlap (the pipeline of TrainStep.capture_dual),
and alone?  HIP events on each stream around its graph replay, averaged over steady-state steps."""
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
if getattr(data, "image_feats", None) is not None:
    data.image_feats = [m.contiguous(memory_format=torch.channels_last) for m in data.image_feats]
ts = TrainStep(model, lr=args.lr, weight_decay=args.weight_decay)
ts.capture_dual(data)
d = ts._dual
nch = d["n"]
sgs, sh = ts._streams(d["masked"], nch)
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
print("alone: backbone graph %.3f ms, head graph %.3f ms" % (timed(sgs[0], d["gb"][0]), timed(sh, d["gh"][0])))
torch.cuda.synchronize()
ts.capture_dual(data)
d = ts._dual
sgs, sh = ts._streams(d["masked"], nch)
for _ in range(10):
    ts.replay_dual(join=False)
torch.cuda.synchronize()
evs = []
for _ in range(40):                                          # the body of replay_dual(join=False) with events around the graphs
    c, c_next = d["k"] % nch, (d["k"] + 1) % nch
    hb, he, gb_, ge = E(), E(), E(), E()
    with torch.cuda.stream(sh):
        sh.wait_event(d["ev_b"][c])
        hb.record(sh); d["gh"][c].replay(); he.record(sh); d["ev_h"][c].record(sh)
    sg = sgs[c_next]
    with torch.cuda.stream(sg):
        sg.wait_event(d["ev_h"][c_next])
        gb_.record(sg); d["gb"][c_next].replay(); ge.record(sg); d["ev_b"][c_next].record(sg)
    d["k"] += 1
    evs.append((hb, he, gb_, ge))
torch.cuda.synchronize()
s = evs[8:]
h = sum(a.elapsed_time(b) for a, b, _, _ in s) / len(s)
g = sum(a.elapsed_time(b) for _, _, a, b in s) / len(s)
gap = sum(s[i][1].elapsed_time(s[i + 1][0]) for i in range(len(s) - 1)) / (len(s) - 1)      # head k end -> head k+1 start
step = s[0][0].elapsed_time(s[-1][0]) / (len(s) - 1)
print("overlapped, %d backbone chains: backbone graph %.3f ms, head graph %.3f ms, idle between heads %.3f ms, step %.3f ms (head CUs: %d)"
      % (nch, g, h, gap, step, (ts.head_cus if ts.head_cus > 0 else 32) if d["masked"] else 0))
