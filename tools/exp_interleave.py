"""Experiment: throughput of TWO backbone+ROI graphs (independent workspaces) replayed concurrently on two streams,
against one graph replayed back to back.  The backbone is a chain of ~45 launches, half of them latency bound: two
interleaved chains fill each other's gaps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from eventad_amd import synthetic
from eventad_amd.dagr.utils.buffers import format_data
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
W, H, n, B, live = synthetic.CONFIGS[cfg]
B = int(sys.argv[2]) if len(sys.argv) > 2 else B            # clips per backbone pass (default: the configuration's batch)
dev = torch.device("cuda:0")
graphs, keep = [], []
NG = int(sys.argv[3]) if len(sys.argv) > 3 else 2           # graphs (= streams) in flight
if NG > 2:
    # three or more backbone passes in flight on whole-chip streams: a training run in that form at 32 M events once ended in a
    # GPU memory fault (a likely cause in the training step's buffer handling is fixed; this form has not been run since and
    # TrainStep.capture_dual refuses it too)
    raise SystemExit("exp_interleave: more than 2 concurrent backbone graphs on unpartitioned streams are refused")
for i in range(NG):
    args, model, sd = bench.build_model(W, H, B, dev, live_cnn=live)
    model.train()
    data = format_data(synthetic.make_config_batch(cfg, first_clip=B * i, batch_size=B).to(dev))
    with torch.no_grad():
        for _ in range(3):
            model.extract_features(data, frames=(1,))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            f = model.extract_features(data, frames=(1,))
    graphs.append(g); keep.append((model, data, f))
s = [torch.cuda.Stream() for _ in range(NG)]
E = lambda: torch.cuda.Event(enable_timing=True)
def run(which, reps=40):
    torch.cuda.synchronize()
    a, b = E(), E()
    a.record()
    for st in s:
        st.wait_event(a)
    for _ in range(reps):
        for i in which:
            with torch.cuda.stream(s[i]):
                graphs[i].replay()
    for st in s:
        torch.cuda.current_stream().wait_stream(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * len(which))
for _ in range(2):
    print("%d clips per pass -- one chain: %.3f ms per pass; %d interleaved chains: %.3f ms per pass" % (B, run([0]), NG, run(list(range(NG)))))
