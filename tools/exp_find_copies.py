"""Which device-to-device copies one training step issues (torch profiler, eager step), with the Python frames that issue them."""
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
data = format_data(synthetic.make_config_batch(cfg).to(dev))
if getattr(data, "image_feats", None) is not None:
    data.image_feats = [m.contiguous(memory_format=torch.channels_last) for m in data.image_feats]
model.train()
ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
for _ in range(3):
    ts.step(data)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(data)
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::zeros", "aten::empty", "aten::contiguous", "aten::to"):
        st = [f for f in (e.stack or []) if "eventad_amd" in f or "bench" in f][:3]
        key = (e.name, tuple(st))
        seen[key] = seen.get(key, 0) + 1
for (name, st), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(c, name, " <- ".join(s.strip() for s in st))
