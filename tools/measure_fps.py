"""The reference's own speed figure ("FPS": boxes of the current + previous frame per second of evaluation forward passes,
utils/fps.py:6-96; 595.48 in its committed test run, batch size 6, hardware not stated) on synthetic ROL-shaped clips:
256x256, 10 tracks per clip, live ResNet-50 trunk, evaluation mode, one synchronised forward per batch.
usage: python tools/measure_fps.py [events_per_clip=50000] [batch=6] [num_batches=20]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from eventad_amd import synthetic
from eventad_amd.utils.fps import measure_fps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 20
W = H = 256
dev = torch.device("cuda:0")
args, model, _ = build_model(W, H, B, dev, live_cnn=True)
warm = 5
loader = [synthetic.make_batch(W, H, n, B, first_clip=i * B, with_image=True) for i in range(warm + nb)]
res = measure_fps(model, loader, dev, warmup_batches=warm, num_batches=nb)
res.update({"events_per_clip": n, "batch": B, "sensor": [W, H], "trunk": "live", "ms_per_batch": 1e3 * res["total_time"] / max(res["batches"], 1)})
print(json.dumps(res))
