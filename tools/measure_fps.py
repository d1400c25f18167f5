"""The reference's own speed figure ("FPS": boxes of the current + previous frame per second of evaluation forward passes,
utils/fps.py:6-96; 595.48 in its committed test run, batch size 6, hardware not stated) on synthetic ROL-shaped clips:
256x256, 10 tracks per clip, live ResNet-50 trunk, evaluation mode, one synchronised forward per batch.
usage: python tools/measure_fps.py [events_per_clip=50000] [batch=6] [num_batches=20] [graph=0]
graph=1: the forward as one replayed hipGraph (eventad_amd.infer_step.InferStep; every batch has the captured shapes), driven
through the same measure_fps."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from eventad_amd import synthetic
from eventad_amd.utils.fps import measure_fps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 20
use_graph = len(sys.argv) > 4 and sys.argv[4] == "1"
W = H = 256
dev = torch.device("cuda:0")
args, model, _ = build_model(W, H, B, dev, live_cnn=True)
warm = 5
loader = [synthetic.make_batch(W, H, n, B, first_clip=i * B, with_image=True) for i in range(warm + nb)]
if use_graph:
    from eventad_amd.infer_step import InferStep
    from eventad_amd.dagr.utils.buffers import format_data

    class Replayed(torch.nn.Module):
        """model(data, labels, testing=True) of the timing loop -> load the batch into the graph's buffers + replay"""
        def __init__(self, m, first):
            super().__init__()
            self.step = InferStep(m)
            self.step.capture(first)

        def forward(self, data, labels, testing=True):
            return self.step.replay(data)

    model = Replayed(model, format_data(synthetic.make_batch(W, H, n, B, first_clip=9999, with_image=True).to(dev)))
res = measure_fps(model, loader, dev, warmup_batches=warm, num_batches=nb)
res.update({"events_per_clip": n, "batch": B, "sensor": [W, H], "trunk": "live", "launch": "hipGraph replay" if use_graph else "eager", "ms_per_batch": 1e3 * res["total_time"] / max(res["batches"], 1)})
print(json.dumps(res))
