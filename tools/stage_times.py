"""Per-call HIP-event timings of one backbone forward on a config's batch (eager launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
W, H, n, B, live = synthetic.CONFIGS[cfg]
dev = torch.device("cuda:0")
data = format_data(synthetic.make_config_batch(cfg).to(dev))
if live:
    g = torch.Generator().manual_seed(0)
    data.image_feats = [torch.randn(B, c, h, w, generator=g).to(dev) for c, (h, w) in zip(synthetic.FEATURE_CHANNELS, synthetic.feature_map_sizes(W, H))]
nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
for _ in range(3):
    nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
nb.timing = {}
for _ in range(10):
    nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
s = nb.timing_summary()
tot = 0.0
for k, v in s.items():
    print("%-36s %8.1f us" % (k, v[1] * 1e3)); tot += v[1]
print("total %.1f us; N %d E %d; level nodes %s" % (tot * 1e3, data.pos.shape[0], int(nb.ws["deg"][: data.pos.shape[0]].sum()),
      [int(L.n_dev.item()) for L in nb.ws["levels"]]))
