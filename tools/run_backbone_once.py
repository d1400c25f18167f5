"""Runs the C2 backbone a few times with the default kernels (target of the rocprofv3 --pmc passes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
W, H, n, B, live = synthetic.CONFIGS[cfg]
dev = torch.device("cuda:0")
data = format_data(synthetic.make_config_batch(cfg).to(dev))
nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
for _ in range(4):
    nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
torch.cuda.synchronize()
print("N", data.pos.shape[0], "E", int(nb.ws["deg"].sum()))
