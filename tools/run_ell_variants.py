import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
W, H, n, B, live = synthetic.CONFIGS["C2"]
dev = torch.device("cuda:0")
data = format_data(synthetic.make_config_batch("C2").to(dev))
nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
nb.fused_mark = False
for sw in (-1, -2):
    nb.strip_width = sw
    for _ in range(3):
        nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
torch.cuda.synchronize()
