"""Event-level kernel times vs batch size at fixed density (256x256, 50k events/clip): fixed cost vs per-event cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
dev = torch.device("cuda:0")
for B in (2, 4, 8, 16, 32):
    data = format_data(synthetic.make_config_batch("C2", batch_size=B).to(dev))
    nb = NativeBackbone(weights.random_graph_state(), 256, 256, dev)
    for _ in range(3):
        nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
    nb.timing = {}
    for _ in range(8):
        nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
    s = nb.timing_summary()
    print(B, {k.replace("eventad_", ""): round(v[1] * 1e3, 1) for k, v in s.items() if any(t in k for t in ("l1_conv", "build_ell", "pool_seg0", "sort_events"))})
