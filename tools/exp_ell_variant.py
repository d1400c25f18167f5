"""Edge search alone on a synthetic batch: time per EVENTAD_ELL_VARIANT.  usage: exp_ell_variant.py W H events_per_clip B"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
W, H, n, B = (int(a) for a in sys.argv[1:5])
dev = torch.device("cuda:0")
data = format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True).to(dev))
for v in ("0", "2", "3", "4"):
    os.environ["EVENTAD_ELL_VARIANT"] = v
    nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
    for _ in range(2):
        nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
    nb.timing = {}
    for _ in range(5):
        nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
    s = nb.timing_summary()
    print("variant %s: edge search %.1f us (N %d, mean degree %.2f)" % (v, [x[1] for k, x in s.items() if "build_ell" in k][0] * 1e3, data.pos.shape[0],
          float(nb.ws["deg"][: data.pos.shape[0]].float().mean())))
