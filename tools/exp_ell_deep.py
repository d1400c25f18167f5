"""Edge search alone at the 32 M-event configuration's density (256x256, 2 M events per clip, B clips): the deep-pixel first
pass (variant 5, the default from 8 events per pixel) against the register form (variant 2).
usage: exp_ell_deep.py [B [W H events_per_clip]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H, n = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 256, 2000000)
dev = torch.device("cuda:0")
data = format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True).to(dev))
nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
rows = {}
for v in (2, 5, 6, 2, 5, 6):
    nb.ell_variant = v
    for _ in range(2):
        nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
    nb.timing = {}
    for _ in range(5):
        nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
    s = nb.timing_summary()
    N = data.pos.shape[0]
    print("variant %d: edge search %.3f ms (N %d, mean degree %.2f)" % (v, [x[1] for k, x in s.items() if "build_ell" in k][0], N,
          float(nb.ws["deg"][:N].float().mean())), flush=True)
    rows[v] = (nb.ws["deg"][:N].clone(), nb.ws["ell_src"][:N].clone())
print("same rows:", all(torch.equal(a, b) for a, b in zip(rows[2], rows[5])))
