"""How many runs of consecutive ELL slots share one LUT cell and hold consecutive sources (dense inputs)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
W, H, n, B, live = synthetic.CONFIGS[cfg]
dev = torch.device("cuda:0")
data = format_data(synthetic.make_config_batch(cfg).to(dev))
if live:
    g = torch.Generator().manual_seed(0)
    data.image_feats = [torch.randn(B, c, h, w, generator=g).to(dev) for c, (h, w) in zip(synthetic.FEATURE_CHANNELS, synthetic.feature_map_sizes(W, H))]
nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
torch.cuda.synchronize()
N = data.pos.shape[0]
M = min(N, 4_000_000)
src = nb.ws["ell_src"][: 16 * N].view(N, 16)[:M].long()
stp = nb.ws["ell_step"][: 16 * N].view(N, 16)[:M].long()
deg = nb.ws["deg"][:N][:M].long()
e = torch.arange(16, device=dev)[None, :]
live_slot = e < deg[:, None]
same = torch.zeros_like(live_slot)
same[:, 2:] = (stp[:, 2:] == stp[:, 1:-1]) & (src[:, 2:] == src[:, 1:-1] - 1)      # slot 0 is the self edge: never merged
start = live_slot & ~same
runs = start.sum(1)
# macro slots: a run cut into pieces of at most 4
pos_in_run = torch.zeros((M, 16), dtype=torch.long, device=dev)
for k in range(1, 16):
    pos_in_run[:, k] = torch.where(same[:, k], pos_in_run[:, k - 1] + 1, torch.zeros_like(pos_in_run[:, k]))
macro = (live_slot & (pos_in_run % 4 == 0)).sum(1)
print("%s: nodes %d  mean degree %.2f  mean runs %.2f  mean macro slots (<=4 per piece) %.2f" % (cfg, M, deg.float().mean(), runs.float().mean(), macro.float().mean()))
# what a wave pays: 16 destinations in lock step, sorted by the count inside groups of 64
def wave_cost(v):
    v = v[: (M // 64) * 64].view(-1, 64).sort(1, descending=True)[0].view(-1, 4, 16)
    return v.max(2)[0].float().mean().item()
print("lock-step cost per destination (max of 16 after ranking 64): degree %.2f  runs %.2f  macro %.2f" % (wave_cost(deg), wave_cost(runs), wave_cost(macro)))
