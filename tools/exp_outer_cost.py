"""Large-radius layer-1 convolution: how many (pass, slot) steps take the global-memory table branch, with the passes of 16
destinations formed by degree (as the kernel does) or by the row's first outer slot?  Cost model: 1 per LDS step, 3 per
global step.  usage: exp_outer_cost.py <config> <resident cells>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
cfg, nlds = sys.argv[1], int(sys.argv[2])
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
M = (min(N, 2_000_000) // 64) * 64
stp = nb.ws["ell_step"][:N][:M].long() & 0xffff
deg = nb.ws["deg"][:N][:M].long().clamp(max=16)
e = torch.arange(16, device=dev)[None, :]
live_slot = e < deg[:, None]
outer = live_slot & (stp >= nlds) & (e > 0)
print("%s: resident cells %d of %d; outer slots %.1f %% of live slots; rows with an outer slot %.1f %%" % (
    cfg, nlds, nb.geom.steps, 100.0 * outer.sum().item() / live_slot.sum().item(), 100.0 * outer.any(1).float().mean().item()))
first_outer = torch.where(outer.any(1), outer.float().argmax(1), torch.full_like(deg, 16))

def cost(order_key):
    # groups of 64 rows, ranked by the key (descending, stable), passes of 16
    idx = torch.sort(order_key.view(-1, 64), dim=1, descending=True, stable=True)[1]
    o = torch.gather(outer.view(-1, 64, 16), 1, idx[:, :, None].expand(-1, -1, 16)).view(-1, 4, 16, 16)
    d = torch.gather(deg.view(-1, 64), 1, idx).view(-1, 4, 16)
    maxd = d.max(2)[0]                                         # slots a pass runs
    any_outer = o.any(2)                                       # [groups, passes, slot]
    run = e[None] < maxd[:, :, None]
    return (run & ~any_outer).sum().item() + 3 * (run & any_outer).sum().item(), run.sum().item()

c_deg, steps_deg = cost(deg)
c_f, steps_f = cost(first_outer * 17 + deg)
print("passes by degree: %d steps, cost %d; by first outer slot: %d steps, cost %d (%.2f x)" % (steps_deg, c_deg, steps_f, c_f, c_f / c_deg))
