"""Where the wave-per-pixel edge search spends its passes: from the ELL rows of a configuration's batch, how many events are
complete after the first pass (GS1 spiral steps), how many 64-step passes the rest take, and how many pixels are visited by
a pass for the sake of how many events.  usage: python tools/exp_ell_passes.py C5 [GS1]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd.dagr.utils.buffers import format_data
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
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
steps = (2 * nb.geom.radius_px + 1) ** 2
print("radius %d px, time window %d us, FIFO depth %d" % (nb.geom.radius_px, nb.geom.delta_t_us, nb.geom.Q))
stp = nb.ws["ell_step"][: 16 * N].view(N, 16).long()
deg = nb.ws["deg"][:N].long()
key = nb.ws["key_s"][:N].long() if "key_s" in nb.ws else None
per_pixel = N / float(B * W * H)
gs1 = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if steps <= 64 else (16 if per_pixel >= 5 else 32))
e = torch.arange(16, device=dev)[None, :]
last = torch.where(e < deg[:, None], stp, torch.zeros_like(stp)).max(1)[0]
last = torch.where(deg < 16, torch.full_like(last, steps - 1), last)          # an incomplete row has seen the whole spiral
done1 = last < gs1
extra = torch.where(done1, torch.zeros_like(last), (last - gs1) // 64 + 1)
print("%s: %d events, %.2f per pixel, %d spiral steps, first pass %d steps" % (cfg, N, per_pixel, steps, gs1))
print("mean degree %.2f, rows with fewer than 16: %.2f %%" % (deg.float().mean(), 100.0 * (deg < 16).float().mean()))
print("complete after the first pass: %.2f %%; 64-step passes per event over all events: %.3f" % (100.0 * done1.float().mean(), extra.float().mean()))
for k in range(0, int(extra.max().item()) + 1):
    print("  events with %d further passes: %.2f %%" % (k, 100.0 * (extra == k).float().mean()))
hist = torch.bincount(torch.clamp(last, max=steps - 1), minlength=steps).float().cumsum(0) / N
for s in (0, 1, 4, 8, 9, 15, 16, 24, 25, 31, 48, 63, 80, 120, steps - 1):
    if s < steps:
        print("  last step needed <= %3d: %.2f %%" % (s, 100.0 * hist[s]))
if key is not None:
    # per pixel (events of one pixel are adjacent in the sorted order): events in the pixel, and the further passes the pixel's wave makes
    pix, inv = torch.unique_consecutive(key, return_inverse=True)
    cnt = torch.bincount(inv)
    pmax = torch.zeros_like(cnt).scatter_reduce(0, inv, extra, "amax")
    pn = torch.zeros_like(cnt).scatter_add(0, inv, (extra > 0).long())
    print("pixels with events: %d (%.2f events each); first-pass event rounds per pixel (4 events of a 16-step group at a time): %.2f"
          % (cnt.numel(), cnt.float().mean(), ((cnt + 64 // gs1 - 1) // (64 // gs1)).float().mean()))
    print("further passes per pixel: %.3f; events carried into them per pixel: %.3f" % (pmax.float().mean(), pn.float().mean()))
