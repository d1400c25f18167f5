"""How much of Z = X.[W_0..W_24|root|skip] the blends of a pooled level read: nodes, edges, and the (source node, kernel
cell) pairs that some edge's 4-tap blend touches, per level.  usage: exp_tail_cells.py [config]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from eventad_amd import synthetic
from eventad_amd.dagr.utils.buffers import format_data
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
W, H, n, B, live = synthetic.CONFIGS[cfg]
dev = torch.device("cuda:0")
args, model, sd = bench.build_model(W, H, B, dev, live_cnn=live)
data = format_data(synthetic.make_config_batch(cfg).to(dev))
with torch.no_grad():
    model.dagr_model.backbone(data)
nb = model.dagr_model.backbone._native
g = nb.geom
for lv, L in enumerate(nb._tail_view):
    nn, ne = int(L.n_dev.item()), int(L.e_dev.item())
    src = L.csr_src[:ne].long()
    pk = L.csr_dxy[:ne]
    dx = ((pk & 0xffff) ^ 0x8000) - 0x8000
    dy = pk >> 16
    dvx, dvy = g.lut_div[lv + 1]
    def cells(d, div):
        v = (d.float() / div + 0.5) * 4.0
        t = v.to(torch.int64)
        return t % 5, (t + 1) % 5, v - torch.floor(v)
    x0, x1, fx = cells(dx, dvx)
    y0, y1, fy = cells(dy, dvy)
    used = torch.zeros((nn, 25), dtype=torch.bool, device=dev)
    # a tap with weight exactly 0 still reads its cell in the kernel; count both ways
    for cx, cy in ((x0, y0), (x1, y0), (x0, y1), (x1, y1)):
        used[src, cx + 5 * cy] = True
    nz = torch.zeros((nn, 25), dtype=torch.bool, device=dev)
    for cx, cy, wx, wy in ((x0, y0, 1 - fx, 1 - fy), (x1, y0, fx, 1 - fy), (x0, y1, 1 - fx, fy), (x1, y1, fx, fy)):
        m = (wx * wy) != 0
        nz[src[m], (cx + 5 * cy)[m]] = True
    print("level %d: nodes %d edges %d (%.1f per node); cells read per source node %.1f of 25 (non-zero weight %.1f); histogram of cells %s"
          % (lv, nn, ne, ne / max(nn, 1), used.sum().item() / max(nn, 1), nz.sum().item() / max(nn, 1),
             used.sum(0).tolist()))
