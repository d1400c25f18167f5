"""Micro-benchmark of eventad_build_ell variants on the C2 batch (run on the GPU box)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd import synthetic, weights
from eventad_amd.backbone import NativeBackbone
from eventad_amd._lib import call, ptr
from eventad_amd.dagr.utils.buffers import format_data

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
W, H, n, B, live = synthetic.CONFIGS[cfg]
dev = torch.device("cuda:0")
raw = synthetic.make_config_batch(cfg)
data = format_data(raw.to(dev))
nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
feats = data.image_feats
if feats is None:                                            # live-CNN configs: any maps of the right shape will do here
    feats = [torch.randn(B, c, h, w, device=dev) for c, (h, w) in zip(synthetic.FEATURE_CHANNELS, synthetic.feature_map_sizes(W, H))]
res = nb.forward(data.pos, data.x, data.batch, B, feats)
torch.cuda.synchronize()
ws, g = res["ws"], nb.geom
N = data.pos.shape[0]
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, it=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def ell(bitmap, sw=0):
    call("eventad_build_ell", ptr(ws["sid"]), ptr(ws["ts_s"]), ptr(ws["ev2"]), ptr(ws["key_s"]), ptr(ws["pix_start"]),
         ptr(ws["counters"]), N, B, g.W, g.H, g.radius_px, g.delta_t_us, 16, g.Q, ptr(ws["ell_src"]), ptr(ws["ell_step"]),
         ptr(ws["deg"]), ptr(ws["node_slot0"]), ptr(ws["node_vox0"]), g.grid[0][0], g.grid[0][1], bitmap, sw, 0, st)


def ell_old():
    call("eventad_build_ell", ptr(ws["sid"]), ptr(ws["ts_s"]), None, ptr(ws["key_s"]), ptr(ws["pix_start"]),
         ptr(ws["counters"]), N, B, g.W, g.H, g.radius_px, g.delta_t_us, 16, g.Q, ptr(ws["ell_src"]), ptr(ws["ell_step"]),
         ptr(ws["deg"]), None, None, g.grid[0][0], g.grid[0][1], None, 0, 0, st)


print("N", N, "E", int(ws["deg"].sum()), "max deg", int(ws["deg"].max()), "deg==16 frac", float((ws["deg"] == 16).float().mean()))
print("build_ell reference-style: %.1f us" % timeit(ell_old))
ref_src, ref_deg = ws["ell_src"].clone(), ws["deg"].clone()
same = lambda: torch.equal(ws["deg"], ref_deg) and bool((torch.where(torch.arange(16, device=dev).view(1, -1) < ref_deg.view(-1, 1), ws["ell_src"] - ref_src, 0) == 0).all())
for name, v in (("register rows (spiral)", 0), ("row scan", 1)):
    t = timeit(lambda: ell(None, v))
    print("build_ell %-22s: %.1f us  (matches %s)  longest segment %d" % (name, t, same(), int(ws["counters"][3])))
    print("           + fused mark           : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap), v)))
L = ws["levels"][0]
print("mark_ell separate                : %.1f us" % timeit(lambda: call("eventad_pool_edges_mark_ell", ptr(ws["ell_src"]), ptr(ws["deg"]),
      ptr(ws["node_slot0"]), N, 16, L.nx, L.ny, B, ptr(L.bitmap), ptr(ws["counters"]), st)))
