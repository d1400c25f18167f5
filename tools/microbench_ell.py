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
res = nb.forward(data.pos, data.x, data.batch, B, data.image_feats)
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
print("build_ell batched + mark : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap))))
print("build_ell batched no mark: %.1f us" % timeit(lambda: ell(None)))
print("build_ell reference-style: %.1f us" % timeit(ell_old))
ref_src, ref_deg = ws["ell_src"].clone(), ws["deg"].clone()
t = timeit(lambda: ell(None, -1))
ok = torch.equal(ws["deg"], ref_deg) and bool((torch.where(torch.arange(16, device=dev).view(1, -1) < ref_deg.view(-1, 1), ws["ell_src"] - ref_src, 0) == 0).all())
print("build_ell register rows  : %.1f us  (matches %s)" % (t, ok))
print("           + fused mark  : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap), -1)))
t = timeit(lambda: ell(None, -4))
ok = torch.equal(ws["deg"], ref_deg) and bool((torch.where(torch.arange(16, device=dev).view(1, -1) < ref_deg.view(-1, 1), ws["ell_src"] - ref_src, 0) == 0).all())
print("build_ell converged lanes: %.1f us  (matches %s)  lmax %d" % (t, ok, int(ws["counters"][3])))
print("           + fused mark  : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap), -4)))
t = timeit(lambda: ell(None, -6))
ok = torch.equal(ws["deg"], ref_deg) and bool((torch.where(torch.arange(16, device=dev).view(1, -1) < ref_deg.view(-1, 1), ws["ell_src"] - ref_src, 0) == 0).all())
print("build_ell row scan       : %.1f us  (matches %s)" % (t, ok))
print("           + fused mark  : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap), -6)))
t = timeit(lambda: ell(None, -5))
ok = torch.equal(ws["deg"], ref_deg) and bool((torch.where(torch.arange(16, device=dev).view(1, -1) < ref_deg.view(-1, 1), ws["ell_src"] - ref_src, 0) == 0).all())
print("build_ell LDS row tiles  : %.1f us  (matches %s)" % (t, ok))
print("           + fused mark  : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap), -5)))
t = timeit(lambda: ell(None, -3))
ok = torch.equal(ws["deg"], ref_deg) and bool((torch.where(torch.arange(16, device=dev).view(1, -1) < ref_deg.view(-1, 1), ws["ell_src"] - ref_src, 0) == 0).all())
print("build_ell wave-per-event : %.1f us  (matches %s)" % (t, ok))
print("           + fused mark  : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap), -3)))
t = timeit(lambda: ell(None, -2))
ok = torch.equal(ws["deg"], ref_deg) and bool((torch.where(torch.arange(16, device=dev).view(1, -1) < ref_deg.view(-1, 1), ws["ell_src"] - ref_src, 0) == 0).all())
print("build_ell row-parallel   : %.1f us  (matches %s)" % (t, ok))
print("           + fused mark  : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap), -2)))
for sw in (16, 32, 64, 128):
    t = timeit(lambda: ell(None, sw))
    ok = torch.equal(ws["deg"], ref_deg) and bool((torch.where(torch.arange(16, device=dev).view(1, -1) < ref_deg.view(-1, 1), ws["ell_src"] - ref_src, 0) == 0).all())
    print("build_ell LDS strip SW=%3d : %.1f us  (matches %s)" % (sw, t, ok))
    print("           + fused mark   : %.1f us" % timeit(lambda: ell(ptr(ws["levels"][0].bitmap), sw)))
L = ws["levels"][0]
print("mark_ell separate        : %.1f us" % timeit(lambda: call("eventad_pool_edges_mark_ell", ptr(ws["ell_src"]), ptr(ws["deg"]),
      ptr(ws["node_slot0"]), N, 16, L.nx, L.ny, B, ptr(L.bitmap), ptr(ws["counters"]), st)))
x = torch.empty(N * 16, dtype=torch.int32, device=dev)
print("memset 25.6MB (N*16*4)    : %.1f us" % timeit(lambda: x.zero_()))
y = torch.empty_like(x)
print("copy 25.6MB               : %.1f us" % timeit(lambda: y.copy_(x)))
