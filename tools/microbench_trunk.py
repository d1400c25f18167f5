"""Per-shape timing of the ResNet-50 trunk convolutions (csrc/cnn.hip) at a given batch / input size."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd._lib import call, ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
st = torch.cuda.current_stream().cuda_stream
WS = torch.empty(8 * 160 * 128 * 128, device="cuda")          # split-K scratch of eventad_conv2d_nhwc
def timeit(fn, it=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
shapes = [("stem 7x7/2", S, 4, 64, 7, 2, 1)]
h = S // 4
cin = 64
for li, (mid, n, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
    for blk in range(n):
        s = stride if blk == 0 else 1
        shapes.append(("l%d.%d 1x1a" % (li + 1, blk), h, cin, mid, 1, 1, 1))
        shapes.append(("l%d.%d 3x3/%d" % (li + 1, blk, s), h, mid, mid, 3, s, 1))
        ho = h // s
        shapes.append(("l%d.%d 1x1b" % (li + 1, blk), ho, mid, mid * 4, 1, 1, 1))
        if blk == 0:
            shapes.append(("l%d.%d down/%d" % (li + 1, blk, s), h, cin, mid * 4, 1, s, 1))
        cin, h = mid * 4, ho
agg = {}
for name, H, Cin, Cout, Kk, s, _ in shapes:
    key = (H, Cin, Cout, Kk, s)
    agg.setdefault(key, [name, 0])[1] += 1
tot = 0.0; totf = 0.0
for (H, Cin, Cout, Kk, s), (name, cnt) in agg.items():
    pad = Kk // 2
    OH = (H + 2 * pad - Kk) // s + 1
    x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Kk * Kk * Cin, Cout, device="cuda"); o = torch.empty(B, OH, OH, Cout, device="cuda")
    ms = timeit(lambda: call("eventad_conv2d_nhwc", ptr(x), ptr(w), ptr(o), None, None, None, None, B, H, H, Cin, OH, OH, Cout, Kk, Kk, s, pad, 0, ptr(WS), WS.numel(), st))
    fl = 2.0 * B * OH * OH * Cout * Kk * Kk * Cin
    tot += ms * cnt; totf += fl * cnt
    print("%-14s x%d  in %3dx%3d %4d->%4d k%d s%d  rows %6d : %.3f ms  %5.1f TF   (%.2f ms total)" % (name, cnt, H, H, Cin, Cout, Kk, s, B * OH * OH, ms, fl / ms / 1e9, ms * cnt))
print("trunk convs: %.2f ms, %.1f GFLOP, %.1f TF average" % (tot, totf / 1e9, totf / tot / 1e9))
