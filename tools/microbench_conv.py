"""A few representative trunk convolutions (csrc/cnn.hip) timed back to back: for trying kernel variants.
usage: microbench_conv.py [B S]   (default 8 960: the 1280x720 pixel count)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd._lib import call, ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 960
st = torch.cuda.current_stream().cuda_stream
WS = torch.empty(8 * 160 * 128 * 128, device="cuda")
def timeit(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
# name, H (input side), Cin, Cout, k, stride
shapes = [("l1 1x1b", S // 4, 64, 256, 1, 1), ("l1 3x3", S // 4, 64, 64, 3, 1), ("l2 1x1a", S // 8, 512, 128, 1, 1),
          ("l2 3x3", S // 8, 128, 128, 3, 1), ("l3 3x3", S // 16, 256, 256, 3, 1), ("l3 1x1b", S // 16, 256, 1024, 1, 1),
          ("l4 3x3", S // 32, 512, 512, 3, 1)]
tot = 0.0
for name, H, Cin, Cout, Kk, s in shapes:
    pad = Kk // 2
    OH = (H + 2 * pad - Kk) // s + 1
    x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Kk * Kk * Cin, Cout, device="cuda"); o = torch.empty(B, OH, OH, Cout, device="cuda")
    ms = timeit(lambda: call("eventad_conv2d_nhwc", ptr(x), ptr(w), ptr(o), None, None, None, None, B, H, H, Cin, OH, OH, Cout, Kk, Kk, s, pad, 0, ptr(WS), WS.numel(), st))
    fl = 2.0 * B * OH * OH * Cout * Kk * Kk * Cin
    tot += ms
    print("%-8s in %3d^2 %4d->%4d k%d rows %7d : %.3f ms %6.1f TF" % (name, H, Cin, Cout, Kk, B * OH * OH, ms, fl / ms / 1e9))
print("sum %.3f ms" % tot)
