import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd._lib import call, ptr
st = torch.cuda.current_stream().cuda_stream
WS = torch.empty(8 * 160 * 128 * 128, device="cuda")          # split-K scratch of eventad_conv2d_nhwc
def timeit(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (M, N, K) in [(17920, 1728, 84), (17920, 1664, 64), (4480, 1728, 132), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device="cuda"); Bm = torch.randn(K, N, device="cuda"); C = torch.empty(M, N, device="cuda")
    ms = timeit(lambda: call("eventad_gemm_f32", ptr(A), K, ptr(Bm), N, ptr(C), N, M, None, N, K, None, None, None, None, 0, 0, st))
    ref = A @ Bm
    print("gemm %5dx%5dx%5d: %.3f ms  %.1f TF  maxerr %.2e" % (M, N, K, ms, 2 * M * N * K / ms / 1e9, float((C - ref).abs().max())))
for (B, H, Cin, Cout, Kk, s) in [(8, 64, 64, 64, 3, 1), (8, 64, 256, 64, 1, 1), (8, 32, 128, 128, 3, 1), (8, 16, 256, 256, 3, 1), (8, 128 * 2, 4, 64, 7, 2)]:
    pad = Kk // 2
    OH = (H + 2 * pad - Kk) // s + 1
    x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Kk * Kk * Cin, Cout, device="cuda"); o = torch.empty(B, OH, OH, Cout, device="cuda")
    ms = timeit(lambda: call("eventad_conv2d_nhwc", ptr(x), ptr(w), ptr(o), None, None, None, None, B, H, H, Cin, OH, OH, Cout, Kk, Kk, s, pad, 0, ptr(WS), WS.numel(), st))
    print("conv B%d %dx%d %d->%d k%d s%d: %.3f ms  %.1f TF" % (B, H, H, Cin, Cout, Kk, s, ms, 2 * B * OH * OH * Cout * Kk * Kk * Cin / ms / 1e9))
