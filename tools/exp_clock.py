"""Does a short MFMA kernel run faster once the GPU has been busy for a while?  The same shallow-K product timed in
groups of 50 launches, back to back."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventad_amd._lib import call, ptr
st = torch.cuda.current_stream().cuda_stream
M, N, K = 17920, 1728, 84
A = torch.randn(M, K, device="cuda"); Bm = torch.randn(K, N, device="cuda"); C = torch.empty(M, N, device="cuda")
f = lambda: call("eventad_gemm_f32", ptr(A), K, ptr(Bm), N, ptr(C), N, M, None, N, K, None, None, None, None, 0, 0, st)
for _ in range(3): f()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for g in range(40):
    for _ in range(50): f()
    ev[g + 1].record()
torch.cuda.synchronize()
print("us per launch, groups of 50:", " ".join("%.1f" % (ev[g].elapsed_time(ev[g + 1]) * 20) for g in range(40)))
