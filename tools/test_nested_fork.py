"""hipGraph capture with a fork inside a forked stream (does the runtime accept nested parallel branches?)."""
import sys, torch
x = torch.zeros(1 << 20, device="cuda"); y = torch.zeros(1 << 20, device="cuda"); z = torch.zeros(1 << 20, device="cuda")
A, Bs, C = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "nested"
def body():
    main = torch.cuda.current_stream()
    A.wait_stream(main); C.wait_stream(main)
    with torch.cuda.stream(A):
        if mode == "nested":
            Bs.wait_stream(A)
            with torch.cuda.stream(Bs):
                y.add_(1)
        x.add_(1)
        if mode == "nested":
            A.wait_stream(Bs)
        x.add_(y)
    with torch.cuda.stream(C):
        z.add_(2)
    main.wait_stream(A); main.wait_stream(C)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    body()
print("captured", flush=True)
for _ in range(3): g.replay()
torch.cuda.synchronize()
print("ok", float(x[0]), float(y[0]), float(z[0]))
