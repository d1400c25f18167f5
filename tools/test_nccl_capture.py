"""Can an RCCL all-reduce be captured in a hipGraph here?  Run with torchrun --nproc-per-node 1."""
import os, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(852072, device="cuda")
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y = x * 2
    dist.all_reduce(y)
    z = y + 1
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print("captured all_reduce replay ok:", float(z[0]), "world", dist.get_world_size())
dist.destroy_process_group()
