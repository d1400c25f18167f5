"""Runs the frozen backbone (CNN branch included when the configuration has one) a few times with the default kernels:
the target program of the rocprofv3 --pmc passes (tools/pmc_passes.sh)."""
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
    for _ in range(4):
        model.dagr_model.backbone(data)
torch.cuda.synchronize()
nb = model.dagr_model.backbone._native
print("N", data.pos.shape[0], "E", int(nb.ws["deg"][: data.pos.shape[0]].sum()))
