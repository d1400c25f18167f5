"""Generates tests/golden/misc_golden.json and tests/golden/hook_golden.pt by IMPORTING the three reference modules
that need nothing but torch / yaml:

    config/eventad_config.py                       -> parsed argument namespaces for several command lines
    src/dagr/model/networks/ema.py                 -> ModelEMA: state after construction and after update() calls
    src/dagr/model/networks/net_img.py             -> HookModule: which activations are tapped (raw conv1 output,
                                                      layer1..4) and how the 1x1 reducers are applied

Run in the build container only (the reference does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_misc_golden.py

What is stored is data only: argument dictionaries, tensors.  The ResNet-50 under the reference's HookModule is this
repository's own plain-torch module (torchvision is not installed), so the hook fixture pins the hook / reducer
semantics of HookModule, not torchvision's trunk.
"""
import importlib.util
import json
import os
import sys
from pathlib import Path

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


YAML = os.path.join(ROOT, "eventad_amd", "config", "dagr-s-dsec.yaml")
EXTRA = os.path.join(HERE, "eventad_extra.yaml")
COMMAND_LINES = {
    "defaults": ["--config", YAML],
    "overrides": ["--config", YAML, "--radius", "0.02", "--batch_size", "3", "--no_events", "--aug_zoom", "2.0",
                  "--pooling_dim_at_output", "4x6", "--checkpoint", "/tmp/x.pth", "--keep_temporal_ordering"],
    "second_yaml": ["--config", YAML, "--eventad_config", EXTRA, "--x_dim", "32", "--learning_rate", "0.01"],
}


def _plain(ns):
    out = {}
    for k, v in sorted(vars(ns).items()):
        out[k] = str(v) if isinstance(v, Path) else v
    return out


def config_golden():
    ref = _load("ref_eventad_config", os.path.join(REF, "config", "eventad_config.py"))
    gold = {}
    for name, argv in COMMAND_LINES.items():
        old = sys.argv
        sys.argv = ["train.py"] + argv
        try:
            gold[name] = {"argv": [a.replace(ROOT, "$ROOT") for a in argv],
                          "args": {k: (v.replace(ROOT, "$ROOT") if isinstance(v, str) else v)
                                   for k, v in _plain(ref.parse_eventad_args()).items()}}
        finally:
            sys.argv = old
    return gold


def ema_golden():
    ref = _load("ref_ema", os.path.join(REF, "src", "dagr", "model", "networks", "ema.py"))
    g = torch.Generator().manual_seed(77)
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4))
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    net.train()
    ema = ref.ModelEMA(net, decay=0.9, updates=0)
    rec = {"init": {k: v.tolist() for k, v in net.state_dict().items()},
           "ema_training_flag": bool(ema.ema.training),
           "ema_requires_grad": [bool(p.requires_grad) for p in ema.ema.parameters()], "steps": []}
    for step in range(3):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn(p.shape, generator=g))
            net[1].running_mean.add_(0.5)
        ema.update(net)
        rec["steps"].append({"model": {k: v.tolist() for k, v in net.state_dict().items()},
                             "ema": {k: v.tolist() for k, v in ema.ema.state_dict().items()},
                             "updates": ema.updates})
    return rec


def hook_golden():
    ref = _load("ref_net_img", os.path.join(REF, "src", "dagr", "model", "networks", "net_img.py"))
    from eventad_amd import weights
    from tests.torch_resnet import TorchResNet50
    H = W = 64
    sd = weights.random_cnn_state()
    trunk = TorchResNet50()
    hook = ref.HookModule(trunk, input_channels=3, height=H, width=W,                       # as net.py:41-49 calls it
                          feature_layers=["conv1", "layer1", "layer2", "layer3", "layer4"], output_layers=["layer3", "layer4"],
                          feature_channels=[16, 64, 64, 64, 64], output_channels=[256, 256])
    missing = hook.load_state_dict({k[len("backbone.net."):]: v for k, v in sd.items()}, strict=True)
    hook.eval()
    image = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        features, outputs = hook(image)
    torch.save({"image_seed": 5, "H": H, "W": W, "features": [f.clone() for f in features],
                "output_shapes": [list(o.shape) for o in outputs],
                "state_dict_keys": sorted(hook.state_dict().keys())}, os.path.join(HERE, "hook_golden.pt"))
    print("hook features", [tuple(f.shape) for f in features], "bytes", os.path.getsize(os.path.join(HERE, "hook_golden.pt")))


if __name__ == "__main__":
    gold = {"config": config_golden(), "ema": ema_golden()}
    with open(os.path.join(HERE, "misc_golden.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    print("config cases", list(gold["config"]), "ema steps", len(gold["ema"]["steps"]))
    hook_golden()
