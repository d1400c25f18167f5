"""Mirrors and oracle pieces pinned by fixtures that the reference's own importable modules produced
(tests/golden/make_misc_golden.py): config/eventad_config.py, src/dagr/model/networks/ema.py and the hook / 1x1-reducer
semantics of src/dagr/model/networks/net_img.py::HookModule (the ResNet-50 under it is this repository's own
plain-torch module, so the trunk's arithmetic is NOT pinned by this -- only what is tapped and how it is reduced)."""
import json
import os
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "misc_golden.json")))


def test_parse_eventad_args_equals_reference(golden_dir):
    from eventad_amd.config.eventad_config import parse_eventad_args
    for name, case in _gold(golden_dir)["config"].items():
        argv = [a.replace("$ROOT", ROOT) for a in case["argv"]]
        got = {k: (str(v) if isinstance(v, Path) else v) for k, v in vars(parse_eventad_args(argv)).items()}
        want = {k: (v.replace("$ROOT", ROOT) if isinstance(v, str) else v) for k, v in case["args"].items()}
        assert got == want, (name, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)})


def test_model_ema_equals_reference(golden_dir):
    from eventad_amd.dagr.model.networks.ema import ModelEMA
    g = _gold(golden_dir)["ema"]
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4))
    net.load_state_dict({k: torch.tensor(v) for k, v in g["init"].items()})
    net.train()
    ema = ModelEMA(net, decay=0.9, updates=0)
    assert ema.ema.training == g["ema_training_flag"] and [p.requires_grad for p in ema.ema.parameters()] == g["ema_requires_grad"]
    for step in g["steps"]:
        net.load_state_dict({k: torch.tensor(v) for k, v in step["model"].items()})
        ema.update(net)
        assert ema.updates == step["updates"]
        for k, v in ema.ema.state_dict().items():
            assert torch.equal(v, torch.tensor(step["ema"][k], dtype=v.dtype)), k


def test_hook_module_taps_and_reducers_equal_reference(golden_dir):
    from eventad_amd import weights
    from eventad_amd.dagr.model.networks.net_img import HookModule
    from oracle import resnet_ref
    gold = torch.load(os.path.join(golden_dir, "hook_golden.pt"), weights_only=True)
    sd = weights.random_cnn_state()
    image = torch.rand(1, 3, gold["H"], gold["W"], generator=torch.Generator().manual_seed(gold["image_seed"]))
    got = resnet_ref.hook_module_forward(sd, image)
    assert len(got) == len(gold["features"]) == 5
    for a, b in zip(got, gold["features"]):
        assert a.shape == b.shape and torch.allclose(a, b, atol=1e-5, rtol=1e-5), float((a - b).abs().max())
    # the product's parameter holder exposes exactly the reference module's state-dict keys (strict loads both ways)
    assert sorted(HookModule().state_dict().keys()) == gold["state_dict_keys"]
