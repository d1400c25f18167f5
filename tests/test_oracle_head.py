"""Pins oracle/head_ref.py against golden vectors produced by the imported reference
(models/EventAD.py, see tests/golden/make_head_golden.py)."""
import os
from types import SimpleNamespace

import pytest
import torch

from oracle import head_ref
from tests.helpers import seeded_head_state, make_head_case, grad_sample_index

CASES = ["head_case_small", "head_case_rol", "head_case_dense"]


def _run_oracle(gold):
    kw = gold["kw"]
    case = make_head_case(**kw)
    p = {k: v.clone().requires_grad_(True) for k, v in seeded_head_state(kw["seed"]).items()}
    out4 = SimpleNamespace(x=case["out4_x"], pos=case["out4_pos"], batch=case["out4_batch"])
    data = SimpleNamespace(**{k: case[k] for k in ("num_graphs", "bbox", "bbox_batch", "bbox0",
                                                   "bbox0_batch", "width", "height")})
    feats = head_ref.roi_features(out4, data)
    loss, outs = head_ref.head_forward(p, feats, data, data.bbox[:, 4])
    return p, feats, loss, outs


@pytest.mark.parametrize("name", CASES)
def test_head_oracle_matches_reference_golden(name, golden_dir):
    gold = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    p, feats, loss, outs = _run_oracle(gold)
    assert torch.allclose(feats, gold["feats"], atol=1e-6, rtol=1e-5)
    logits = torch.cat([o[2] for o in outs], 0)
    targets = torch.cat([o[3] for o in outs], 0)
    assert logits.shape == gold["logits"].shape
    assert torch.equal(targets, gold["targets"])
    assert torch.allclose(logits, gold["logits"], atol=2e-6, rtol=1e-5)
    assert torch.equal(logits.argmax(1), gold["logits"].argmax(1))
    assert abs(float(loss) - float(gold["loss"])) < 1e-4 * max(1.0, abs(float(gold["loss"])))
    loss.backward()
    for k, g in gold["grads"].items():
        if g is None:           # fusion_module.attention.* is never used by forward
            assert k.startswith("fusion_module.attention.")
            continue
        mine = p[k].grad.flatten()
        assert abs(float(mine.norm()) - g["norm"]) <= 1e-4 * max(1.0, g["norm"]), k
        idx = grad_sample_index(mine.numel(), gold["kw"]["seed"])
        assert torch.allclose(mine[idx], g["sample"], atol=1e-5, rtol=1e-4), k


def test_zero_object_batch_gives_zero_loss_with_grad():
    # EventAD.py:330-331
    case = make_head_case(seed=5, num_graphs=2, tracks=0, nodes_per_clip=5, drop_prob=0.0,
                          empty_clip=None, far_box=False)
    p = seeded_head_state(5)
    out4 = SimpleNamespace(x=case["out4_x"], pos=case["out4_pos"], batch=case["out4_batch"])
    data = SimpleNamespace(**{k: case[k] for k in ("num_graphs", "bbox", "bbox_batch", "bbox0",
                                                   "bbox0_batch", "width", "height")})
    feats = head_ref.roi_features(out4, data)
    loss, outs = head_ref.head_forward(p, feats, data, data.bbox[:, 4])
    assert outs == [] and float(loss) == 0.0 and loss.requires_grad
