"""Host logic of the product head (static masked formulation, eventad_amd/models/EventAD.py) on CPU
against the reference's golden vectors (tests/golden/make_head_golden.py).  No native calls here."""
import os
from types import SimpleNamespace

import pytest
import torch

from eventad_amd.models.EventAD import EventADModel, T
from tests.helpers import seeded_head_state, make_head_case, grad_sample_index

CASES = ["head_case_small", "head_case_rol", "head_case_dense"]


def _model(seed):
    m = EventADModel(torch.nn.Module())
    res = m.load_state_dict(seeded_head_state(seed), strict=False)
    assert all(k.startswith("fusion_module.attention.") for k in res.missing_keys) and not res.unexpected_keys
    return m.eval()


@pytest.mark.parametrize("name", CASES)
def test_masked_head_matches_reference_golden(name, golden_dir):
    gold = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    kw = gold["kw"]
    case = make_head_case(**kw)
    m = _model(kw["seed"])
    data = SimpleNamespace(**{k: case[k] for k in ("num_graphs", "bbox", "bbox_batch", "bbox0", "bbox0_batch",
                                                   "width", "height")})
    loss, logits, valid, targets = m.head(gold["feats"], data, data.bbox[:, 4])
    got = logits[valid]                               # row-major (clip, track) order = the reference's order
    assert got.shape == gold["logits"].shape
    assert torch.equal(targets[valid], gold["targets"])
    assert torch.allclose(got, gold["logits"], atol=5e-6, rtol=1e-5)
    assert torch.equal(got.argmax(1), gold["logits"].argmax(1))
    assert abs(float(loss) - float(gold["loss"])) < 1e-4 * max(1.0, float(gold["loss"]))
    loss.backward()
    params = dict(m.named_parameters())
    for k, g in gold["grads"].items():
        if g is None:
            assert params[k].grad is None
            continue
        mine = params[k].grad.flatten()
        assert abs(float(mine.norm()) - g["norm"]) <= 1e-4 * max(1.0, g["norm"]), k
        assert torch.allclose(mine[grad_sample_index(mine.numel(), kw["seed"])], g["sample"], atol=1e-5, rtol=1e-4), k


def test_state_dict_keys_match_reference_contract():
    from eventad_amd.weights import HEAD_SHAPES
    m = EventADModel(torch.nn.Module())
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == HEAD_SHAPES
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 1115234      # trainable parameters of the reference head (GRUs, fusion, attention, classifier)


def test_no_objects_gives_zero_loss_with_grad():
    case = make_head_case(seed=5, num_graphs=2, tracks=0, nodes_per_clip=5, drop_prob=0.0, empty_clip=None, far_box=False)
    m = _model(5)
    data = SimpleNamespace(**{k: case[k] for k in ("num_graphs", "bbox", "bbox_batch", "width", "height")})
    loss, logits, valid, _ = m.head(torch.zeros(2, 2, T, 64), data, data.bbox[:, 4])
    assert float(loss) == 0.0 and loss.requires_grad and not bool(valid.any())
    loss.backward()
