"""Generates tests/golden/head_case*.pt by IMPORTING the reference's models/EventAD.py.

Run in the build container only (the reference does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_head_golden.py

What is stored (data only -- no reference source): the seeded inputs of each case
(out4 nodes, boxes), and the reference's outputs on them in eval mode with the
feature cache cleared: batch_features, every logit with its (clip, track), the
summed cross-entropy and, per head parameter, the gradient's L2 norm plus 4096
sampled entries (all entries for small tensors).  Head weights are not stored:
they are regenerated from the case seed by tests/helpers.py::seeded_head_state.
"""
import os
import sys
from types import SimpleNamespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root, for tests.helpers
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from models.EventAD import EventADModel  # noqa: E402  (the reference, imported not copied)
from tests.helpers import seeded_head_state, make_head_case, grad_sample_index  # noqa: E402


class _StubDagr(torch.nn.Module):
    """Stands in for dagr_model: returns precomputed [out3, out4]."""

    def __init__(self):
        super().__init__()
        self.out = None

    def extract_features(self, data):
        return self.out


def run_case(name, **kw):
    case = make_head_case(**kw)
    stub = _StubDagr()
    model = EventADModel(stub)
    sd = seeded_head_state(kw["seed"])
    missing = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "attention." not in k or "soft_attention" in k], missing
    model.eval()
    out4 = SimpleNamespace(x=case["out4_x"], pos=case["out4_pos"], batch=case["out4_batch"])
    stub.out = [None, out4]
    data = SimpleNamespace(num_graphs=case["num_graphs"], x=torch.ones(4, 1), bbox=case["bbox"],
                           bbox_batch=case["bbox_batch"], bbox0=case["bbox0"],
                           bbox0_batch=case["bbox0_batch"], width=case["width"], height=case["height"])
    model.batch_feature_cache = None
    labels = data.bbox[:, 4]
    losses, outputs, out_labels = model(data, labels)
    feats = model.batch_feature_cache.clone()
    loss = losses["cross_entropy"]
    loss.backward()
    logits = torch.cat([o for frame in outputs for o in frame], 0) if outputs else torch.zeros(0, 2)
    targets = torch.cat([t for frame in out_labels for t in frame], 0) if out_labels else torch.zeros(0, dtype=torch.long)
    grads = {}
    for k, p in model.named_parameters():
        if k.startswith("dagr_model"):
            continue
        if p.grad is None:
            grads[k] = None
            continue
        g = p.grad.detach().flatten()
        grads[k] = dict(norm=g.norm().item(), sample=g[grad_sample_index(g.numel(), kw["seed"])].clone())
    gold = dict(kw=kw, feats=feats, logits=logits.detach(), targets=targets, loss=loss.detach(), grads=grads)
    torch.save(gold, os.path.join(HERE, name + ".pt"))
    print(name, "objects:", len(logits), "loss:", float(loss), "bytes:",
          os.path.getsize(os.path.join(HERE, name + ".pt")))


if __name__ == "__main__":
    torch.manual_seed(0)
    run_case("head_case_small", seed=11, num_graphs=3, tracks=4, nodes_per_clip=12, drop_prob=0.3,
             empty_clip=1, far_box=True)
    run_case("head_case_rol", seed=12, num_graphs=8, tracks=10, nodes_per_clip=35, drop_prob=0.2,
             empty_clip=None, far_box=True)
    run_case("head_case_dense", seed=13, num_graphs=6, tracks=30, nodes_per_clip=35, drop_prob=0.0,
             empty_clip=None, far_box=False)
