"""Data-parallel step on CPU with gloo, world_size 2: each rank runs the head on its own
clips, gradients of the used parameters are summed in ONE flat-bucket all-reduce, clipping happens after
the reduction.  Checked against a single process that adds the two ranks' gradients by hand."""
import os
import socket
from types import SimpleNamespace

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import seeded_head_state, make_head_case


def _case(rank):
    kw = dict(seed=20 + rank, num_graphs=3, tracks=6, nodes_per_clip=10, drop_prob=0.2, empty_clip=None, far_box=False)
    case = make_head_case(**kw)
    g = torch.Generator().manual_seed(100 + rank)
    feats = torch.randn(3, 2, 31, 64, generator=g)
    data = SimpleNamespace(**{k: case[k] for k in ("num_graphs", "bbox", "bbox_batch", "width", "height")})
    return feats, data


def _model():
    from eventad_amd.models.EventAD import EventADModel
    torch.manual_seed(0)                                           # the unused MultiheadAttention is default-initialised
    m = EventADModel(torch.nn.Module())
    m.load_state_dict(seeded_head_state(3), strict=False)
    return m.eval()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eventad_amd.train_step import TrainStep
    torch.set_num_threads(1)
    m = _model()
    ts = TrainStep(m, lr=1e-3, weight_decay=1e-5, grad_clip=1.0)
    assert ts.world == 2 and ts.bucket.numel() == 852066          # 3.41 MB bucket
    feats, data = _case(rank)
    loss, norm = ts.step_from_feats(feats, data)
    torch.save({"sd": {k: v.clone() for k, v in m.state_dict().items()}, "norm": float(norm), "loss": float(loss)},
               os.path.join(out, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_step_equals_summed_gradient_step(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k            # replicas stay identical
    assert abs(r0["norm"] - r1["norm"]) < 1e-6
    # single-process reference: sum of the two ranks' gradients, then clip + AdamW
    m = _model()
    params = [p for n, p in m.named_parameters() if not n.startswith("fusion_module.attention.")]
    total = 0.0
    for rank in range(2):
        feats, data = _case(rank)
        loss, _, _, _ = m.head(feats, data, data.bbox[:, 4])
        loss.backward()                                            # .grad accumulates = sum over ranks
        total += float(loss)
    norm = torch.nn.utils.clip_grad_norm_(params, 1.0)
    torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-5).step()
    assert abs(float(norm) - r0["norm"]) < 1e-4 * max(1.0, float(norm))
    for k, v in m.state_dict().items():
        assert torch.allclose(v, r0["sd"][k], atol=1e-6, rtol=1e-5), k
