"""Data-parallel gradient exchange on CPU with gloo, world_size 2: the SAME objects the GPU step uses
(eventad_amd/ddp.py: flat bucket layout, GradSync.reduce_and_update = all-reduce(SUM) then clip + AdamW) driven with
CPU tensors.  Each rank computes the head gradients of its own clips (torch autograd over the masked formulation --
test scaffolding: the product's head kernels need a GPU), writes them into the flat bucket and runs the shared
sequence.  Checked against a single process that adds the two ranks' gradients by hand."""
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


def _flat_step(m, feats, data, world_expected):
    """One rank's step through the shared bucket code; returns (loss, norm)."""
    from eventad_amd import ddp
    off, total = ddp.flat_layout()
    assert total == 852068 and sum(p.numel() for n, p in m.named_parameters() if n in dict(ddp.PARAMS)) == ddp.USED_PARAMETER_COUNT
    params = dict(m.named_parameters())
    flat_p, flat_g = torch.zeros(total), torch.zeros(total)
    P, G = ddp.views(flat_p), ddp.views(flat_g)
    for name, _ in ddp.PARAMS:                                     # parameters become views of the flat buffer, as on the GPU
        P[name].copy_(params[name].data)
        params[name].data = P[name]
    loss, _, _, _ = m.head(feats, data, data.bbox[:, 4])
    loss.backward()
    for name, _ in ddp.PARAMS:
        G[name].copy_(params[name].grad)
    sync = ddp.GradSync(flat_g)
    assert sync.world == world_expected and sync.bytes == 4 * total and sync.active == (world_expected > 1)
    opt = torch.optim.AdamW([flat_p.requires_grad_(True)], lr=1e-3, weight_decay=1e-5)

    def update():                                                  # clip by the norm of the SUMMED gradient, then AdamW
        flat_p.grad = flat_g
        norm = torch.nn.utils.clip_grad_norm_([flat_p], 1.0)
        opt.step()
        return norm

    norm = sync.reduce_and_update(update)
    return float(loss), float(norm)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _model()
    feats, data = _case(rank)
    loss, norm = _flat_step(m, feats, data, 2)
    torch.save({"sd": {k: v.detach().clone() for k, v in m.state_dict().items()}, "norm": norm, "loss": loss},
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
    for rank in range(2):
        feats, data = _case(rank)
        loss, _, _, _ = m.head(feats, data, data.bbox[:, 4])
        loss.backward()                                            # .grad accumulates = sum over ranks
    norm = torch.nn.utils.clip_grad_norm_(params, 1.0)
    torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-5).step()
    assert abs(float(norm) - r0["norm"]) < 1e-4 * max(1.0, float(norm))
    for k, v in m.state_dict().items():
        assert torch.allclose(v, r0["sd"][k], atol=1e-6, rtol=1e-5), k


def test_single_process_bucket_is_inactive_and_train_step_needs_a_gpu():
    from eventad_amd import ddp
    from eventad_amd.train_step import TrainStep
    sync = ddp.GradSync(torch.zeros(8))
    assert sync.world == 1 and not sync.active
    assert sync.reduce_and_update(lambda: 5) == 5
    import pytest
    with pytest.raises(RuntimeError):
        TrainStep(_model())                                        # CPU model: no fallback
