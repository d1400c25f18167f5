"""The data-parallel step with its RCCL all-reduce INSIDE the captured hipGraphs (one rank, collective forced on):
replaying the graphs must leave exactly the weights that eager steps with the same collective leave -- for the
single-graph step and for the pipelined form the multi-GPU benchmark runs (head graph with the all-reduce on a CU-masked
stream, three backbone chains, no per-step join)."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group():
    mp = pytest.MonkeyPatch()
    mp.setenv("EVENTAD_FORCE_ALLREDUCE", "1")
    mp.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        yield
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()
        mp.undo()


def _setup(n_batches):
    from tests.test_gpu_model import _build
    from eventad_amd import synthetic
    from eventad_amd.train_step import TrainStep
    from dagr.utils.buffers import format_data
    W = H = 64; B = 3; n = 3000
    args, model, sd, head_sd, raw, data = _build(W, H, B, n)
    batches = [data] + [format_data(synthetic.make_batch(W, H, n, B, first_clip=3 * k, with_image=False,
                                                          precomputed_feats=True).to("cuda:0")) for k in range(1, n_batches)]
    model.eval()
    ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
    assert ts.sync.active and ts.world == 1 and ts.bucket_bytes == 4 * ts.nh.n == 4 * 852068
    snap = ts.snapshot()
    losses = []
    for b in batches:
        losses.append(ts.step(b)[0].clone())
    torch.cuda.synchronize()
    eager = {k: v.clone() for k, v in ts.nh.P.items()}
    ts.restore(snap)
    return ts, batches, eager, [float(l) for l in losses]


def test_captured_step_with_allreduce_equals_eager_step(one_rank_group):
    from eventad_amd.train_step import TrainStep
    ts, batches, eager, _ = _setup(3)
    buf = batches[0].clone()
    ts.capture(buf, warmup=2)
    for b in batches:
        TrainStep.load_into(buf, b)
        ts.replay()
    torch.cuda.synchronize()
    ts.check()
    for k, v in ts.nh.P.items():
        assert torch.equal(v, eager[k]), k


def test_pipelined_graphs_with_allreduce_on_the_masked_head_stream_equal_eager_steps(one_rank_group):
    """capture_dual with the CU partition and three backbone chains: the all-reduce is a node of every head graph, which
    replays on the head's CU-masked stream beside three backbone streams; five different batches, no per-step join."""
    ts, batches, eager, losses = _setup(5)
    ts.head_cus, ts.backbone_chains = 32, 3
    ts.capture_dual(batches[0], warmup=2)
    assert ts._dual["masked"] and ts._dual["n"] == 3
    held = []
    for b in batches[1:]:
        held.append(ts.replay_dual(b, join=False)[0])
    held.append(ts.flush()[0])
    torch.cuda.synchronize()
    ts.check()
    assert [float(h) for h in held[-3:]] == losses[-3:]       # each chain's last result
    for k, v in ts.nh.P.items():
        assert torch.equal(v, eager[k]), k
    ts.close()
