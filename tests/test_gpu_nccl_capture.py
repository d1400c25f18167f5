"""The data-parallel step with its RCCL all-reduce INSIDE the captured hipGraph (one rank, collective forced on):
replaying the graph must leave exactly the weights that eager steps with the same collective leave."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_captured_step_with_allreduce_equals_eager_step(monkeypatch):
    from tests.test_gpu_model import _build
    from eventad_amd import synthetic
    from eventad_amd.train_step import TrainStep
    monkeypatch.setenv("EVENTAD_FORCE_ALLREDUCE", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        W = H = 64; B = 3; n = 3000
        args, model, sd, head_sd, raw, data = _build(W, H, B, n)
        from dagr.utils.buffers import format_data
        batches = [data] + [format_data(synthetic.make_batch(W, H, n, B, first_clip=3 * k, with_image=False,
                                                              precomputed_feats=True).to("cuda:0")) for k in (1, 2)]
        model.eval()
        ts = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
        assert ts.sync.active and ts.world == 1 and ts.bucket_bytes == 4 * ts.nh.n == 4 * 852068
        snap = ts.snapshot()
        for b in batches:
            ts.step(b)
        torch.cuda.synchronize()
        eager = {k: v.clone() for k, v in ts.nh.P.items()}
        ts.restore(snap)
        buf = batches[0].clone()
        ts.capture(buf, warmup=2)
        for b in batches:
            TrainStep.load_into(buf, b)
            ts.replay()
        torch.cuda.synchronize()
        ts.check()
        for k, v in ts.nh.P.items():
            assert torch.equal(v, eager[k]), k
    finally:
        dist.destroy_process_group()
