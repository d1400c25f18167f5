"""Data-parallel gradient exchange of the EventAD head: ONE flat fp32 bucket, ONE all-reduce(SUM) per step.

The reference has no distributed code.  Each rank runs the reference step (utils/train.py:100-152) on its own clips;
the gradients of the USED head parameters (852,066 floats = 3.41 MB; `fusion_module.attention.*` never receives a
gradient, models/EventAD.py:20) are summed over ranks before clip_grad_norm_ + AdamW.  The loss is the reference's SUM
over objects, so summed gradients equal one large-batch reference step with per-rank hidden-state chains; there is no
division by the world size.

This module is device agnostic on purpose: the layout of the bucket and the reduce -> update sequence are the same
objects on the GPU (backend "nccl" = RCCL over xGMI, the bucket is the HIP head's gradient buffer) and in the
world_size-2 gloo tests on CPU tensors."""
import torch
import torch.distributed as dist

# used head parameters in flat-buffer order (name, shape)
PARAMS = [
    ("gru_net_event.gru.weight_ih_l0", (768, 64)), ("gru_net_event.gru.weight_hh_l0", (768, 256)),
    ("gru_net_event.gru.bias_ih_l0", (768,)), ("gru_net_event.gru.bias_hh_l0", (768,)),
    ("gru_net_event.gru.weight_ih_l1", (768, 256)), ("gru_net_event.gru.weight_hh_l1", (768, 256)),
    ("gru_net_event.gru.bias_ih_l1", (768,)), ("gru_net_event.gru.bias_hh_l1", (768,)),
    ("gru_net_cor.gru.weight_ih_l0", (96, 4)), ("gru_net_cor.gru.weight_hh_l0", (96, 32)),
    ("gru_net_cor.gru.bias_ih_l0", (96,)), ("gru_net_cor.gru.bias_hh_l0", (96,)),
    ("fusion_module.event_proj.weight", (256, 256)), ("fusion_module.event_proj.bias", (256,)),
    ("fusion_module.coord_proj.weight", (256, 32)), ("fusion_module.coord_proj.bias", (256,)),
    ("fusion_module.fusion.0.weight", (256, 512)), ("fusion_module.fusion.0.bias", (256,)),
    ("fusion_module.fusion.3.weight", (2, 256)), ("fusion_module.fusion.3.bias", (2,)),
    ("soft_attention.weight", (256, 1)), ("soft_attention_cor.weight", (32, 1)),
]
USED_PARAMETER_COUNT = 852066


def flat_layout():
    """-> ({name: offset}, total floats): every tensor starts 16-byte aligned inside the flat buffer."""
    off, o = {}, 0
    for name, shape in PARAMS:
        off[name] = o
        n = 1
        for s in shape:
            n *= s
        o += (n + 3) // 4 * 4
    return off, o


def views(flat):
    """{name: view of `flat` with the parameter's shape} for a flat buffer laid out by flat_layout()."""
    off, total = flat_layout()
    assert flat.numel() == total and flat.dim() == 1
    out = {}
    for name, shape in PARAMS:
        n = 1
        for s in shape:
            n *= s
        out[name] = flat[off[name]:off[name] + n].view(shape)
    return out


class GradSync:
    """The step's only collective.  `bucket` is the flat gradient buffer itself (no staging copy)."""

    def __init__(self, bucket, process_group=None, force=False):
        self.bucket = bucket
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        self.bytes = 4 * bucket.numel()

    def allreduce(self):
        if self.active:
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)

    def reduce_and_update(self, update):
        """all-reduce(SUM) of the bucket, THEN `update()` (clip by the global norm of the summed gradient + AdamW):
        clipping before the reduction would clip per-rank norms, which is not the large-batch step."""
        self.allreduce()
        return update()
