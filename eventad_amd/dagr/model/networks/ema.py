"""Mirror of `dagr.model.networks.ema.ModelEMA` (reference src/dagr/model/networks/ema.py:6-51).
EventAD only uses it as a holder: `.ema` = eval-mode deep copy with gradients off
(utils/model.py:28-39); `update()` follows ema.py:40-51 operation by operation (pinned by a fixture produced by the reference's class)."""
import copy
import math

import torch


class ModelEMA:
    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = copy.deepcopy(model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def update(self, model):
        self.updates += 1
        d = self.decay(self.updates)
        src = model.state_dict()
        for k, v in self.ema.state_dict().items():
            if v.dtype.is_floating_point:
                v *= d                                           # two roundings, as the reference (ema.py:47-51)
                v += (1.0 - d) * src[k].detach()
