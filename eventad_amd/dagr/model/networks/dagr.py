"""Mirror of `dagr.model.networks.dagr.DAGR` as EventAD uses it (reference
src/dagr/model/networks/dagr.py:14-130): constructor `DAGR(args, height, width)`, attributes
`.backbone/.head/.height/.width`, `cache_luts(width, height, radius)`, `extract_features(data)`.
The YOLOX detection head (GNNHead/CNNHead, dagr.py:151-320) is never executed by EventAD and is out
of scope; its checkpoint keys (`head.*`) are accepted and dropped by load_state_dict."""
import torch
import torch.nn as nn

from .net import Net


class DAGR(nn.Module):
    def __init__(self, args, height, width):
        super().__init__()
        self.conf_threshold, self.nms_threshold = 0.001, 0.65
        self.height, self.width = height, width
        self.batch_size = getattr(args, "batch_size", 1)
        self.backbone = Net(args, height=height, width=width)
        self.head = nn.Module()                      # detection head: out of scope (never run by EventAD)
        # Reference checkpoints carry the detection head's tensors (`head.*`; as `dagr_model.head.*` inside an EventAD
        # checkpoint, utils/model.py:116).  They are taken out of an incoming state dict before the strict key check --
        # in a hook, because a parent module's load never calls a child's load_state_dict -- kept as they came, and
        # written back by state_dict(), so a checkpoint loaded here and saved again still loads strictly in the reference.
        self._head_passthrough = {}
        self._register_load_state_dict_pre_hook(self._take_head_keys)
        self._register_state_dict_hook(DAGR._emit_head_keys)

    def _take_head_keys(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        mine = [k for k in state_dict if k.startswith(prefix + "head.")]
        if mine:
            self._head_passthrough = {k[len(prefix):]: state_dict.pop(k) for k in mine}

    @staticmethod
    def _emit_head_keys(module, state_dict, prefix, local_metadata):
        for k, v in module._head_passthrough.items():
            state_dict[prefix + k] = v

    def cache_luts(self, width, height, radius):
        """Reference: builds 21 LUTs of up to 14 GB (dagr.py:36-71).  Native: one 49..729-entry LUT
        pair for the event-level layer + concatenated kernel matrices for the pooled layers."""
        self.backbone.radius = float(radius)
        self.backbone.prepare()

    def forward(self, x, reset=True, return_targets=True, filtering=True):
        raise NotImplementedError("DAGR detection forward is outside the EventAD hot path (EventAD only calls extract_features, models/EventAD.py:335-403)")

    def train(self, mode=True):
        self._in_eval = not mode                               # (set by nn.Module.train for the whole tree below)
        return super().train(mode)

    def extract_features(self, x):
        # evaluation mode on every call, as the reference (dagr.py:108-110) -- but the walk over the ~900 modules only when
        # something switched modes since the last one
        if self.training or not getattr(self, "_in_eval", False):
            self.eval()
        with torch.no_grad():
            features = self.backbone(x)
            return features[0] if isinstance(features, (list, tuple)) else features
