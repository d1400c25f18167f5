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

    def load_state_dict(self, state_dict, strict=True, **kw):
        kept = {k: v for k, v in state_dict.items() if not k.startswith("head.")}
        res = super().load_state_dict(kept, strict=strict, **kw)
        self.backbone._native = None
        return res

    def cache_luts(self, width, height, radius):
        """Reference: builds 21 LUTs of up to 14 GB (dagr.py:36-71).  Native: one 49..729-entry LUT
        pair for the event-level layer + concatenated kernel matrices for the pooled layers."""
        self.backbone.radius = float(radius)
        self.backbone.prepare()

    def forward(self, x, reset=True, return_targets=True, filtering=True):
        raise NotImplementedError("DAGR detection forward is outside the EventAD hot path (SURVEY.md section 2, #2)")

    def extract_features(self, x):
        self.eval()
        with torch.no_grad():
            features = self.backbone(x)
            return features[0] if isinstance(features, (list, tuple)) else features
