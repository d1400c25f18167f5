"""Minimal batch container that duck-types the torch_geometric `Batch` attributes the
EventAD hot path reads (x, pos, t, batch, image, bbox, bbox_batch, bbox0,
bbox0_batch, width, height, time_window, num_graphs, sequence; input schema of
src/dagr/data/utils.py:6-19 + PyG collate with follow_batch=['bbox','bbox0'])."""
import torch


class EventBatch:
    _TENSOR_FIELDS = ("x", "pos", "t", "batch", "image", "bbox", "bbox_batch", "bbox0", "bbox0_batch",
                      "width", "height", "time_window")

    def __init__(self, **kw):
        self.image_feats = None          # optional: precomputed CNN feature maps (5 x [B,C,h,w])
        self.sequence = None
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device, non_blocking=True):
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device, non_blocking=non_blocking))
            elif isinstance(v, (list, tuple)) and v and all(isinstance(e, torch.Tensor) for e in v):
                setattr(self, k, [e.to(device, non_blocking=non_blocking) for e in v])
        return self

    def clone(self):
        out = EventBatch()
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                v = v.clone()
            elif isinstance(v, (list, tuple)) and v and all(isinstance(e, torch.Tensor) for e in v):
                v = [e.clone() for e in v]
            setattr(out, k, v)
        return out

    @staticmethod
    def from_clips(clips):
        """Collate per-clip dicts the way PyG's DataLoader(follow_batch=['bbox','bbox0']) does."""
        cat = lambda k: torch.cat([c[k] for c in clips], 0)
        idx = lambda k: torch.cat([torch.full((len(c[k]),), i, dtype=torch.long) for i, c in enumerate(clips)])
        b = EventBatch(x=cat("x"), pos=cat("pos"), t=cat("t"), batch=idx("x"),
                       bbox=cat("bbox"), bbox_batch=idx("bbox"), bbox0=cat("bbox0"), bbox0_batch=idx("bbox0"),
                       width=torch.tensor([c["width"] for c in clips]),
                       height=torch.tensor([c["height"] for c in clips]),
                       time_window=torch.tensor([c["time_window"] for c in clips]),
                       num_graphs=len(clips), sequence=[c.get("sequence", "synthetic") for c in clips])
        if "image" in clips[0] and clips[0]["image"] is not None:
            b.image = torch.cat([c["image"] for c in clips], 0)
        else:
            b.image = None
        if clips[0].get("image_feats") is not None:
            b.image_feats = [torch.cat([c["image_feats"][i] for c in clips], 0) for i in range(5)]
        return b
