"""Mirror of the reference's `dagr.utils.buffers.format_data` (src/dagr/utils/buffers.py:33-44):
int -> float normalisation of pos / x / image, in place on the batch object.  Pure elementwise
torch ops on whatever device the batch lives on (the kernels downstream take the result)."""
import torch


def format_data(data, normalizer=None):
    if normalizer is None:
        normalizer = torch.stack([data.width[0], data.height[0], data.time_window[0]], dim=-1)
    if getattr(data, "image", None) is not None:
        data.image = data.image.float() / 255.0
    data.pos = torch.cat([data.pos, data.t.view((-1, 1))], dim=-1) / normalizer
    data.t = None
    data.x = data.x.float()
    return data
