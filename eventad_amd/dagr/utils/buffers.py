"""Mirror of the reference's `dagr.utils.buffers.format_data` (src/dagr/utils/buffers.py:33-44):
int -> float normalisation of pos / x / image, in place on the batch object.  Loader-format batches on the GPU
(pos i16, t i32, x i8) go through ONE HIP launch (eventad_format_events); anything else -- CPU tensors (tests, oracle),
other dtypes, a custom normalizer -- through the same elementwise torch ops as the reference."""
import torch


def format_data(data, normalizer=None):
    native = (normalizer is None and isinstance(data.pos, torch.Tensor) and data.pos.is_cuda and data.pos.dtype == torch.int16
              and data.pos.dim() == 2 and data.pos.shape[1] == 2 and data.t is not None and data.t.dtype == torch.int32
              and data.x.dtype == torch.int8
              and all(isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.int64 for v in (data.width, data.height, data.time_window)))
    if getattr(data, "image", None) is not None:
        data.image = data.image.float() / 255.0
    if native:
        from ..._lib import call, ptr
        n = int(data.pos.shape[0])
        pos, x = torch.empty((n, 3), dtype=torch.float32, device=data.pos.device), torch.empty(data.x.shape, dtype=torch.float32, device=data.pos.device)
        # inputs held in names until after the launch: a temporary made by .contiguous() would be freed (and its block
        # re-used by the next temporary) before the kernel reads it
        ins = [t.contiguous() for t in (data.pos, data.t, data.x, data.width, data.height, data.time_window)]
        call("eventad_format_events", ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), n, ptr(ins[3]), ptr(ins[4]), ptr(ins[5]), ptr(pos), ptr(x),
             torch.cuda.current_stream().cuda_stream)
        data.pos, data.x, data.t = pos, x, None
        return data
    if normalizer is None:
        normalizer = torch.stack([data.width[0], data.height[0], data.time_window[0]], dim=-1)
    data.pos = torch.cat([data.pos, data.t.view((-1, 1))], dim=-1) / normalizer
    data.t = None
    data.x = data.x.float()
    return data
