"""Event-window selection and clip assembly for the DSEC/ROL `events_2x.h5` schema.

Schema (scripts/downsample_events.py:58-65, src/dagr/data/dsec_utils.py:80-120): `events/{x u2, y u2, p u1, t u4}`
(t in microseconds relative to `t_offset`, ascending), `t_offset` (i8) and `ms_to_idx` (for millisecond m, the index of
the first event with t >= 1000 m).  The reference reads the file with h5py inside `_load_events`; h5py is not part of
this image, so the functions here take the schema's arrays (any mapping: an open h5py file works as it is, so do
numpy arrays or memory-mapped files) and restate the arithmetic.  NOT pinned by reference outputs (`_load_events`
cannot be imported without h5py); pinned by the hand-derived cases of tests/test_input_side.py.
"""
import numpy as np
import torch

from ..batch import EventBatch


def _scalar(v):
    return v[()] if hasattr(v, "shape") and getattr(v, "shape", None) == () else v


def load_window(ev, t0, num_events=None, num_us=None, height=None, time_window=None):
    """Events of the window that starts at absolute time t0 [us] (dsec_utils.py:80-120).

    ev: mapping with 'events/x', 'events/y', 'events/t', 'events/p' (or 'x','y','t','p'), 't_offset', 'ms_to_idx'.
    Window = num_events events from the first event of t0's millisecond, or the events up to the millisecond
    t0 + num_us (either may be negative: the bounds are sorted).  Returns ((xy i16[n,2], t i32[n,1], p i8[n,1]), tq)
    with p in {-1,+1}, t re-based so that the query time tq (time of the last event of the window) maps to
    `time_window`, events with t <= 0 or y >= height dropped, and tq made absolute (i64)."""
    get = lambda k: ev["events/" + k] if ("events/" + k) in ev else ev[k]
    t_offset = _scalar(ev["t_offset"])
    ms_to_idx = ev["ms_to_idx"]
    ms = int((t0 - t_offset) / 1e3)
    idx0 = int(ms_to_idx[ms])
    idx1 = idx0
    if num_events is not None:
        idx1 = idx0 + num_events
    if num_us is not None:
        idx1 = int(ms_to_idx[ms + int(num_us / 1e3)])
    idx0, idx1 = sorted([idx0, idx1])
    idx0, idx1 = max(idx0, 0), max(idx1, 0)
    x, y, t, p = (np.asarray(get(k)[idx0:idx1]) for k in "xytp")
    tq = t[-1] if idx1 > idx0 else np.asarray(get("t")[max(idx1 - 1, idx0)])
    pol = 2 * p[..., None].astype("int8") - 1
    t_ev = t[..., None]
    xy = np.stack([x, y], axis=-1).astype("int16")
    tt = (time_window - tq + t_ev).astype("int32") if time_window is not None else np.array(tq, copy=True)
    tq = (tq + t_offset).astype("int64") if isinstance(tq, np.ndarray) else np.int64(tq + t_offset)
    mask = tt[:, 0] > 0 if tt.ndim == 2 else np.ones(len(xy), dtype=bool)
    if height is not None:
        mask &= xy[:, 1] < height
    return (xy[mask], tt[mask] if tt.ndim == 2 else tt, pol[mask]), tq


def to_clip(events, bbox, bbox0, width, height, time_window, image=None, sequence="clip"):
    """One loader sample in the path's input schema (src/dagr/data/utils.py:6-19 + dsec_data.py:166-170): x i8[n,1],
    pos i16[n,2], t i32[n], bbox/bbox0 f32[m,6] = (x, y, w, h, class, track), image u8[1,3,H,W] or None."""
    xy, t, p = events
    return dict(x=torch.from_numpy(np.ascontiguousarray(p)).reshape(-1, 1), pos=torch.from_numpy(np.ascontiguousarray(xy)),
                t=torch.from_numpy(np.ascontiguousarray(t)).reshape(-1).to(torch.int32),
                bbox=torch.as_tensor(bbox, dtype=torch.float32).reshape(-1, 6), bbox0=torch.as_tensor(bbox0, dtype=torch.float32).reshape(-1, 6),
                width=int(width), height=int(height), time_window=int(time_window), image=image, sequence=sequence)


def collate_pinned(clips):
    """PyG-style collation (follow_batch=['bbox','bbox0']) into ONE EventBatch whose tensors sit in pinned host memory,
    so `batch.to(device)` is a handful of asynchronous DMA copies instead of pageable-memory staging."""
    b = EventBatch.from_clips(clips)
    if torch.cuda.is_available():
        for k, v in list(b.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(b, k, v.pin_memory())
            elif isinstance(v, (list, tuple)) and v and all(isinstance(e, torch.Tensor) for e in v):
                setattr(b, k, [e.pin_memory() for e in v])
    return b
