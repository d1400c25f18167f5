"""Input side: window selection in the events_2x.h5 schema, clip assembly, collation, and the HIP
format_data against the torch one."""
import numpy as np
import pytest
import torch

from eventad_amd.data.events import load_window, to_clip, collate_pinned


def _file(n_ms=50, per_ms=4, t_offset=1_000_000, seed=0):
    """A tiny recording: per_ms events in every millisecond, ascending t, known ms_to_idx."""
    rng = np.random.default_rng(seed)
    t = np.sort(np.concatenate([ms * 1000 + rng.integers(0, 1000, per_ms) for ms in range(n_ms)])).astype(np.uint32)
    n = t.size
    ev = {"events/x": rng.integers(0, 64, n).astype(np.uint16), "events/y": rng.integers(0, 48, n).astype(np.uint16),
          "events/p": rng.integers(0, 2, n).astype(np.uint8), "events/t": t, "t_offset": np.int64(t_offset),
          "ms_to_idx": np.searchsorted(t, np.arange(n_ms + 1) * 1000, side="left").astype(np.uint64)}
    return ev


def test_window_by_duration_matches_hand_derivation():
    ev = _file()
    t0 = 1_000_000 + 10_000                                   # absolute time of millisecond 10
    (xy, t, p), tq = load_window(ev, t0, num_us=5_000, height=40, time_window=1_000_000)
    lo, hi = int(ev["ms_to_idx"][10]), int(ev["ms_to_idx"][15])
    raw_t = ev["events/t"][lo:hi].astype(np.int64)
    keep = ev["events/y"][lo:hi] < 40
    want_t = 1_000_000 - raw_t[-1] + raw_t                    # last event of the window lands on time_window
    keep &= want_t > 0
    assert tq == raw_t[-1] + 1_000_000 and tq.dtype == np.int64
    assert np.array_equal(t[:, 0], want_t[keep].astype(np.int32)) and t.dtype == np.int32 and t[-1, 0] == 1_000_000
    assert np.array_equal(xy, np.stack([ev["events/x"][lo:hi], ev["events/y"][lo:hi]], -1)[keep].astype(np.int16))
    assert set(np.unique(p)) <= {-1, 1} and np.array_equal(p[:, 0], (2 * ev["events/p"][lo:hi].astype(np.int8) - 1)[keep])


def test_window_by_count_backwards_and_empty():
    ev = _file()
    t0 = 1_000_000 + 20_000
    (xy, t, p), tq = load_window(ev, t0, num_events=-12, time_window=1_000_000)       # the 12 events BEFORE millisecond 20
    hi = int(ev["ms_to_idx"][20])
    assert len(xy) == 12 and tq == int(ev["events/t"][hi - 1]) + 1_000_000
    (xy, t, p), tq = load_window(ev, t0, num_events=0, time_window=1_000_000)         # empty window: tq = t[max(idx1-1, idx0)]
    assert len(xy) == 0 and tq == int(ev["events/t"][hi]) + 1_000_000                 # = the first event of the millisecond


def test_clip_assembly_and_collation():
    ev = _file()
    clips = []
    for i, ms in enumerate((5, 12, 30)):
        events, tq = load_window(ev, 1_000_000 + ms * 1000, num_us=4_000, height=48, time_window=1_000_000)
        bbox = np.array([[1, 2, 10, 10, i % 2, 3], [5, 5, 8, 8, 0, 7]], np.float32)[: 2 - (i == 1)]
        clips.append(to_clip(events, bbox, bbox, 64, 48, 1_000_000, sequence="rec_%d" % i))
    b = collate_pinned(clips)
    n = [len(c["pos"]) for c in clips]
    assert b.num_graphs == 3 and b.pos.dtype == torch.int16 and b.t.dtype == torch.int32 and b.x.dtype == torch.int8
    assert b.pos.shape == (sum(n), 2) and torch.equal(b.batch, torch.repeat_interleave(torch.arange(3), torch.tensor(n)))
    assert torch.equal(b.bbox_batch, torch.tensor([0, 0, 1, 2, 2])) and b.sequence == ["rec_0", "rec_1", "rec_2"]
    assert all(int(clips[i]["t"][-1]) == 1_000_000 for i in range(3))


@pytest.mark.gpu
def test_hip_format_data_equals_torch_format_data():
    """eventad_format_events (one launch: int16/int32/int8 -> normalised fp32 pos[N,3] and x) against the torch
    restatement of src/dagr/utils/buffers.py:33-44 -- bit for bit."""
    import eventad_amd.compat
    eventad_amd.compat.install()
    from dagr.utils.buffers import format_data
    from eventad_amd import synthetic
    raw = synthetic.make_batch(256, 256, 20000, 3, with_image=True, precomputed_feats=False)
    want = format_data(raw.clone())                                                  # CPU tensors: torch path
    got = format_data(raw.clone().to("cuda"))                                        # CUDA tensors: HIP kernel
    assert got.pos.dtype == torch.float32 and got.t is None
    assert torch.equal(got.pos.cpu(), want.pos) and torch.equal(got.x.cpu(), want.x)
    assert torch.allclose(got.image.cpu(), want.image, atol=1e-7, rtol=0)          # image / 255 stays a torch op on either device


@pytest.mark.gpu
def test_grouped_copy_moves_every_segment_exactly():
    """eventad_copy_group: several device-to-device copies in one launch -- aligned and unaligned ends, sizes that are not
    multiples of 16 bytes, an empty segment; bytes around every destination stay untouched."""
    import ctypes
    from eventad_amd._lib import call
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    sizes = [4_800_000, 1_600_004, 37, 0, 16, 3_200_001, 5]
    offs = [0, 16, 3, 0, 8, 1, 7]                             # byte offsets of source and destination inside their buffers
    srcs, dsts, want = [], [], []
    for i, (n, o) in enumerate(zip(sizes, offs)):
        a = torch.randint(0, 256, (n + 64,), dtype=torch.uint8, generator=g).to(dev)
        b = torch.randint(0, 256, (n + 64,), dtype=torch.uint8, generator=g).to(dev)
        w = b.clone()
        q = o + (5 if i % 2 else 0)                           # even segments: both ends equally aligned (16-byte path at o = 0, 16)
        w[q: q + n] = a[o: o + n]
        srcs.append(a[o: o + n]); dsts.append(b[q: q + n]); want.append((b, w))
    k = len(sizes)
    sp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in srcs])
    dp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in dsts])
    nb = (ctypes.c_uint64 * k)(*sizes)
    call("eventad_copy_group", k, ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p), ctypes.cast(nb, ctypes.c_void_p),
         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for b, w in want:
        assert torch.equal(b, w)
    with pytest.raises(Exception):
        call("eventad_copy_group", 17, ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p), ctypes.cast(nb, ctypes.c_void_p), 0)
