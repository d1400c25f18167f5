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


# ---- label side (eventad_amd/data/tracks.py): hand-derived cases
TRACK_DTYPE = np.dtype([("t", "<u8"), ("x", "<f8"), ("y", "<f8"), ("w", "<f8"), ("h", "<f8"), ("class_id", "<u1"),
                        ("class_confidence", "<f4"), ("track_id", "<u4")])


def _tracks(rows):
    a = np.zeros(len(rows), dtype=TRACK_DTYPE)
    for i, (t, x, y, w, h, c, tid) in enumerate(rows):
        a[i] = (t, x, y, w, h, c, 1.0, tid)
    return a


def test_track_filtering_and_frame_pairs():
    from eventad_amd.data import tracks as T
    assert T.construct_pairs(np.array([7, 3, 4, 9, 8])).tolist() == [[3, 4], [7, 8], [8, 9]]
    assert T.construct_pairs(np.array([5])).shape == (0, 2)
    mapping = T.compute_class_mapping(("car", "pedestrian"), ["pedestrian", "rider", "car", "bus"],
                                      dict(pedestrian="pedestrian", rider=None, car="car", bus="car"))
    assert mapping.tolist() == [1, -1, 0, 0]
    ts = np.array([100, 200, 300, 400, 500], dtype=np.uint64)
    tr = _tracks([(100, 30, 30, 60, 90, 2, 1), (200, 33, 30, 60, 90, 2, 1),          # car, frames 0 and 1
                  (200, 600, 10, 90, 60, 1, 2),                                      # rider at frame 1: class dropped
                  (300, 36, 30, 60, 90, 2, 1), (300, 300, 300, 9, 9, 0, 3),          # frame 2: car + a tiny pedestrian
                  (500, 900, 30, 60, 90, 2, 1)])                                     # frame 4 (frame 3 has no box)
    pairs, masks = T.filter_tracks({"seq": (tr, ts)}, image_width=320, image_height=215, class_remapping=mapping,
                                   min_bbox_height=5, min_bbox_diag=10, scale=3)
    # scale 3: the car is 20 x 30 (kept), the pedestrian 3 x 3 (too small), the rider has no class
    assert masks["seq"].tolist() == [True, True, False, True, False, True]
    assert pairs["seq"].tolist() == [[0, 1], [1, 2]]                                 # frames 0,1,2,4 are valid: runs of two
    scaled = T.crop_tracks(T.rescale_tracks(tr, 3), 320, 215)
    assert np.allclose(scaled["x"][:2], [10, 11]) and np.allclose(scaled["w"][0], 20) and np.allclose(scaled["h"][0], 30)
    assert scaled["x"][5] == 300 and scaled["w"][5] == 19                            # x 300, x2 = 320 clipped to 319
    # perfect tracks only: both frames of a pair hold the same ids and every box overlaps itself by IoU >= 0.1
    pairs2, _ = T.filter_tracks({"seq": (tr, ts)}, 320, 215, mapping, 5, 10, scale=3, only_perfect_tracks=True)
    assert pairs2["seq"].tolist() == [[0, 1], [1, 2]]
    jump = tr.copy(); jump["x"][3] = 600                                             # the car jumps in frame 2: IoU 0 with frame 1
    pairs3, _ = T.filter_tracks({"seq": (jump, ts)}, 320, 215, mapping, 5, 10, scale=3, only_perfect_tracks=True)
    assert pairs3["seq"].tolist() == [[0, 1]]
    a, b = _tracks([(1, 0, 0, 10, 10, 0, 1)]), _tracks([(2, 5, 0, 10, 10, 0, 1)])
    assert abs(float(T.compute_iou(a, b)[0]) - 50 / 150) < 1e-9


def test_sample_assembly_from_arrays():
    from eventad_amd.data import tracks as T
    mapping = np.array([1, -1, 0, 0])
    ts = np.array([1_000_000, 1_050_000], dtype=np.uint64)
    tr = _tracks([(1_000_000, 30, 30, 60, 90, 2, 5), (1_000_000, 300, 90, 90, 120, 0, 2),
                  (1_050_000, 60, 30, 60, 90, 2, 5), (1_050_000, 330, 90, 90, 120, 0, 2)])
    rng = np.random.default_rng(3)
    n = 400
    ev = dict(x=rng.integers(0, 320, n).astype(np.uint16), y=rng.integers(0, 240, n).astype(np.uint16),
              t=np.sort(rng.integers(1_000_000, 1_050_000, n)).astype(np.int64), p=rng.integers(0, 2, n).astype(np.uint8))
    clip = T.assemble_sample(tr, np.ones(4, bool), ts, (0, 1), ev, width=320, height=215, scale=3, class_remapping=mapping)
    keep = ev["y"] < 215
    assert clip["pos"].shape == (int(keep.sum()), 2) and clip["pos"].dtype == torch.int16
    assert int(clip["t"][-1]) == 1_000_000 and torch.equal(clip["t"], torch.from_numpy((1_000_000 + ev["t"][keep] - ev["t"][keep][-1]).astype(np.int32)))
    assert set(clip["x"].flatten().tolist()) <= {-1, 1}
    # bbox = second frame, bbox0 = first frame: (x, y, w, h, class, track) at a third of the sensor size, classes remapped
    assert torch.allclose(clip["bbox0"], torch.tensor([[10., 10, 20, 30, 0, 5], [100, 30, 30, 40, 1, 2]]))
    assert torch.allclose(clip["bbox"], torch.tensor([[20., 10, 20, 30, 0, 5], [110, 30, 30, 40, 1, 2]]))
    # a shorter window: events before t0 + 20 ms, boxes interpolated to that time (40 % of the way, sorted by track id)
    clip2 = T.assemble_sample(tr, np.ones(4, bool), ts, (0, 1), ev, 320, 215, 3, mapping, num_us=20_000)
    assert clip2["pos"].shape[0] == int((keep & (ev["t"] < 1_020_000)).sum())
    assert torch.allclose(clip2["bbox"], torch.tensor([[104., 30, 30, 40, 1, 2], [14, 10, 20, 30, 0, 5]]))
    from eventad_amd.data.events import collate_pinned
    b = collate_pinned([clip, clip2])
    assert b.num_graphs == 2 and b.bbox.shape == (4, 6) and b.bbox_batch.tolist() == [0, 0, 1, 1]
