"""Label side of the DSEC/ROL loader: track filtering, box preprocessing and sample assembly.

Restates the numpy arithmetic of src/dagr/data/dsec_utils.py:5-78,122-196 and src/dagr/data/dsec_data.py:18-43,112-186
over plain arrays.  `tracks` is the structured array of a sequence's `tracks.npy` (fields t, x, y, w, h, class_id,
track_id; boxes in sensor pixels), `image_timestamps` the sequence's frame times.  The reference reaches these arrays
through the un-vendored `dsec_det` package and reads events through h5py; neither is part of this image, so
`assemble_sample` takes the arrays themselves (an h5py file or `numpy.load(..., mmap_mode="r")` arrays work as they are).
NOT pinned by reference outputs (dsec_utils.py cannot be imported without h5py): pinned by the hand-derived cases of
tests/test_input_side.py."""
import numpy as np

from .events import to_clip


def construct_pairs(indices, n=2):
    """Runs of n consecutive image indices among `indices` -> i64[m, n] (dsec_utils.py:5-12)."""
    indices = np.sort(np.asarray(indices))
    if indices.size < n:
        return np.zeros((0, n), dtype=indices.dtype if indices.size else np.int64)
    rows = np.stack([indices[i:indices.size - n + 1 + i] for i in range(n)])
    mask = np.ones(rows.shape[1], dtype=bool)
    for i, row in enumerate(rows):
        mask &= rows[0] + i == row
    return rows[:, mask].T


def rescale_tracks(tracks, scale):
    """Boxes divided by the event down-sampling factor (dsec_utils.py:14-18)."""
    tracks = tracks.copy()
    for k in "xywh":
        tracks[k] = tracks[k] / scale
    return tracks


def crop_tracks(tracks, width, height):
    """Boxes clipped to [0, width-1] x [0, height-1] (dsec_utils.py:20-36)."""
    tracks = tracks.copy()
    x1, y1 = tracks["x"], tracks["y"]
    x2, y2 = x1 + tracks["w"], y1 + tracks["h"]
    x1, x2 = np.clip(x1, 0, width - 1), np.clip(x2, 0, width - 1)
    y1, y2 = np.clip(y1, 0, height - 1), np.clip(y2, 0, height - 1)
    tracks["x"], tracks["y"], tracks["w"], tracks["h"] = x1, y1, x2 - x1, y2 - y1
    return tracks


def compute_class_mapping(classes, all_classes, mapping):
    """dataset class id -> index into `classes`, -1 for classes that are dropped (dsec_utils.py:191-196)."""
    out = []
    for c in all_classes:
        m = mapping[c]
        out.append(classes.index(m) if m in classes else -1)
    return np.array(out)


def map_classes(class_ids, old_to_new_mapping):
    new = np.asarray(old_to_new_mapping)[class_ids]
    return new, new > -1


def filter_small_bboxes(w, h, bbox_height=20, bbox_diag=30):
    """Keep boxes with diagonal > bbox_diag and both sides > bbox_height (dsec_utils.py:43-48)."""
    diag = np.sqrt(h ** 2 + w ** 2)
    return (diag > bbox_diag) & (w > bbox_height) & (h > bbox_height)


def compute_iou(track0, track1):
    """Row-wise IoU of two equally long box arrays (dsec_utils.py:149-168)."""
    x1, x2 = track0["x"], track0["x"] + track0["w"]
    y1, y2 = track0["y"], track0["y"] + track0["h"]
    x1g, x2g = track1["x"], track1["x"] + track1["w"]
    y1g, y2g = track1["y"], track1["y"] + track1["h"]
    xk1, yk1 = np.maximum(x1, x1g), np.maximum(y1, y1g)
    xk2, yk2 = np.minimum(x2, x2g), np.minimum(y2, y2g)
    inter = np.zeros_like(x1)
    m = (yk2 > yk1) & (xk2 > xk1)
    inter[m] = (xk2[m] - xk1[m]) * (yk2[m] - yk1[m])
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter + 1e-9
    return inter / union


def is_invalid_track(track):
    """A frame pair is invalid when the two frames do not hold the same track ids or a box moves so far that its IoU
    with itself drops below 0.1 (dsec_utils.py:134-147)."""
    track = [tr[tr["track_id"].argsort()] for tr in track]
    first = track[0]
    for cur in track[1:]:
        if len(first) != len(cur) or not (cur["track_id"] == first["track_id"]).all():
            return True
        if len(first) and np.min(compute_iou(first, cur)) < 0.10:
            return True
    return False


def compute_img_idx_to_track_idx(t, t_query):
    """For every column of t_query (frame times of the pairs' first / second frame): [start, end) rows of the time-sorted
    track array per queried time -> i64[n_columns, 2, m] (dsec_utils.py:171-183)."""
    def one(tq):
        uniq, counts = np.unique(t, return_counts=True)
        idx = np.concatenate([np.array([0]), counts]).cumsum()
        spans = np.stack([idx[:-1], idx[1:]], axis=-1)
        return spans[np.isin(uniq, tq)].T
    return np.stack([one(tq) for tq in np.asarray(t_query).T])


def filter_by_only_perfect_tracks(tracks, img_idx_to_track_idx, tracks_mask=None):
    """dsec_utils.py:122-132, literally: the array [frame of the pair][start | end][pair] is unpacked along its FIRST axis,
    so slice j of pair i runs from row (first frame's start | end) to row (second frame's start | end) -- the rows of the
    first and of the second frame when the two frames are neighbours in the time-sorted track array."""
    i0, i1 = img_idx_to_track_idx
    mask = np.ones(i0.shape[1], dtype=bool)
    for i in range(i0.shape[1]):
        track = [tracks[i0[j][i]:i1[j][i]] for j in range(len(i0))]
        if tracks_mask is not None:
            track = [t[tracks_mask[i0[j][i]:i1[j][i]]] for j, t in enumerate(track)]
        mask[i] = not is_invalid_track(track)
    return mask


def filter_tracks(sequences, image_width, image_height, class_remapping, min_bbox_height=0, min_bbox_diag=0, scale=1,
                  only_perfect_tracks=False):
    """sequences: {name: (tracks, image_timestamps)}.  -> ({name: i64[m,2] pairs of consecutive frames that both carry a
    surviving box}, {name: bool mask over the sequence's track rows}) (dsec_utils.py:50-78)."""
    pairs, masks = {}, {}
    for name, (tracks, image_timestamps) in sequences.items():
        scaled = crop_tracks(rescale_tracks(tracks, scale), image_width, image_height)
        _, class_mask = map_classes(scaled["class_id"], class_remapping)
        final = filter_small_bboxes(scaled["w"], scaled["h"], min_bbox_height, min_bbox_diag) & class_mask
        valid = np.unique(np.nonzero(np.isin(image_timestamps, scaled[final]["t"]))[0])
        p = construct_pairs(valid, 2)
        if only_perfect_tracks and len(p):
            brackets = np.asarray(image_timestamps)[p]
            keep = filter_by_only_perfect_tracks(scaled, compute_img_idx_to_track_idx(tracks["t"], brackets), tracks_mask=final)
            p = p[keep]
        pairs[name], masks[name] = p, final
    return pairs, masks


def tracks_to_array(tracks):
    """-> f[m,6] = (x, y, w, h, class_id, track_id): the `bbox` layout of the path (dsec_data.py:18-19)."""
    return np.stack([tracks["x"], tracks["y"], tracks["w"], tracks["h"], tracks["class_id"], tracks["track_id"]], axis=1)


def interpolate_tracks(detections_0, detections_1, t):
    """Boxes at time t between two frames that hold the same tracks (dsec_data.py:23-43)."""
    assert len(detections_1) == len(detections_0)
    if len(detections_0) == 0:
        return detections_1
    t0, t1 = detections_0["t"][0], detections_1["t"][0]
    assert t0 < t1
    detections_0 = detections_0[detections_0["track_id"].argsort()]
    detections_1 = detections_1[detections_1["track_id"].argsort()]
    r = (t - t0) / (t1 - t0)
    out = detections_0.copy()
    for k in "xywh":
        out[k] = detections_0[k] * (1 - r) + detections_1[k] * r
    return out


def preprocess_detections(detections, scale, width, height, class_remapping):
    """dsec_data.py:112-116."""
    detections = crop_tracks(rescale_tracks(detections, scale), width, height)
    detections["class_id"], _ = map_classes(detections["class_id"], class_remapping)
    return detections


def preprocess_events(events, height, time_window):
    """Events below the cropped height, times re-based so that the last event sits at `time_window`, polarity in {-1, +1}
    (dsec_data.py:118-124).  events: dict of x, y, t, p arrays."""
    mask = events["y"] < height
    events = {k: np.asarray(v)[mask] for k, v in events.items()}
    if len(events["t"]) > 0:
        events["t"] = time_window + events["t"].astype(np.int64) - np.int64(events["t"][-1])
    events["p"] = 2 * events["p"].reshape((-1, 1)).astype("int8") - 1
    return events


def assemble_sample(tracks, track_mask, image_timestamps, pair, events, width, height, scale, class_remapping, time_window=1000000,
                    num_us=-1, min_bbox_height=0, min_bbox_diag=0, image=None, sequence="sequence", no_eval=False):
    """One loader sample as DSEC.__getitem__ builds it (dsec_data.py:126-186): boxes of the pair's two frames (bbox0 = first,
    bbox = second, or interpolated to t0 + num_us), the events between the two frames (dict of x, y, t, p with absolute t,
    already cut to the frame interval by the caller's reader), preprocessing, small-box filter.  -> a clip dict for
    EventBatch.from_clips / collate_pinned (eventad_amd/data/events.py::to_clip)."""
    i0, i1 = pair
    ts0, ts1 = np.asarray(image_timestamps)[[i0, i1]]
    kept = tracks[track_mask]
    det0, det1 = kept[kept["t"] == ts0], kept[kept["t"] == ts1]
    det0 = preprocess_detections(det0, scale, width, height, class_remapping)
    det1 = preprocess_detections(det1, scale, width, height, class_remapping)
    if num_us >= 0:
        ts1 = ts0 + num_us
        events = {k: np.asarray(v)[np.asarray(events["t"]) < ts1] for k, v in events.items()}
        if not no_eval:
            det1 = interpolate_tracks(det0, det1, ts1)
    ev = preprocess_events(events, height, time_window)
    b1, b0 = tracks_to_array(det1).astype(np.float32), tracks_to_array(det0).astype(np.float32)
    b1 = b1[filter_small_bboxes(b1[:, 2], b1[:, 3], min_bbox_height, min_bbox_diag)]
    b0 = b0[filter_small_bboxes(b0[:, 2], b0[:, 3], min_bbox_height, min_bbox_diag)]
    xy = np.stack([ev["x"], ev["y"]], axis=-1).astype("int16")
    return to_clip((xy, ev["t"].astype("int32"), ev["p"]), b1, b0, width, height, time_window, image=image, sequence=sequence)
