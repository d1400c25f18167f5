"""Metrics of the EventAD test loop, numpy only (interface of the reference's utils/evaluation.py:4-315).

Same function names, arguments and result keys as the reference so `test.py` can import them unchanged; the
bodies are vectorised restatements.  sklearn's roc_curve+auc and average_precision_score are replaced by
`roc_auc` / `average_precision` below.  Two habits of the reference are reproduced only while
`reference_quirks=True` (the default, so that tables match): before integrating it scales down the scores of the
anomalous samples above their median (x(1-1.35) for objects, x(1-0.75) for frames) and it subtracts 0.1 from AP.
Pinned by tests/golden/eval_golden.json, which tests/golden/make_eval_golden.py produces by calling the reference's
own module."""
import numpy as np

THRESHOLDS = (0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)
_TTA_FPS = 30.0            # frames -> seconds in mTTA (evaluation.py:196)
_RESP_RATE = 20.0          # the value the reference rebinds `fps` to inside its response loop (evaluation.py:278)
_RESP_INIT = 4             # its `init_frame`


def _cum_counts(labels, scores):
    """(false positives, true positives) accumulated over the distinct scores in descending order -- ties collapse
    onto their last element, stable sort, as sklearn's _binary_clf_curve does."""
    y = np.asarray(labels, dtype=np.float64) == 1.0
    s = np.asarray(scores, dtype=np.float64)
    order = np.argsort(s, kind="mergesort")[::-1]
    y, s = y[order].astype(np.float64), s[order]
    last = np.r_[np.flatnonzero(np.diff(s)), y.size - 1]
    tp = np.cumsum(y)[last]
    return 1.0 + last - tp, tp


def roc_auc(labels, scores):
    """Trapezoidal area under (fpr, tpr); nan when a class is missing."""
    fp, tp = _cum_counts(labels, scores)
    if fp[-1] <= 0 or tp[-1] <= 0:
        return float("nan")
    x, y = np.r_[0.0, fp] / fp[-1], np.r_[0.0, tp] / tp[-1]
    return float(np.sum(np.diff(x) * (y[1:] + y[:-1]) * 0.5))


def average_precision(labels, scores):
    """Step-wise AP: recall increments weighted by the precision at each distinct score."""
    fp, tp = _cum_counts(labels, scores)
    if tp[-1] <= 0:
        return 0.0
    return float(np.sum(np.diff(np.r_[0.0, tp / tp[-1]]) * (tp / (tp + fp))))


def _damp_upper_half(scores, positive, factor):
    """The reference's score rewrite: positives above the positives' median are multiplied by `factor`."""
    out = np.array(scores, dtype=np.float64, copy=True)
    idx = np.flatnonzero(positive)
    if idx.size:
        hit = idx[out[idx] > np.median(out[idx])]
        out[hit] = out[hit] * factor
    else:
        np.median(out[idx])                                    # the reference takes this median too (nan + numpy's warning)
    return out


def calculate_bbox_metrics(labels, scores, reference_quirks=True):
    """Object level: {'auc', 'ap'}."""
    if len(labels) == 0 or len(scores) == 0:
        print("[eval] no objects: AUC / AP undefined")
        return {'auc': np.nan, 'ap': np.nan}
    labels = np.asarray(labels)
    s = _damp_upper_half(scores, labels > 0.5, 1 - 2.7 * 0.5) if reference_quirks else np.asarray(scores, dtype=np.float64)
    out = {'ap': average_precision(labels, s) - (0.1 if reference_quirks else 0.0), 'auc': roc_auc(labels, s)}
    print("[eval] objects: AUC %.4f  AP %.4f" % (out['auc'], out['ap']))
    return out


def calculate_frame_metrics(frame_data, reference_quirks=True):
    """Frame level: a frame is anomalous when any of its objects is, its score is the largest object score.
    -> {'auc_frame', 'frame_scores', 'frame_labels'}."""
    frames = [f for per_video in frame_data.values() for f in per_video.values()]
    frame_labels = np.array([1 if any(l > 0.5 for l in f['labels']) else 0 for f in frames])
    frame_scores = np.array([max(f['scores']) if f['scores'] else 0.0 for f in frames])
    out = {'frame_scores': frame_scores, 'frame_labels': frame_labels}
    if frame_labels.size == 0 or np.unique(frame_labels).size < 2:
        print("[eval] frames: fewer than two classes, AUC-Frame undefined")
        out['auc_frame'] = np.nan
        return out
    s = _damp_upper_half(frame_scores, frame_labels > 0.5, 1 - 1.5 * 0.5) if reference_quirks else frame_scores
    out['auc_frame'] = roc_auc(frame_labels, s)
    print("[eval] frames: AUC %.4f" % out['auc_frame'])
    return out


def _int_keyed(predictions):
    """[(frame, score)] of the entries whose frame id converts to int, in dictionary order."""
    out = []
    for frame, score in predictions.items():
        try:
            out.append((int(frame), score))
        except (ValueError, TypeError):
            pass
    return out


def _mean_over_thresholds(per_threshold):
    means = [np.mean(v) for v in per_threshold.values() if v]
    return means


def calculate_tta_metrics(video_predictions, video_first_anomaly, video_toa=None):
    """mTTA: for every video with a known first anomalous frame (the `video_toa` table first, else the labels) and
    every threshold, the distance from the LAST earlier frame scoring >= threshold to that frame; averaged per
    threshold, then over the thresholds that have a value, in seconds at 30 fps.  -> {'mtta'}."""
    gaps = {t: [] for t in THRESHOLDS}
    for video, predictions in video_predictions.items():
        if video_toa and video in video_toa:
            onset = video_toa[video]
        elif video in video_first_anomaly:
            onset = video_first_anomaly[video]
        else:
            continue
        try:
            onset = int(onset)
        except (ValueError, TypeError):
            print("[eval] video %s: first anomalous frame %r is not an integer, skipped" % (video, onset))
            continue
        frames = _int_keyed(predictions)
        for t in THRESHOLDS:
            early = [f for f, s in frames if s >= t and f < onset]
            if early:
                gaps[t].append(onset - max(early))
    means = _mean_over_thresholds(gaps)
    mtta = np.mean(np.array(means) / _TTA_FPS) if means else np.nan
    print("[eval] mTTA %s" % ("%.4f s" % mtta if means else "undefined"))
    return {'mtta': mtta}


def calculate_response_metrics(video_predictions, fps=579):
    """mRESPONSE: per video and threshold, from the first frame with a positive score to the first frame at or after it
    scoring >= threshold, plus 20 + 4 frames, at 20 frames/s, plus the time to process one frame (1/fps).
    -> {'mresponse'}."""
    per_frame = 1.0 / fps
    print("[eval] one frame takes %.6f s at %.2f frames/s" % (per_frame, fps))
    delays = {t: [] for t in THRESHOLDS}
    for predictions in video_predictions.values():
        frames = sorted(_int_keyed(predictions), key=lambda fs: fs[0])
        start = next((i for i, (_, s) in enumerate(frames) if s > 0), None)
        if start is None:
            continue
        for t in THRESHOLDS:
            hit = next((i for i in range(start, len(frames)) if frames[i][1] >= t), None)
            if hit is not None:
                delays[t].append((frames[hit][0] + _RESP_RATE + _RESP_INIT - frames[start][0]) / _RESP_RATE + per_frame)
    means = _mean_over_thresholds(delays)
    mresp = np.mean(means) if means else np.nan
    print("[eval] mRESPONSE %s" % ("%.4f s" % mresp if means else "undefined"))
    return {'mresponse': mresp}
