"""Metric functions of the reference's test loop (utils/evaluation.py), numpy only.

Same names, arguments, printed lines and returned dict keys as the reference; the ROC / PR integrals
sklearn provides there (roc_curve + auc, average_precision_score) are restated below.  The reference rewrites
the scores of the upper half of the anomalous samples before integrating (evaluation.py:27-35, 86-94) and
subtracts 0.1 from AP; that behaviour is kept behind `reference_quirks` (default True = identical tables),
`reference_quirks=False` gives the plain metrics.  Pinned against the reference's own functions by
tests/golden/eval_golden.json (made by tests/golden/make_eval_golden.py)."""
import numpy as np


def _binary_curve(labels, scores):
    """Cumulative true/false positives at each distinct score, scores descending (sklearn's
    _binary_clf_curve: stable mergesort, ties collapsed to their last element)."""
    y = (np.asarray(labels) == 1).astype(np.float64) if np.asarray(labels).dtype != bool else np.asarray(labels, dtype=np.float64)
    s = np.asarray(scores, dtype=np.float64)
    order = np.argsort(s, kind="mergesort")[::-1]
    y, s = y[order], s[order]
    idx = np.r_[np.where(np.diff(s))[0], y.size - 1]
    tps = np.cumsum(y)[idx]
    fps = 1 + idx - tps
    return fps, tps


def roc_auc(labels, scores):
    """Area under the ROC curve by the trapezoidal rule (= sklearn auc(*roc_curve(labels, scores)[:2]))."""
    fps, tps = _binary_curve(labels, scores)
    fps, tps = np.r_[0, fps], np.r_[0, tps]
    if fps[-1] <= 0 or tps[-1] <= 0:
        return np.nan
    trapezoid = getattr(np, "trapezoid", None) or np.trapz
    return float(trapezoid(tps / tps[-1], fps / fps[-1]))


def average_precision(labels, scores):
    """sum_n (R_n - R_{n-1}) P_n over the distinct scores (= sklearn average_precision_score)."""
    fps, tps = _binary_curve(labels, scores)
    if tps[-1] <= 0:
        return 0.0
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    return float(np.sum(np.diff(np.r_[0, recall]) * precision))


def calculate_bbox_metrics(labels, scores, reference_quirks=True):
    """Object-level AUC and AP (evaluation.py:4-42)."""
    metrics = {}
    if len(labels) == 0 or len(scores) == 0:
        print("Warning: Empty labels or scores array")
        metrics['auc'] = np.nan
        metrics['ap'] = np.nan
        return metrics
    labels = np.asarray(labels)
    scores = np.copy(scores)
    if reference_quirks:
        anomaly_indices = np.where(labels > 0.5)[0]
        median_anomaly = np.median(scores[anomaly_indices])             # nan (with numpy's warning) when there is none
        for idx in anomaly_indices:
            if scores[idx] > median_anomaly:
                scores[idx] = scores[idx] * (1 - 2.7 * 0.5)
        metrics['ap'] = average_precision(labels, scores) - 0.1
    else:
        metrics['ap'] = average_precision(labels, scores)
    metrics['auc'] = roc_auc(labels, scores)
    print(f"AUC: {metrics['auc']:.4f}, AP: {metrics['ap']:.4f}")
    return metrics


def calculate_frame_metrics(frame_data, reference_quirks=True):
    """Frame-level AUC: a frame is anomalous if any object is, its score is the maximum object score
    (evaluation.py:44-113)."""
    metrics = {}
    frame_scores, frame_labels = [], []
    for video_id, frames in frame_data.items():
        for frame_id, data in frames.items():
            frame_labels.append(1 if any(label > 0.5 for label in data['labels']) else 0)
            frame_scores.append(max(data['scores']) if data['scores'] else 0.0)
    frame_scores = np.array(frame_scores)
    frame_labels = np.array(frame_labels)
    adjusted_scores = np.copy(frame_scores)
    anomaly_indices = np.where(frame_labels > 0.5)[0]
    if reference_quirks and len(anomaly_indices) > 0:
        median_anomaly = np.median(frame_scores[anomaly_indices])
        for idx in anomaly_indices:
            if frame_scores[idx] > median_anomaly:
                adjusted_scores[idx] = frame_scores[idx] * (1 - 1.5 * 0.5)
    if len(frame_labels) == 0:
        print("Warning: No valid frame-level data")
        metrics['auc_frame'] = np.nan
    elif len(np.unique(frame_labels)) < 2:
        print("Warning: Missing one or more categories in frame-level labels, cannot calculate complete frame-level metrics")
        metrics['auc_frame'] = np.nan
    else:
        metrics['auc_frame'] = roc_auc(frame_labels, adjusted_scores)
        print(f"AUC-Frame: {metrics['auc_frame']:.4f}")
    metrics['frame_scores'] = frame_scores
    metrics['frame_labels'] = frame_labels
    return metrics


THRESHOLDS = [0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]


def calculate_tta_metrics(video_predictions, video_first_anomaly, video_toa=None):
    """mTTA (evaluation.py:115-204): per video and threshold, frames between the LAST early detection
    (score >= threshold before the first anomalous frame) and that frame; mean per threshold, mean over the
    thresholds that have a value, / 30 fps."""
    metrics, metric = {}, {}
    tta_values = {t: [] for t in THRESHOLDS}
    for video_id, predictions in video_predictions.items():
        if video_toa and video_id in video_toa:
            first = video_toa[video_id]
        elif video_id in video_first_anomaly:
            first = video_first_anomaly[video_id]
        else:
            continue
        try:
            first = int(first)
        except (ValueError, TypeError):
            print(f"Warning: Cannot convert first anomalous frame '{first}' to integer, skipping video {video_id}")
            continue
        for t in THRESHOLDS:
            det = []
            for frame_id_str, score in predictions.items():
                try:
                    frame_id = int(frame_id_str)
                except (ValueError, TypeError):
                    continue
                if score >= t and frame_id < first:
                    det.append(frame_id)
            if det:
                tta_values[t].append(first - max(det))
    for t in THRESHOLDS:
        metric[f'tta_{t}'] = np.mean(tta_values[t]) if tta_values[t] else np.nan
    valid = [metric[f'tta_{t}'] for t in THRESHOLDS if not np.isnan(metric[f'tta_{t}'])]
    if valid:
        metrics['mtta'] = np.mean(np.array(valid) / 30.0)
        print(f"mTTA: {metrics['mtta']:.4f}")
    else:
        metrics['mtta'] = np.nan
        print("mTTA: N/A (no valid TTA values)")
    return metrics


def calculate_response_metrics(video_predictions, fps=579):
    """mRESPONSE (evaluation.py:206-315): per video and threshold, (first frame with score >= threshold at or
    after the first frame with score > 0, + 20 + 4, - that first frame) / 20 + 1/fps seconds."""
    metrics, metric = {}, {}
    response_values = {t: [] for t in THRESHOLDS}
    frame_processing_time = 1.0 / fps
    print(f"Processing time per frame: {frame_processing_time:.6f} seconds (based on FPS: {fps:.2f})")
    for video_id, predictions in video_predictions.items():
        sorted_predictions = []
        for frame_id_str, score in predictions.items():
            try:
                sorted_predictions.append((int(frame_id_str), score))
            except (ValueError, TypeError):
                continue
        sorted_predictions.sort(key=lambda x: x[0])
        if not sorted_predictions:
            continue
        first_nonzero_idx = next((i for i, (_, s) in enumerate(sorted_predictions) if s > 0), None)
        if first_nonzero_idx is None:
            continue
        for t in THRESHOLDS:
            threshold_idx = next((i for i, (_, s) in enumerate(sorted_predictions) if i >= first_nonzero_idx and s >= t), None)
            if threshold_idx is None:
                continue
            rate, init_frame = 20.0, 4                          # the reference rebinds `fps` to 20.0 here (evaluation.py:278)
            response_frames = sorted_predictions[threshold_idx][0] + rate + init_frame - sorted_predictions[first_nonzero_idx][0]
            response_values[t].append(response_frames / rate + frame_processing_time)
    for t in THRESHOLDS:
        if response_values[t]:
            metric[f'response_{t}'] = np.mean(response_values[t])
        else:
            metric[f'response_{t}'] = np.nan
            print(f"RESPONSE_{t}: N/A (no video data)")
    valid = [metric[f'response_{t}'] for t in THRESHOLDS if not np.isnan(metric[f'response_{t}'])]
    if valid:
        metrics['mresponse'] = np.mean(valid)
        print(f"mRESPONSE: {metrics['mresponse']:.4f} seconds")
    else:
        metrics['mresponse'] = np.nan
        print("mRESPONSE: N/A (no valid response time values)")
    return metrics
