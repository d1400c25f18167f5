"""collect_predictions of the EventAD test loop (interface of the reference's utils/test.py:9-179) without per-object
host syncs.

The reference reads every score and label off the device with `.item()` (two syncs per object, one more per box for the
first-anomaly scan).  Here a model that offers `predict_tensors(data, labels)` -- the MI355X `EventADModel` does --
returns the padded [B, 31] logit / target / valid tables of the batch and ONE transfer per batch brings them to the
host; any other model is driven through its `(losses, outputs, labels)` return.  The result dictionary has the
reference's keys, value types and ordering, including its indexing habits: the j-th NON-EMPTY clip of a batch is
looked up as `sample_id[j]` / `sequence[j]`, and box i is paired with `sequence[i]` when the first anomalous frame of
a video is searched.  Pinned by tests/golden/eval_golden.json (made by running the reference's function)."""
import traceback

import numpy as np
import torch

from .utils import parse_sample_id

try:
    from tqdm import tqdm
except Exception:                                              # pragma: no cover
    def tqdm(it, **kw):
        return it


def _tables_of(model, data, labels):
    """[(scores, labels)] per non-empty clip, clip order; score = logit of class 1 as a Python float (None = unusable)."""
    if hasattr(model, "predict_tensors"):
        logits, valid, targets = model.predict_tensors(data, labels)
        host = torch.cat([logits[..., 1].float(), valid.float(), targets.float()], dim=1).cpu().numpy()    # the one copy
        T = logits.shape[1]
        score, keep, target = host[:, :T], host[:, T:2 * T] > 0.5, host[:, 2 * T:]
        return [([float(np.float32(score[b, k])) for k in ks], [int(target[b, k]) for k in ks])
                for b in range(score.shape[0]) for ks in [np.flatnonzero(keep[b])] if ks.size]
    _, outputs, targets = model(data, labels, testing=True)
    tables = []
    for outs, tgts in zip(outputs, targets):
        scores = []
        for o in outs:
            if o.dim() in (1, 2):
                scores.append((o[1] if o.dim() == 1 else o[0, 1]).item())
            else:
                print("[collect] output of rank %d ignored (shape %s)" % (o.dim(), tuple(o.shape)))
                scores.append(None)
        tables.append((scores, [t.item() for t in tgts]))
    return tables


class _Collector:
    def __init__(self):
        self.scores, self.labels, self.ids = [], [], []
        self.frame_data, self.first_anomaly, self.video_predictions = {}, {}, {}

    def note_anomalies(self, anomalous, video_ids, batch_ids):
        """utils/test.py:74-83: box i <-> video_ids[i], frame id parsed from batch_ids[i]."""
        if not isinstance(batch_ids, list):
            return
        for i, (flag, video) in enumerate(zip(anomalous, video_ids)):
            if flag and i < len(batch_ids):
                frame = parse_sample_id(batch_ids[i])[1]
                if video not in self.first_anomaly or frame < self.first_anomaly[video]:
                    self.first_anomaly[video] = frame

    def add(self, sample_id, video, score, label):
        frame = parse_sample_id(sample_id)[1]
        self.scores.append(score); self.labels.append(label); self.ids.append(sample_id)
        cell = self.frame_data.setdefault(video, {}).setdefault(frame, {'scores': [], 'labels': []})
        cell['scores'].append(score); cell['labels'].append(label)
        per_video = self.video_predictions.setdefault(video, {})
        per_video[frame] = max(per_video.get(frame, 0.0), score)


def collect_predictions(model, test_loader, device, video_toa=None, threshold=0.5):
    """Same arguments and result dictionary as the reference's collect_predictions."""
    from dagr.utils.buffers import format_data                 # the mirror registered by eventad_amd.compat.install()
    print("[collect] running the test set ...")
    acc = _Collector()
    done = skipped = 0
    with torch.no_grad():
        for i, data in enumerate(tqdm(test_loader, desc="Testing Progress")):
            try:
                data = format_data(data.to(device))
                boxes = getattr(data, 'bbox', None)
                if boxes is None or (isinstance(boxes, torch.Tensor) and boxes.shape[0] == 0):
                    print("[collect] batch %d has no boxes, skipped" % i)
                    skipped += 1
                    continue
                batch_ids = data.sample_id if hasattr(data, 'sample_id') else ["batch_%d_sample_%d" % (i, j) for j in range(len(data.x))]
                labels = boxes[:, 4]
                video_ids = getattr(data, 'sequence', None)
                if video_ids is None:
                    video_ids = ["batch_%d" % i] * len(labels)
                elif isinstance(video_ids, str):
                    video_ids = [video_ids] * len(labels)
                acc.note_anomalies((labels > 0.5).cpu().numpy(), video_ids, batch_ids)            # one transfer for all boxes
                tables = _tables_of(model, data, labels)
                done += 1
                for j, (scores, targets) in enumerate(tables):
                    video = video_ids[j] if j < len(video_ids) else "batch_%d" % i
                    named = isinstance(batch_ids, list) and j < len(batch_ids)
                    for k, (score, target) in enumerate(zip(scores, targets)):
                        if score is not None:
                            acc.add("%s_%d" % (batch_ids[j], k) if named else "batch_%d_frame_%d_obj_%d" % (i, j, k), video, score, target)
            except Exception as e:                             # the reference logs and moves on (utils/test.py:147-152)
                print("[collect] batch %d failed: %s" % (i, e))
                traceback.print_exc()
                skipped += 1
    if not acc.labels:
        raise RuntimeError("No valid predictions or labels collected, please check dataset and model")
    print("[collect] %d of %d batches used, %d skipped, %d objects" % (done, len(test_loader), skipped, len(acc.labels)))
    return {
        'all_preds': np.array([1 if s > threshold else 0 for s in acc.scores]),
        'all_labels': np.array(acc.labels),
        'all_scores': np.array(acc.scores),
        'sample_ids': acc.ids,
        'frame_data': acc.frame_data,
        'video_first_anomaly': acc.first_anomaly,
        'video_predictions': acc.video_predictions,
        'valid_batch_count': done,
        'skipped_batch_count': skipped,
    }
