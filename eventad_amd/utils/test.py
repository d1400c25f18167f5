"""collect_predictions of the reference's test loop (utils/test.py:9-179) without per-object host syncs.

The reference pulls every score and label off the device with `.item()` (two syncs per object, plus one per
box for the first-anomaly scan).  Here a model that offers `predict_tensors(data, labels)` -- the MI355X
`EventADModel` does -- hands back the padded [B, 31] logit/target/valid tables of the batch and ONE transfer
per batch brings them to the host; any other model is driven through its `(losses, outputs, labels)` return
exactly as the reference does.  The returned dictionary has the reference's keys, value types and ordering,
including its indexing habits (the j-th NON-EMPTY clip of a batch is looked up as `sample_id[j]` /
`sequence[j]`; box i is paired with `sequence[i]` in the first-anomaly scan)."""
import numpy as np
import torch

from .utils import parse_sample_id

try:                                                           # progress bar as in the reference when tqdm is around
    from tqdm import tqdm
except Exception:                                              # pragma: no cover
    def tqdm(it, **kw):
        return it


def _batch_tables(model, data, labels):
    """-> per non-empty clip (in clip order): (scores list[float], labels list[int])."""
    if hasattr(model, "predict_tensors"):
        logits, valid, targets = model.predict_tensors(data, labels)
        packed = torch.cat([logits[..., 1].float(), valid.float(), targets.float()], dim=1).cpu().numpy()   # one D2H copy
        T = logits.shape[1]
        sc, va, tg = packed[:, :T], packed[:, T:2 * T] > 0.5, packed[:, 2 * T:]
        out = []
        for b in range(sc.shape[0]):
            ks = np.nonzero(va[b])[0]
            if ks.size:
                out.append(([float(np.float32(sc[b, k])) for k in ks], [int(tg[b, k]) for k in ks]))
        return out
    losses, batch_outputs, batch_labels = model(data, labels, testing=True)
    out = []
    for frame_outputs, frame_labels in zip(batch_outputs, batch_labels):
        scores = []
        for o in frame_outputs:
            if o.dim() == 1:
                scores.append(o[1].item())
            elif o.dim() == 2:
                scores.append(o[0, 1].item())
            else:
                print(f"Warning: Unexpected output dimensions: {o.dim()}, shape: {o.shape}")
                scores.append(None)
        out.append((scores, [l.item() for l in frame_labels]))
    return out


def collect_predictions(model, test_loader, device, video_toa=None, threshold=0.5):
    """Same arguments and result dictionary as the reference's collect_predictions."""
    from dagr.utils.buffers import format_data                 # the mirror registered by eventad_amd.compat.install()
    print("Starting to collect prediction results...")
    all_labels, all_scores, sample_ids = [], [], []
    frame_data, video_first_anomaly, video_predictions = {}, {}, {}
    valid_batch_count = skipped_batch_count = 0
    with torch.no_grad():
        for i, data in enumerate(tqdm(test_loader, desc="Testing Progress")):
            try:
                data = data.to(device)
                data = format_data(data)
                batch_ids = data.sample_id if hasattr(data, 'sample_id') else [f"batch_{i}_sample_{j}" for j in range(len(data.x))]
                if not hasattr(data, 'bbox') or data.bbox is None or (isinstance(data.bbox, torch.Tensor) and data.bbox.shape[0] == 0):
                    print(f"Test batch {i} has no valid bounding boxes, skipping")
                    skipped_batch_count += 1
                    continue
                labels = data.bbox[:, 4]
                if hasattr(data, 'sequence'):
                    video_ids = data.sequence
                    if isinstance(video_ids, str):
                        video_ids = [video_ids] * len(labels)
                else:
                    video_ids = [f"batch_{i}"] * len(labels)
                # first anomalous frame per video (utils/test.py:74-83): box i is paired with video_ids[i]
                anomalous = (labels > 0.5).cpu().numpy()                                       # one transfer for all boxes
                for label_idx, (is_anom, vid_id) in enumerate(zip(anomalous, video_ids)):
                    if is_anom and isinstance(batch_ids, list) and label_idx < len(batch_ids):
                        _, frame_id, _ = parse_sample_id(batch_ids[label_idx])
                        if vid_id not in video_first_anomaly or frame_id < video_first_anomaly[vid_id]:
                            video_first_anomaly[vid_id] = frame_id
                tables = _batch_tables(model, data, labels)
                valid_batch_count += 1
                for j, (scores, labs) in enumerate(tables):
                    vid_id = video_ids[j] if j < len(video_ids) else f"batch_{i}"
                    for k, (score, label) in enumerate(zip(scores, labs)):
                        if score is None:
                            continue
                        if isinstance(batch_ids, list) and j < len(batch_ids):
                            sample_id = f"{batch_ids[j]}_{k}"
                        else:
                            sample_id = f"batch_{i}_frame_{j}_obj_{k}"
                        _, frame_id, obj_id = parse_sample_id(sample_id)
                        all_scores.append(score)
                        all_labels.append(label)
                        sample_ids.append(sample_id)
                        fd = frame_data.setdefault(vid_id, {}).setdefault(frame_id, {'scores': [], 'labels': []})
                        fd['scores'].append(score)
                        fd['labels'].append(label)
                        vp = video_predictions.setdefault(vid_id, {})
                        vp[frame_id] = max(vp.get(frame_id, 0.0), score)
            except Exception as e:
                print(f"Error processing batch {i}: {e}")
                import traceback
                traceback.print_exc()
                skipped_batch_count += 1
                continue
    if not all_labels:
        raise RuntimeError("No valid predictions or labels collected, please check dataset and model")
    print(f"Testing complete: Valid batches {valid_batch_count}/{len(test_loader)}, Skipped batches {skipped_batch_count}")
    print(f"Collected predictions for {len(all_labels)} samples")
    all_preds = np.array([1 if score > threshold else 0 for score in all_scores])
    return {
        'all_preds': all_preds,
        'all_labels': np.array(all_labels),
        'all_scores': np.array(all_scores),
        'sample_ids': sample_ids,
        'frame_data': frame_data,
        'video_first_anomaly': video_first_anomaly,
        'video_predictions': video_predictions,
        'valid_batch_count': valid_batch_count,
        'skipped_batch_count': skipped_batch_count,
    }
