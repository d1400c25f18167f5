"""Seeded inputs shared by tests/golden/make_eval_golden.py (which feeds them to the REFERENCE's utils/evaluation.py
and utils/test.py) and tests/test_eval_loop.py (which feeds them to eventad_amd/utils/)."""
import types
import numpy as np
import torch


def clean(o):
    """numpy / nan -> JSON-able (nan as the string 'nan'; dict keys as strings)."""
    if isinstance(o, dict):
        return {str(k): clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [clean(v) for v in o]
    if isinstance(o, np.ndarray):
        return clean(o.tolist())
    if isinstance(o, (np.floating, float)):
        return "nan" if np.isnan(o) else float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    return o


def _videos(rng, n_videos, n_frames, p_anom, quantise):
    frame_data, vp, first = {}, {}, {}
    labels, scores = [], []
    for v in range(n_videos):
        vid = "video_%02d" % v
        onset = int(rng.integers(n_frames // 3, n_frames)) if rng.random() < p_anom else None
        for f in range(0, n_frames, int(rng.integers(1, 3))):
            n_obj = int(rng.integers(1, 5))
            sc = rng.normal(0.3 if onset is None or f < onset else 0.8, 0.35, n_obj)
            if quantise:
                sc = np.round(sc, 1)                              # many tied scores
            lb = [1 if (onset is not None and f >= onset and rng.random() < 0.6) else 0 for _ in range(n_obj)]
            sc = [float(np.float32(s)) for s in sc]
            frame_data.setdefault(vid, {})[f] = {"scores": sc, "labels": lb}
            vp.setdefault(vid, {})[f] = max(0.0, max(sc))
            labels += lb; scores += sc
            if any(lb) and vid not in first:
                first[vid] = f
    return frame_data, vp, first, labels, scores


def metric_cases():
    cases = {}
    for name, seed, nv, nf, pa, q in (("small", 1, 3, 12, 1.0, False), ("ties", 2, 6, 40, 0.7, True), ("large", 3, 25, 90, 0.6, False),
                                      ("one_class_frames", 4, 2, 6, 0.0, False)):
        rng = np.random.default_rng(seed)
        fd, vp, first, labels, scores = _videos(rng, nv, nf, pa, q)
        if name == "one_class_frames":                            # object labels still need both classes for the bbox metrics
            labels = list(labels); labels[0] = 1
        toa = {"video_00": 7, "video_03": "11"} if name != "small" else None
        cases[name] = dict(labels=labels, scores=scores, frame_data=fd, video_predictions=vp, video_first_anomaly=first,
                           video_toa=toa, fps=579 if name != "large" else 131.5)
    return cases


class StubBatch(types.SimpleNamespace):
    def to(self, device):
        return self


class StubModel:
    """Returns prescribed per-batch tables; nested=True through the reference's (losses, outputs, labels) protocol,
    nested=False through predict_tensors (the sync-free protocol)."""
    def __init__(self, tables, nested):
        self.tables, self.nested, self.i = tables, nested, 0
        if not nested:
            self.predict_tensors = self._predict

    def _next(self):
        t = self.tables[self.i]; self.i += 1
        return t

    def _predict(self, data, labels=None):
        logits, valid, targets = self._next()
        return logits, valid, targets

    def __call__(self, data, labels=None, testing=False):
        logits, valid, targets = self._next()
        outs, labs = [], []
        for b in range(logits.shape[0]):
            ks = torch.nonzero(valid[b]).flatten().tolist()
            if ks:
                outs.append([logits[b, k].unsqueeze(0) for k in ks])
                labs.append([targets[b, k].unsqueeze(0) for k in ks])
        return {"cross_entropy": torch.zeros(())}, outs, labs


def loop_case(n_batches=5, B=4, T=31, seed=7):
    """Batches in the loader's format (before format_data) + the model's tables for each.  Includes a batch without
    boxes (skipped), clips without valid objects, a single-string `sequence`, and a batch without sample ids."""
    g = torch.Generator().manual_seed(seed)
    batches, tables = [], []
    for i in range(n_batches):
        n_ev = 40
        M = 0 if i == 2 else int(torch.randint(3, 9, (1,), generator=g))
        bbox = torch.rand((M, 6), generator=g) * 50
        if M:
            bbox[:, 4] = (torch.rand(M, generator=g) < 0.4).float()
        d = StubBatch(x=torch.ones((n_ev, 1), dtype=torch.int8), pos=torch.randint(0, 64, (n_ev, 2), generator=g).to(torch.int16),
                      t=torch.sort(torch.randint(0, 1000000, (n_ev,), generator=g).to(torch.int32)).values,
                      batch=torch.sort(torch.randint(0, B, (n_ev,), generator=g)).values,
                      image=torch.randint(0, 255, (B, 3, 8, 8), generator=g).to(torch.uint8),
                      bbox=bbox, bbox_batch=torch.sort(torch.randint(0, B, (M,), generator=g)).values,
                      width=torch.full((B,), 64), height=torch.full((B,), 64), time_window=torch.full((B,), 1000000),
                      num_graphs=B)
        d.sequence = "solo_video" if i == 3 else ["vid_%d" % ((i * B + b) // 3) for b in range(B)]
        if i != 4:
            d.sample_id = ["clip_frame_%d_x" % (10 * i + 2 * b) for b in range(B)]
        batches.append(d)
        if M:
            logits = torch.randn((B, T, 2), generator=g)
            valid = torch.rand((B, T), generator=g) < 0.12
            valid[:, 0] = False
            if i == 1:
                valid[1] = False                                  # a clip without objects shifts the j indexing
            targets = (torch.rand((B, T), generator=g) < 0.3).long()
            tables.append((logits, valid, targets))
    return batches, tables
