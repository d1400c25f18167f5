"""Test/validation loop: eventad_amd/utils/{evaluation,test}.py against golden vectors produced by the
reference's own utils/evaluation.py and utils/test.py (tests/golden/make_eval_golden.py)."""
import io, json, os, contextlib
import numpy as np
import pytest
import torch

import eventad_amd.compat
from tests.eval_cases import metric_cases, loop_case, StubModel, clean

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eval_golden.json")))


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _close(a, b, tol=1e-12):
    if isinstance(a, dict):
        assert set(a) == set(b), (set(a), set(b))
        for k in a:
            _close(a[k], b[k], tol)
    elif isinstance(a, list):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _close(x, y, tol)
    elif isinstance(a, float) and isinstance(b, (float, int)):
        assert abs(a - b) <= tol * max(1.0, abs(a)), (a, b)
    else:
        assert a == b, (a, b)


@pytest.mark.parametrize("rec", GOLD["metrics"], ids=[r["name"] for r in GOLD["metrics"]])
def test_metric_functions_match_the_reference(rec):
    from eventad_amd.utils import evaluation as E
    c = metric_cases()[rec["name"]]
    _close(clean(_quiet(E.calculate_bbox_metrics, np.array(c["labels"]), np.array(c["scores"]))), rec["bbox"])
    _close(clean(_quiet(E.calculate_frame_metrics, c["frame_data"])), rec["frame"])
    _close(clean(_quiet(E.calculate_tta_metrics, c["video_predictions"], c["video_first_anomaly"], c["video_toa"])), rec["tta"])
    _close(clean(_quiet(E.calculate_response_metrics, c["video_predictions"], fps=c["fps"])), rec["response"])


def test_plain_metrics_without_the_score_rewrite():
    from eventad_amd.utils import evaluation as E
    y = np.array([0, 0, 1, 1, 0, 1, 0, 0]); s = np.array([.1, .4, .35, .8, .2, .9, .5, .05])
    m = _quiet(E.calculate_bbox_metrics, y, s, reference_quirks=False)
    assert abs(m["auc"] - 13 / 15) < 1e-12                                      # 13 of the 15 (pos, neg) pairs are ordered correctly
    assert abs(m["ap"] - (1 + 1 + 3 / 5) / 3) < 1e-12                           # precision at the three positives: 1, 1, 3/5
    assert np.isnan(E.roc_auc(np.zeros(4), np.arange(4.0)))                     # one class only


@pytest.mark.parametrize("protocol", ["predict_tensors", "nested_lists"])
def test_collect_predictions_matches_the_reference(protocol):
    eventad_amd.compat.install(eval_loop=True)
    from utils.test import collect_predictions                                  # the alias a maintainer's test.py imports
    batches, tables = loop_case()
    res = _quiet(collect_predictions, StubModel(tables, nested=(protocol == "nested_lists")), batches, torch.device("cpu"), None, 0.5)
    _close(clean(res), GOLD["loop"], tol=0.0)
    assert res["all_preds"].dtype == np.int64 and res["all_scores"].dtype == np.float64


def test_parse_sample_id():
    from eventad_amd.utils.utils import parse_sample_id
    assert parse_sample_id("seq_frame_12_obj_3") == (None, 12, 3)
    assert parse_sample_id("clip_frame_7_x_2") == (None, 7, 0)
    assert parse_sample_id("nothing") == (None, 0, 0) and parse_sample_id(17) == (None, 0, 0)


@pytest.mark.gpu
def test_collect_predictions_on_the_hip_model_equals_its_nested_list_output():
    """Real model on the GPU: the one-transfer-per-batch path must give exactly what the reference protocol gives."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from tests.test_gpu_model import _build
    eventad_amd.compat.install(eval_loop=True)
    from utils.test import collect_predictions
    from eventad_amd.batch import EventBatch
    args, model, sd, head_sd, raw, data = _build(64, 64, 3, 3000)
    model.eval()

    class Loader(list):
        pass
    mk = lambda: Loader([raw.clone()])
    fast = _quiet(collect_predictions, model, mk(), torch.device("cuda"), None, 0.5)

    class Nested:                                                   # hides predict_tensors: reference protocol
        def __init__(self, m): self.m = m
        def __call__(self, *a, **k): return self.m(*a, **k)
    slow = _quiet(collect_predictions, Nested(model), mk(), torch.device("cuda"), None, 0.5)
    assert len(fast["all_scores"]) > 0
    _close(clean(fast), clean(slow), tol=0.0)


def test_measure_fps_counts_boxes_of_both_frames_and_skips_empty_batches():
    """measure_fps keeps the reference's bookkeeping (utils/fps.py:37-86): the loader index counts warm-up batches, batches
    without boxes are skipped, the unit is len(bbox) + len(bbox0), the model is called as model(data, labels, testing=True)
    in evaluation mode."""
    from types import SimpleNamespace
    import torch
    from eventad_amd.utils.fps import measure_fps

    class Batch(SimpleNamespace):
        def to(self, device):
            return self

    def batch(m, m0):
        return Batch(bbox=torch.zeros(m, 6), bbox0=torch.zeros(m0, 6), num_graphs=2, pos=torch.zeros(4, 2, dtype=torch.int16),
                     t=torch.zeros(4, dtype=torch.int32), x=torch.zeros(4, 1, dtype=torch.int8), batch=torch.zeros(4, dtype=torch.long),
                     width=torch.tensor([8, 8]), height=torch.tensor([8, 8]), time_window=torch.tensor([1000, 1000]), image=None)

    calls = []

    class Model(torch.nn.Module):
        def forward(self, data, labels, testing=False):
            calls.append((int(data.bbox.shape[0]), int(labels.shape[0]), testing, self.training))
            return None

    loader = [batch(3, 2), batch(0, 0), batch(4, 4), batch(5, 1), batch(0, 3), batch(2, 2), batch(7, 7)]
    res = measure_fps(Model().train(), loader, torch.device("cpu"), warmup_batches=2, num_batches=3)
    # index 0 is warm-up, index 1 is skipped (still inside the warm-up range), 2, 3 and 5 are timed, 4 has no boxes, 6 is beyond num_batches
    assert [c[0] for c in calls] == [3, 4, 5, 2]
    assert all(c[2] is True and c[3] is False and c[0] == c[1] for c in calls)
    assert res["batches"] == 3 and res["avg_batch_size"] == (8 + 6 + 4) / 3
    assert res["total_time"] > 0 and abs(res["fps"] - 18 / res["total_time"]) < 1e-9
    assert set(res) >= {"fps", "avg_batch_size", "total_time"}
