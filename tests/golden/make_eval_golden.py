"""Golden vectors for the test/validation loop, made by RUNNING THE REFERENCE in this container:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_eval_golden.py
imports /root/reference/utils/evaluation.py (numpy + sklearn) and /root/reference/utils/test.py (its
`dagr.utils.buffers` import is served by eventad_amd.compat) and stores their inputs and outputs in
tests/golden/eval_golden.json.  Only data is stored -- no reference source."""
import json, os, sys, io, contextlib
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import eventad_amd.compat
eventad_amd.compat.install()
sys.path.insert(0, "/root/reference")
import utils.evaluation as ref_eval                              # noqa: E402  (the reference's)
import utils.test as ref_test                                    # noqa: E402
from tests.eval_cases import metric_cases, loop_case, StubModel, clean  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


out = {"metrics": [], "loop": None}
for name, c in metric_cases().items():
    rec = {"name": name}
    rec["bbox"] = clean(quiet(ref_eval.calculate_bbox_metrics, np.array(c["labels"]), np.array(c["scores"])))
    rec["frame"] = clean(quiet(ref_eval.calculate_frame_metrics, c["frame_data"]))
    rec["tta"] = clean(quiet(ref_eval.calculate_tta_metrics, c["video_predictions"], c["video_first_anomaly"], c["video_toa"]))
    rec["response"] = clean(quiet(ref_eval.calculate_response_metrics, c["video_predictions"], fps=c["fps"]))
    out["metrics"].append(rec)

batches, tables = loop_case()
res = quiet(ref_test.collect_predictions, StubModel(tables, nested=True), batches, torch.device("cpu"), None, 0.5)
out["loop"] = clean(res)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "eval_golden.json"), "w"), indent=1, sort_keys=True)
print("wrote eval_golden.json:", len(out["metrics"]), "metric cases,", len(res["all_scores"]), "objects in the loop case")
