"""Validation-AUC parity (the bar: AUC within 1e-3 of the reference): >= 2000 labelled objects from the
seeded generator, raw class-1 logit as the score, unadjusted ROC AUC (reference utils/visualization.py:81-82
definition), native GPU path vs the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _auc(scores, labels):
    order = torch.argsort(scores)
    ranks = torch.empty_like(scores, dtype=torch.float64)
    ranks[order] = torch.arange(1, len(scores) + 1, dtype=torch.float64)
    s_sorted = scores[order]                                   # average ranks over ties
    i = 0
    while i < len(s_sorted):
        j = i
        while j + 1 < len(s_sorted) and s_sorted[j + 1] == s_sorted[i]:
            j += 1
        if j > i:
            ranks[order[i:j + 1]] = (i + 1 + j + 1) / 2.0
        i = j + 1
    pos = labels == 1
    n1, n0 = int(pos.sum()), int((~pos).sum())
    return float((ranks[pos].sum() - n1 * (n1 + 1) / 2.0) / (n1 * n0))


def test_auc_parity_over_2000_objects():
    import eventad_amd.compat as compat
    compat.install()
    from config.eventad_config import parse_eventad_args
    from dagr.model.networks.dagr import DAGR
    from dagr.model.networks.ema import ModelEMA
    from dagr.utils.buffers import format_data
    from models.EventAD import EventADModel
    from eventad_amd import synthetic, weights
    from oracle import dagr_ref, head_ref
    from tests.helpers import seeded_head_state
    W = H = 64; B = 8; n = 2500
    args = parse_eventad_args([])
    args.batch_size = B
    dev = torch.device("cuda:0")
    sd = weights.random_graph_state()
    dagr = DAGR(args, height=H, width=W).to(dev)
    ema = ModelEMA(dagr)
    ema.ema.load_state_dict(sd, strict=False)
    ema.ema.cache_luts(radius=args.radius, height=H, width=W)
    model = EventADModel(ema.ema).to(dev).eval()
    head_sd = seeded_head_state(4)
    model.load_state_dict(head_sd, strict=False)
    cfg = dagr_ref.make_config(W, H, B)
    luts = dagr_ref.cache_luts(sd, cfg)
    got, want, labs = [], [], []
    for it in range(26):                                        # 26 batches x 8 clips x 10 tracks = 2080 objects
        raw = synthetic.make_batch(W, H, n, B, first_clip=it * B, with_image=False, precomputed_feats=True)
        data = format_data(raw.clone().to(dev))
        with torch.no_grad():
            _, outputs, labels = model(data, data.bbox[:, 4])
        got.append(torch.cat([o for f in outputs for o in f], 0).cpu())
        labs.append(torch.cat([t for f in labels for t in f], 0).cpu())
        d = dagr_ref.format_data(raw.clone())
        _, o4 = dagr_ref.backbone_forward(sd, d, cfg, d.image_feats, mode="grouped", luts=luts)
        _, outs = head_ref.head_forward(head_sd, head_ref.roi_features(o4, d), d, d.bbox[:, 4])
        want.append(torch.cat([o[2] for o in outs], 0))
        assert torch.equal(labs[-1], torch.cat([o[3] for o in outs], 0))
    got, want, labs = torch.cat(got), torch.cat(want), torch.cat(labs)
    assert len(labs) >= 2000 and 0 < int(labs.sum()) < len(labs)
    assert float((got - want).abs().max()) < 1e-3
    assert torch.equal(got.argmax(1), want.argmax(1))
    a_native, a_oracle = _auc(got[:, 1].double(), labs), _auc(want[:, 1].double(), labs)
    assert abs(a_native - a_oracle) <= 1e-3, (a_native, a_oracle)
    print("AUC native %.6f oracle %.6f over %d objects, max |dlogit| %.2e" % (a_native, a_oracle, len(labs), float((got - want).abs().max())))
