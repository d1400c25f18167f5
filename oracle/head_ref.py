"""oracle/head_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the EventAD head in eval mode
(dropout inactive), functional over a dict of parameter tensors whose keys are
the reference's state-dict names.  Follows /root/reference/models/EventAD.py:

  roi_features        EventADModel.extract_features :335-403 + _process_bbox :406-499
  gru_step            nn.GRU single time step, gate order (r,z,n)   (:62-97)
  fuse                Fuse_GRUNet.forward :30-59
  spatial_attention   SpatialAttention.forward :111-138
  head_forward        EventADModel.forward :178-333

PARITY STATUS: pinned.  tests/golden/head_*.pt were produced by importing the
reference's models/EventAD.py in the build container (tests/golden/make_head_golden.py);
tests/test_oracle_head.py checks this restatement against them (features,
every logit, loss, every parameter gradient).
"""
import torch
import torch.nn.functional as F

MAX_BOXES = 30


def _item(v):
    return v.item() if isinstance(v, torch.Tensor) else v


def process_bbox(bbox, b, node_features, node_positions, node_batch, global_feature, width, height):
    """EventAD.py:406-499 -> (track_id or None, 64-vector)."""
    x, y, w, h = bbox[0].item(), bbox[1].item(), bbox[2].item(), bbox[3].item()
    track_id = int(bbox[5].item()) if bbox.shape[0] > 5 else 1
    if track_id <= 0 or track_id > MAX_BOXES:
        return None, None
    width_val, height_val = _item(width), _item(height)
    x1n, y1n = x / width_val, y / height_val
    x2n, y2n = (x + w) / width_val, (y + h) / height_val
    cxn, cyn = (x1n + x2n) / 2, (y1n + y2n) / 2
    idx_b = torch.where(node_batch == b)[0]
    if len(idx_b) == 0:
        return track_id, global_feature
    p = node_positions[idx_b]
    in_box = (p[:, 0] >= x1n) & (p[:, 0] <= x2n) & (p[:, 1] >= y1n) & (p[:, 1] <= y2n)
    idx_in = idx_b[torch.where(in_box)[0]]
    center = torch.tensor([cxn, cyn])
    if len(idx_in) > 0:
        d = torch.norm(node_positions[idx_in] - center.unsqueeze(0), dim=1)
        wgt = 1.0 / (d + 1e-6)
        wgt = wgt / wgt.sum()
        return track_id, (node_features[idx_in] * wgt.unsqueeze(1)).sum(dim=0)
    d = torch.norm(p - center.unsqueeze(0), dim=1)
    k = min(5, len(d))
    closest = idx_b[torch.topk(d, k, largest=False)[1]]
    return track_id, node_features[closest].mean(dim=0)


def roi_features(out4, data):
    """EventAD.py:335-403 without the feature cache (its key is weak: shapes and sums only) -> f32[B,2,31,64]."""
    B = int(data.num_graphs)
    nf, npos, nb = out4.x, out4.pos[:, :2], out4.batch
    glob = torch.zeros((B, nf.size(1)))
    for b in range(B):
        m = nb == b
        if m.any():
            glob[b] = nf[m].mean(dim=0)
    feats = torch.zeros((B, 2, MAX_BOXES + 1, nf.shape[1]))
    for b in range(B):
        for frame, (boxes, bb) in enumerate(((data.bbox0, data.bbox0_batch), (data.bbox, data.bbox_batch))):
            for idx in torch.where(bb == b)[0]:
                tid, f = process_bbox(boxes[idx], b, nf, npos, nb, glob[b], data.width[b], data.height[b])
                if tid is not None:
                    feats[b, frame, tid] = f
    return feats


def gru_step(x, h, w_ih, w_hh, b_ih, b_hh):
    """One nn.GRU time step for one layer; x [n,I], h [n,Hd]; PyTorch gate order r,z,n."""
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def event_gru(p, x, h):
    """Event_GRUNet (2 layers, eval) on one step: x [1,64], h [2,1,256] -> out [1,256], h' [2,1,256]."""
    g = "gru_net_event.gru."
    h0 = gru_step(x, h[0], p[g + "weight_ih_l0"], p[g + "weight_hh_l0"], p[g + "bias_ih_l0"], p[g + "bias_hh_l0"])
    h1 = gru_step(h0, h[1], p[g + "weight_ih_l1"], p[g + "weight_hh_l1"], p[g + "bias_ih_l1"], p[g + "bias_hh_l1"])
    return h1, torch.stack([h0, h1])


def coord_gru(p, x, h):
    g = "gru_net_cor.gru."
    h0 = gru_step(x, h[0], p[g + "weight_ih_l0"], p[g + "weight_hh_l0"], p[g + "bias_ih_l0"], p[g + "bias_hh_l0"])
    return h0, torch.stack([h0])


def fuse(p, ev, co):
    """Fuse_GRUNet.forward (EventAD.py:30-59), eval mode."""
    f = "fusion_module."
    e = ev @ p[f + "event_proj.weight"].t() + p[f + "event_proj.bias"]
    c = co @ p[f + "coord_proj.weight"].t() + p[f + "coord_proj.bias"]
    z = torch.cat([e, c], dim=1)
    z = F.relu(z @ p[f + "fusion.0.weight"].t() + p[f + "fusion.0.bias"])
    return z @ p[f + "fusion.3.weight"].t() + p[f + "fusion.3.bias"]


def spatial_attention(weight, h_all):
    """EventAD.py:111-138: h_all dict track-> [L,1,H]; softmax over the tracks of this clip."""
    keys = list(h_all.keys())
    if not keys:
        return h_all
    h_in = torch.cat([h_all[k] for k in keys], dim=1)              # [L,T,H]
    alpha = torch.softmax(torch.matmul(torch.tanh(h_in), weight), 1)
    roh = h_in * alpha
    return {k: roh[:, i, :].unsqueeze(1).contiguous() for i, k in enumerate(keys)}


def head_forward(p, feats, data, labels):
    """EventADModel.forward (EventAD.py:178-333).  Returns loss (0-d), list of (b, track, logits[1,2], target[1])."""
    B = int(data.num_graphs)
    h_in_ev, h_in_co = {}, {}
    loss = 0
    outs = []
    for b in range(B):
        cur = feats[b, 1]
        h_out_ev, h_out_co = {}, {}
        for tid in range(1, MAX_BOXES + 1):
            f = cur[tid]
            if torch.sum(f) == 0:
                continue
            rows = torch.where((data.bbox_batch == b) & (data.bbox[:, 5] == tid))[0]
            if len(rows) == 0:
                continue
            row = rows[0]
            bb = data.bbox[row]
            wv, hv = _item(data.width[b]), _item(data.height[b])
            x, y, w, h = bb[0].item(), bb[1].item(), bb[2].item(), bb[3].item()
            coord = torch.tensor([[x / wv, y / hv, w / wv, h / hv]])
            target = labels[row].long().unsqueeze(0) if labels is not None else torch.tensor([0])
            k = str(tid)
            he = h_in_ev[k] if k in h_in_ev else torch.zeros(2, 1, 256)
            hc = h_in_co[k] if k in h_in_co else torch.zeros(1, 1, 32)
            ev_out, he2 = event_gru(p, f.unsqueeze(0), he)
            co_out, hc2 = coord_gru(p, coord, hc)
            logits = fuse(p, ev_out, co_out)
            loss = loss + F.cross_entropy(logits, target)
            outs.append((b, tid, logits, target))
            h_out_ev[k], h_out_co[k] = he2, hc2
        if h_out_ev:
            h_in_ev.update(spatial_attention(p["soft_attention.weight"], h_out_ev))
        if h_out_co:
            h_in_co.update(spatial_attention(p["soft_attention_cor.weight"], h_out_co))
    if isinstance(loss, int):
        loss = torch.tensor(0.0, requires_grad=True)
    return loss, outs
