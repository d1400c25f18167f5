"""Shared, seeded test inputs (CPU torch generators: identical on every box)."""
import math

import torch

HEAD_SHAPES = {
    "gru_net_event.gru.weight_ih_l0": (768, 64), "gru_net_event.gru.weight_hh_l0": (768, 256),
    "gru_net_event.gru.bias_ih_l0": (768,), "gru_net_event.gru.bias_hh_l0": (768,),
    "gru_net_event.gru.weight_ih_l1": (768, 256), "gru_net_event.gru.weight_hh_l1": (768, 256),
    "gru_net_event.gru.bias_ih_l1": (768,), "gru_net_event.gru.bias_hh_l1": (768,),
    "gru_net_cor.gru.weight_ih_l0": (96, 4), "gru_net_cor.gru.weight_hh_l0": (96, 32),
    "gru_net_cor.gru.bias_ih_l0": (96,), "gru_net_cor.gru.bias_hh_l0": (96,),
    "fusion_module.event_proj.weight": (256, 256), "fusion_module.event_proj.bias": (256,),
    "fusion_module.coord_proj.weight": (256, 32), "fusion_module.coord_proj.bias": (256,),
    "fusion_module.fusion.0.weight": (256, 512), "fusion_module.fusion.0.bias": (256,),
    "fusion_module.fusion.3.weight": (2, 256), "fusion_module.fusion.3.bias": (2,),
    "soft_attention.weight": (256, 1), "soft_attention_cor.weight": (32, 1),
}


def seeded_head_state(seed):
    """Head parameters (reference state-dict names, models/EventAD.py) drawn from a CPU generator."""
    g = torch.Generator().manual_seed(1000 + seed)
    sd = {}
    for k in sorted(HEAD_SHAPES):
        shp = HEAD_SHAPES[k]
        if len(shp) == 1:
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) / math.sqrt(shp[1])
    return sd


def grad_sample_index(numel, seed, n=4096):
    g = torch.Generator().manual_seed(2000 + seed)
    return torch.randperm(numel, generator=g)[: min(n, numel)]


def make_head_case(seed, num_graphs, tracks, nodes_per_clip, drop_prob, empty_clip, far_box,
                   width=256, height=256):
    """Seeded out4 nodes + boxes in the path's input schema (eventad_amd/batch.py)."""
    g = torch.Generator().manual_seed(3000 + seed)
    xs, ps, bs = [], [], []
    for b in range(num_graphs):
        n = 0 if b == empty_clip else nodes_per_clip
        px = torch.randint(0, width, (n,), generator=g).float() / width
        py = torch.randint(0, height, (n,), generator=g).float() / height
        pt = torch.rand(n, generator=g)
        xs.append(torch.randn(n, 64, generator=g))
        ps.append(torch.stack([px, py, pt], 1))
        bs.append(torch.full((n,), b, dtype=torch.long))
    bbox, bbox_b, bbox0, bbox0_b = [], [], [], []
    for b in range(num_graphs):
        ids = torch.randperm(30, generator=g)[:tracks] + 1
        for j, tid in enumerate(sorted(ids.tolist())):
            w = float(torch.randint(16, width // 4 + 1, (1,), generator=g))
            h = float(torch.randint(16, height // 4 + 1, (1,), generator=g))
            x = float(torch.rand(1, generator=g)) * (width - w)
            y = float(torch.rand(1, generator=g)) * (height - h)
            if far_box and j == 0:
                w, h, x, y = 2.0, 2.0, 0.5, 0.5
            cls = float(torch.rand(1, generator=g) < 0.2)
            keep = torch.rand(2, generator=g) >= drop_prob
            if keep[0]:
                bbox.append([x, y, w, h, cls, float(tid)])
                bbox_b.append(b)
            if keep[1]:
                dx, dy = torch.randint(-2, 3, (2,), generator=g).tolist()
                bbox0.append([x + dx, y + dy, w, h, cls, float(tid)])
                bbox0_b.append(b)
    return dict(num_graphs=num_graphs,
                out4_x=torch.cat(xs), out4_pos=torch.cat(ps), out4_batch=torch.cat(bs),
                bbox=torch.tensor(bbox, dtype=torch.float32).view(-1, 6),
                bbox_batch=torch.tensor(bbox_b, dtype=torch.long),
                bbox0=torch.tensor(bbox0, dtype=torch.float32).view(-1, 6),
                bbox0_batch=torch.tensor(bbox0_b, dtype=torch.long),
                width=torch.full((num_graphs,), width, dtype=torch.long),
                height=torch.full((num_graphs,), height, dtype=torch.long))
