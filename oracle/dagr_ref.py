"""oracle/dagr_ref.py -- TEST INFRASTRUCTURE ONLY (CPU checker, never the product path).

Functional CPU restatement (torch fp32 on CPU) of the frozen DAGR backbone that
EventAD runs under no_grad.  Every function cites
the reference lines it follows (paths relative to /root/reference).  Third-party
arithmetic that is NOT in the reference tree (torch_geometric 2.6.1,
torch_spline_conv 1.2.2, torch_cluster 1.6.3, torch_scatter 2.1.2 -- pins from
requirements.txt:48-54) is restated from its published algorithm, anchored on
the reference's own plain-torch restatements where they exist:
  * Cartesian  = pos[row]-pos[col]            src/dagr/asynchronous/cartesian.py:6-16
  * propagate  = x[edge_index[0]] -> message -> sum at edge_index[1]
                                               src/dagr/asynchronous/conv.py:86-101
  * pool_edge / voxel index                    src/dagr/asynchronous/max_pool.py:13-26,245-252

PARITY STATUS: the reference has no tests/golden vectors for the backbone and its
dagr package cannot be imported here (torch_geometric & co are absent), so the
backbone part of the oracle is pinned only by the hand-derived known-answer
tests in tests/test_oracle_*.py ("parity unpinned" w.r.t. reference outputs).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import graph as G

KERNEL_SIZE = 5          # dagr-s-dsec.yaml kernel_size
BN_EPS = 1e-5            # torch_geometric BatchNorm default


# --------------------------------------------------------------------------- A1
def format_data(data):
    """src/dagr/utils/buffers.py:33-44 (normalizer=None branch).  In place."""
    normalizer = torch.stack([data.width[0], data.height[0], data.time_window[0]], dim=-1)
    if hasattr(data, "image") and data.image is not None:
        data.image = data.image.float() / 255.0
    data.pos = torch.cat([data.pos, data.t.view((-1, 1))], dim=-1)
    data.t = None
    data.x = data.x.float()
    data.pos = data.pos / normalizer
    return data


# ---------------------------------------------------------------- net.py:19-28
def compute_pooling_at_each_layer(pooling_dim_at_output, num_layers=4):
    py, px = map(int, pooling_dim_at_output.split("x"))
    base = torch.tensor([1.0 / px, 1.0 / py, 1.0 / 1])
    out = []
    for i in range(num_layers):
        p = base / 2 ** (3 - i)
        p[-1] = 1
        out.append(p)
    return torch.stack(out)


def make_config(width, height, batch_size, radius=0.01, max_neighbors=16,
                pooling_dim_at_output="5x7", time_window=1000000):
    """Derived constants of Net.__init__ (net.py:30-99) and DAGR.cache_luts (dagr.py:36-71)."""
    poolings = compute_pooling_at_each_layer(pooling_dim_at_output, 4)             # net.py:66
    max_vals = 2 * poolings[:, :2].max(-1).values                                   # net.py:67
    eff_r = 2 * float(int(radius * width + 2) / width)                              # net.py:70
    voxel = [torch.cat([p, torch.Tensor([1])]) for p in poolings]                   # pooling.py:24
    cart_max = [eff_r, 2 * eff_r, max_vals[1], max_vals[2], max_vals[3]]            # net.py:71,75,81,87,93
    # LUT parameters per Layer: (rx, ry, M)      dagr.py:36-63 + model/utils.py:112-116
    lut = [(int(radius * width + 1), int(radius * width + 1), eff_r)]
    for k in range(4):
        rx = int(math.ceil(float(2 * voxel[k][0].numpy() * width)))
        ry = int(math.ceil(float(2 * voxel[k][1].numpy() * height)))
        lut.append((rx, ry, cart_max[k + 1]))
    return SimpleNamespace(width=width, height=height, batch_size=batch_size, radius=radius,
                           max_neighbors=max_neighbors, time_window=time_window,
                           voxel=voxel, cart_max=cart_max, lut=lut, eff_r=eff_r,
                           pool_aggr=["max", "max", "max", "mean"])                # net.py:76-95


# ------------------------------------------------------------------------- A13
def cartesian(pos, edge_index, max_value, n_dims=3):
    """T.Cartesian(norm=True, cat=False, max_value) as restated in
    src/dagr/asynchronous/cartesian.py:6-16; empty case components.py:31-35."""
    if edge_index.shape[1] == 0:
        return torch.zeros((0, n_dims), dtype=pos.dtype)
    row, col = edge_index
    cart = pos[row] - pos[col]
    return cart / (2 * max_value) + 0.5


# ------------------------------------------------------------------------- A15
def spline_basis(pseudo, kernel_size=KERNEL_SIZE, is_open_spline=1, degree=1):
    """torch_spline_conv 1.2.2 `spline_basis` (csrc/cpu/basis_cpu.cpp), degree 1, restated:
    v = pseudo*(kernel_size - degree*is_open); index += ((int)v + k_mod) % kernel_size * 5^d;
    frac = v - floor(v); basis *= 1 - frac - k_mod + 2*frac*k_mod.  Returns (basis[E,S], index[E,S])."""
    E, D = pseudo.shape
    S = (degree + 1) ** D
    basis = torch.ones(E, S, dtype=pseudo.dtype)
    widx = torch.zeros(E, S, dtype=torch.long)
    for s in range(S):
        k, wi_offset = s, 1
        for d in range(D):
            k_mod = k % (degree + 1)
            k //= degree + 1
            v = pseudo[:, d] * (kernel_size - degree * is_open_spline)
            widx[:, s] += ((v.long() + k_mod) % kernel_size) * wi_offset
            wi_offset *= kernel_size
            v = v - torch.floor(v)
            basis[:, s] *= 1 - v - k_mod + 2 * v * k_mod
    return basis, widx


def init_lut(weight, height, width, rx, Mx, ry=None, My=None):
    """MySplineConv.init_lut, src/dagr/model/layers/spline_conv.py:16-37."""
    ry = ry or rx
    My = My or Mx
    remap = torch.Tensor([[2 * Mx * width, 0, -Mx * width + rx],
                          [0, 2 * My * height, -My * height + ry]])
    dxy = torch.stack(torch.meshgrid(torch.arange(-rx, rx + 1), torch.arange(-ry, ry + 1),
                                     indexing="ij")).float()
    dxy[0] = dxy[0] / (2 * Mx * width) + 0.5
    dxy[1] = dxy[1] / (2 * My * height) + 0.5
    edge_attr = dxy.view((2, -1)).t()
    bil_w, indices = spline_basis(edge_attr)
    _, cin, cout = weight.shape
    lut = torch.empty((edge_attr.shape[0], cin, cout), dtype=weight.dtype)
    for s0 in range(0, edge_attr.shape[0], 2048):       # chunked: the layer-5 LUT is 1 GB at 256x256
        e = slice(s0, s0 + 2048)
        lut[e] = (bil_w[e][..., None, None] * weight[indices[e]]).sum(1)
    return lut.view((2 * rx + 1, 2 * ry + 1, cin, cout)), remap


def lut_cells(weight, height, width, rx, Mx, ix, iy, ry=None, My=None):
    """The LUT entries init_lut would hold at integer cells (ix, iy) -- i64[n] each, 0 <= ix <= 2rx, 0 <= iy <= 2ry --
    evaluated on demand with the same fp32 expressions (spline_conv.py:27-35), so a geometry whose full LUT does not
    fit in memory (14 GB per conv at 1280x720, layer 5) can still be checked.  -> f32[n, Cin, Cout]; equal bit for
    bit to init_lut(...)[0][ix, iy] (tests/test_oracle_ops.py)."""
    ry = ry or rx
    My = My or Mx
    ax = (ix - rx).float() / (2 * Mx * width) + 0.5
    ay = (iy - ry).float() / (2 * My * height) + 0.5
    bil_w, indices = spline_basis(torch.stack([ax, ay], 1))
    return (bil_w[..., None, None] * weight[indices]).sum(1)


def lut_indices(edge_attr, remap):
    """spline_conv.py:41-42: integer LUT cell of every edge."""
    dx = (edge_attr[:, 0] * remap[0, 0] + remap[0, -1] + 1e-3).long()
    dy = (edge_attr[:, 1] * remap[1, 1] + remap[1, -1] + 1e-3).long()
    return dx, dy


def spline_conv(x, edge_index, edge_attr, weight, root_weight, lut_params, height, width,
                mode="gather", chunk=1 << 18, cached=None):
    """MySplineConv.forward/_forward with the cached LUT message
    (spline_conv.py:39-78), aggr='sum', root_weight=True, bias=None.

    mode="gather": literal restatement -- gather lut[dx,dy] per edge (chunked so the
                   [E,Cin,Cout] temporary of spline_conv.py:44 stays small), messages summed at
                   the destination in edge order.
    mode="grouped": same sum, edges grouped by LUT cell and multiplied as one matmul per cell
                   (fp32 summation order differs; used for the timed CPU baseline).
    mode="ondemand": as "gather", but only the LUT cells some edge uses are evaluated (lut_cells) instead of
                   building the whole table first: bit-identical to "gather", memory independent of the geometry."""
    rx, ry, M = lut_params
    n, cout = x.shape[0], weight.shape[2]
    out = torch.zeros((n, cout), dtype=x.dtype)
    E = edge_index.shape[1]
    if E > 0:
        if mode == "ondemand":
            lut = None
            remap = torch.Tensor([[2 * M * width, 0, -M * width + rx], [0, 2 * M * height, -M * height + ry]])
        else:
            lut, remap = cached if cached is not None else init_lut(weight, height, width, rx=rx, Mx=M, ry=ry)
        dx, dy = lut_indices(edge_attr[:, :2], remap)
        assert int(dx.min()) >= 0 and int(dx.max()) <= 2 * rx, "LUT x index out of range"
        assert int(dy.min()) >= 0 and int(dy.max()) <= 2 * ry, "LUT y index out of range"
        src, dst = edge_index
        if mode == "ondemand":
            cells, inv = torch.unique(dx * (2 * ry + 1) + dy, return_inverse=True)
            for s in range(0, E, chunk):
                e = slice(s, min(E, s + chunk))
                used, loc = torch.unique(inv[e], return_inverse=True)         # LUT entries this chunk needs
                cu = cells[used]
                w = lut_cells(weight, height, width, rx, M, cu // (2 * ry + 1), cu % (2 * ry + 1), ry=ry)[loc]
                msg = torch.einsum("nio,ni->no", w, x[src[e]])
                out.index_add_(0, dst[e], msg)
        elif mode == "gather":
            for s in range(0, E, chunk):
                e = slice(s, min(E, s + chunk))
                w = lut[dx[e], dy[e]]                               # [e, Cin, Cout]
                msg = torch.einsum("nio,ni->no", w, x[src[e]])
                out.index_add_(0, dst[e], msg)
        else:
            cell = dx * (2 * ry + 1) + dy
            order = torch.argsort(cell, stable=True)
            cells, counts = torch.unique_consecutive(cell[order], return_counts=True)
            lut_flat = lut.view(-1, lut.shape[2], lut.shape[3])
            start = 0
            for c, cnt in zip(cells.tolist(), counts.tolist()):
                e = order[start:start + cnt]
                out.index_add_(0, dst[e], x[src[e]] @ lut_flat[c])
                start += cnt
    out = out + x @ root_weight.t()                                 # spline_conv.py:72-73
    return out


# ------------------------------------------------------------------------- A16
def batch_norm_eval(x, sd, prefix):
    """BatchNormData (components.py:9-12) in eval mode: nn.BatchNorm1d running stats."""
    return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"],
                        sd[prefix + "weight"], sd[prefix + "bias"], False, 0.1, BN_EPS)


# ------------------------------------------------------------------------- A14
def cache_luts(sd, cfg, prefix="backbone."):
    """DAGR.cache_luts (dagr.py:36-63): the ten backbone LUTs, built once at model set-up."""
    luts = {}
    for li, name in enumerate(["conv_block1.", "layer2.", "layer3.", "layer4.", "layer5."]):
        rx, ry, M = cfg.lut[li]
        for blk in ("conv_block1.", "conv_block2."):
            k = prefix + name + blk + "conv.weight"
            luts[k] = init_lut(sd[k], cfg.height, cfg.width, rx=rx, Mx=M, ry=ry)
    return luts


def layer(sd, prefix, x, edge_index, edge_attr, lut_params, height, width, mode, luts=None):
    """Layer = ConvBlock + ConvBlockWithSkip, src/dagr/model/layers/conv.py:10-72."""
    p1, p2 = prefix + "conv_block1.", prefix + "conv_block2."
    x_skip = x.clone()                                              # shallow_copy, utils.py:158-166
    c1 = luts.get(p1 + "conv.weight") if luts else None
    c2 = luts.get(p2 + "conv.weight") if luts else None
    y = spline_conv(x, edge_index, edge_attr, sd[p1 + "conv.weight"], sd[p1 + "conv.lin.weight"],
                    lut_params, height, width, mode, cached=c1)
    y = F.relu(batch_norm_eval(y, sd, p1 + "norm.module."))         # conv.py:23-28
    z = spline_conv(y, edge_index, edge_attr, sd[p2 + "conv.weight"], sd[p2 + "conv.lin.weight"],
                    lut_params, height, width, mode, cached=c2)
    s = x_skip @ sd[p2 + "lin.mlp.weight"].t()                      # conv.py:50
    s = batch_norm_eval(s, sd, p2 + "norm_skip.module.")
    z = batch_norm_eval(z, sd, p2 + "norm.module.")
    return F.relu(z + s)                                            # conv.py:54


# -------------------------------------------------------------------------- A9
def sample_features(pos, batch, image_feat, width, height):
    """sample_features/_sample_features, src/dagr/model/networks/net.py:200-228."""
    x = pos[:, 0] * width
    y = pos[:, 1] * height
    b = batch.float()
    batch_size = image_feat.shape[0]
    x = 2 * x / (width - 1) - 1
    y = 2 * y / (height - 1) - 1
    batch_size = batch_size if batch_size > 1 else 2
    b = 2 * b / (batch_size - 1) - 1
    grid = torch.stack((x, y, b), dim=-1).view(1, 1, 1, -1, 3)
    vol = image_feat.permute(1, 0, 2, 3).unsqueeze(0)
    out = F.grid_sample(vol, grid=grid, mode="bilinear", align_corners=True)
    return out.view(vol.shape[1], -1).t()


# ------------------------------------------------------------------------- A17
def grid_cluster(pos4, size, start, end):
    """torch_cluster 1.6.3 grid_cluster (csrc/cpu/grid_cpu.cpp) restated: per dim
    c_d = (int64)((pos_d - start_d)/size_d); id = sum c_d * prod_{e<d} ((int64)((end_e-start_e)/size_e)+1)."""
    pos4 = pos4 - start.unsqueeze(0)
    num_voxels = ((end - start) / size).to(torch.long) + 1
    num_voxels = num_voxels.cumprod(0)
    num_voxels = torch.cat([torch.ones(1, dtype=torch.long), num_voxels], 0)[: size.numel()]
    out = (pos4 / size.view(1, -1)).to(torch.long)
    out = out * num_voxels.view(1, -1)
    return out.sum(1)


def consecutive_cluster(src):
    """pooling.py:12-16.  On CPU `scatter_` with duplicate indices is applied in order,
    so perm[c] = the LARGEST node index in cluster c."""
    unique, inv = torch.unique(src, sorted=True, return_inverse=True)
    perm = torch.zeros(unique.size(0), dtype=torch.long)
    perm.scatter_reduce_(0, inv, torch.arange(inv.size(0)), reduce="amax", include_self=False)
    return unique, inv, perm


def round_to_pixel(pos_xy, wh_inv):
    """pooling.py:47-49."""
    q = torch.div(pos_xy + 1e-5, wh_inv, rounding_mode="floor")
    return q * wh_inv


def scatter_mean(src, index, n):
    """torch_scatter scatter(reduce='mean') as used by PyG pool_pos/_avg_pool_x: sum / max(count,1)."""
    out = torch.zeros((n,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(len(index), dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return out / cnt.view(-1, *([1] * (src.dim() - 1)))


def pooling(x, pos, batch, edge_index, voxel_size, cart_max, aggr, width, height, batch_size):
    """Pooling.forward, src/dagr/model/layers/pooling.py:51-97 (self_loop=False,
    keep_temporal_ordering=False, bn=None)."""
    if x.shape[0] == 0:
        return x, pos, batch, edge_index, torch.zeros((0, 3))
    start = torch.Tensor([0, 0, 0, 0])
    end = torch.Tensor([0.9999999, 0.9999999, 0.9999999, batch_size - 1])
    pos4 = torch.cat([pos, batch.float().view(-1, 1)], dim=-1)
    cluster = grid_cluster(pos4, voxel_size, start, end)
    unique, cluster, perm = consecutive_cluster(cluster)
    n_out = unique.numel()
    ei = cluster[edge_index]
    ei = ei[:, ei[0] != ei[1]]
    if ei.shape[1] > 0:
        ei = ei.unique(dim=-1)
    new_batch = batch[perm]
    new_pos = scatter_mean(pos, cluster, n_out)                       # pool_pos
    if aggr == "max":
        new_x = torch.full((n_out, x.shape[1]), float("-inf"), dtype=x.dtype)
        new_x.scatter_reduce_(0, cluster.view(-1, 1).expand_as(x), x, reduce="amax", include_self=True)
    else:
        new_x = scatter_mean(x, cluster, n_out)                       # _avg_pool_x
    wh_inv = 1 / torch.Tensor([[width, height]])
    new_pos[:, :2] = round_to_pixel(new_pos[:, :2], wh_inv)
    edge_attr = cartesian(new_pos, ei, cart_max)
    return new_x, new_pos, new_batch, ei, edge_attr


# ------------------------------------------------------------------- A7 (+A10)
def backbone_forward(sd, data, cfg, image_feats, mode="gather", prefix="backbone.", return_all=False, luts=None):
    """Net.forward, src/dagr/model/networks/net.py:106-197 with use_image=True, num_scales=2.
    `data` is a formatted batch (A1).  `image_feats` = the five HookModule feature maps
    (A8; computed by oracle/resnet_ref.py or supplied as 'precomputed CNN features')."""
    W, H = cfg.width, cfg.height
    pos, batch = data.pos, data.batch
    B = int(data.num_graphs)
    stages = {}
    edge_index = G.build_edges(pos, batch, W, H, cfg.time_window, B, cfg.radius, cfg.max_neighbors)  # net.py:113
    x = torch.cat((data.x, sample_features(pos, batch, image_feats[0], W, H)), dim=1)              # net.py:116
    edge_attr = torch.clamp(cartesian(pos, edge_index, cfg.cart_max[0]), min=0, max=1)             # net.py:120-121
    x = torch.cat((x, pos[:, :2]), dim=1)                                                          # net.py:122-123
    stages["edge_index0"] = edge_index
    names = ["conv_block1.", "layer2.", "layer3.", "layer4.", "layer5."]
    out3 = None
    for li in range(5):
        x = layer(sd, prefix + names[li], x, edge_index, edge_attr, cfg.lut[li], H, W, mode, luts)
        stages["layer%d" % (li + 1)] = SimpleNamespace(x=x, pos=pos, batch=batch, edge_index=edge_index)
        if li == 3:
            out3 = SimpleNamespace(x=x.clone(), pos=pos, batch=batch, edge_index=edge_index, edge_attr=edge_attr)
        if li == 4:
            break
        x = torch.cat((x, sample_features(pos, batch, image_feats[li + 1], W, H)), dim=1)          # sampling_skip
        x, pos, batch, edge_index, edge_attr = pooling(x, pos, batch, edge_index, cfg.voxel[li],
                                                       cfg.cart_max[li + 1], cfg.pool_aggr[li], W, H,
                                                       cfg.batch_size)
        stages["pool%d" % (li + 1)] = SimpleNamespace(x=x, pos=pos, batch=batch, edge_index=edge_index,
                                                      edge_attr=edge_attr)
        x = torch.cat((x, pos[:, :2]), dim=1)
    out4 = SimpleNamespace(x=x, pos=pos, batch=batch, edge_index=edge_index, edge_attr=edge_attr)
    if return_all:
        return [out3, out4], stages
    return [out3, out4]
