"""Host-side spline-kernel preparation for the native layers (product code).

The reference caches, per conv, a LUT of blended kernel matrices for every integer pixel offset
(MySplineConv.init_lut, src/dagr/model/layers/spline_conv.py:16-37) using
torch_spline_conv.spline_basis (degree-1 open B-spline, kernel_size 5).  For the event-level
layer the offsets are the (2r+1)^2 spiral steps of the graph builder, so the LUT is tiny and is
built here once, ordered by spiral step.  The pooled layers never build a LUT (see csrc/tail.hip)."""
import ctypes

import torch

from . import _lib


def basis_degree1(attr):
    """attr f32[n] in pseudo-coordinate space -> (i0, i1, b0, b1): taps ((int)v + k) % 5 with weights
    1 - frac - k + 2*frac*k, v = 4*attr  (torch_spline_conv basis, kernel_size 5, open spline)."""
    v = attr * 4.0
    t = v.long()
    fr = v - torch.floor(v)
    return t % 5, (t + 1) % 5, 1 - fr, 1 - fr - 1 + 2 * fr


def spiral_offsets(radius_px):
    steps = (2 * radius_px + 1) ** 2
    buf = (ctypes.c_int16 * (2 * steps))()
    n = _lib.load().eventad_spiral_offsets(radius_px, ctypes.cast(buf, ctypes.c_void_p))
    assert n == steps
    return torch.tensor(list(buf), dtype=torch.long).view(steps, 2)


def lut_by_spiral_step(weight, geom, cin_pad):
    """weight f32[25,Cin,Cout] -> f32[steps, cin_pad, Cout]: the matrix the reference's LUT holds for the
    pixel offset of each spiral step (offset = pos[src]-pos[dst], asynchronous/cartesian.py:9)."""
    div_x, div_y = geom.lut_div[0]
    d = spiral_offsets(geom.radius_px).float()
    ax = d[:, 0] / div_x + 0.5
    ay = d[:, 1] / div_y + 0.5
    ix0, ix1, bx0, bx1 = basis_degree1(ax)
    iy0, iy1, by0, by1 = basis_degree1(ay)
    w = weight
    lut = ((bx0 * by0)[:, None, None] * w[ix0 + 5 * iy0] + (bx1 * by0)[:, None, None] * w[ix1 + 5 * iy0]
           + (bx0 * by1)[:, None, None] * w[ix0 + 5 * iy1] + (bx1 * by1)[:, None, None] * w[ix1 + 5 * iy1])
    out = torch.zeros(lut.shape[0], cin_pad, lut.shape[2])
    out[:, : lut.shape[1]] = lut
    return out.contiguous()


def fold_bn(sd, prefix, eps=1e-5):
    """Eval-mode BatchNorm (components.py:9-12) as y = x*scale + shift."""
    scale = sd[prefix + "weight"] / torch.sqrt(sd[prefix + "running_var"] + eps)
    shift = sd[prefix + "bias"] - sd[prefix + "running_mean"] * scale
    return scale.contiguous(), shift.contiguous()


def pad_rows(w, rows):
    out = torch.zeros(rows, w.shape[1])
    out[: w.shape[0]] = w
    return out.contiguous()


def quad_row_order(w):
    """Re-order the input-channel rows of f32[..., Cin, 16] weight matrices (Cin = 16 or 20) for the event-level conv
    kernel, whose 4 lanes per destination split the input channels: row i*4 + j = i-th channel of lane j = channel 4j+i
    (i < 4) or 16+j (i = 4)."""
    cin = w.shape[-2]
    assert cin in (16, 20)
    order = [4 * j + i for i in range(4) for j in range(4)] + ([16 + j for j in range(4)] if cin == 20 else [])
    return w[..., order, :].contiguous()
