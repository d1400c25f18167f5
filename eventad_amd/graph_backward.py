"""Backward of the event-level stage through the native kernels (csrc/bwd.hip) -- LUT spline convolution, pool1's voxel
max and the CNN-feature sampler: the gradients a fine-tuning of the DAGR backbone needs.  The reference gets it from autograd over MySplineConv's gather +
scatter-add (src/dagr/model/layers/spline_conv.py:39-78); EventAD itself freezes the backbone (models/EventAD.py:148-150),
so nothing here runs in its training step.

Host logic only: workspace, the launch sequence, and the (linear, tiny) map from the per-spiral-step gradient dLUT back to
dL/d weight[25, Cin, 16] and dL/d root, done by autograd over the same torch expression that builds the LUT."""
import torch

from . import spline
from ._lib import call, load, ptr


class EventConvBackward:
    """Gradients of  out_i = sum_{j->i} LUT[cell_ji]^T x_j + W_root x_i  on the ELL graph of a NativeBackbone workspace
    (pixel-sorted node numbering)."""

    def __init__(self, geom, device):
        self.geom, self.dev = geom, torch.device(device)
        self._n = None

    def _alloc(self, N):
        if self._n is not None and self._n >= N:
            return
        dev = self.dev
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
        self.out_start, self.out_edge = i32(N + 1), i32(16 * N)
        self.cnt, self.scan_tmp, self.tmp = i32(N + 1), i32(N // 2048 + 4), i32(16 * N)
        self.arrival = torch.empty(16 * N, dtype=torch.int16, device=dev)
        self.parts = int(load().eventad_l1_conv_bwd_weight_parts(N))
        self._n = N

    def transpose(self, ws, N):
        """Out-edge lists of the graph in `ws` (rebuild after every forward: the graph changes with the batch)."""
        self._alloc(N)
        call("eventad_ell_transpose", ptr(ws["ell_src"]), ptr(ws["deg"]), N, ptr(self.out_start), ptr(self.out_edge), ptr(self.cnt),
             ptr(self.scan_tmp), ptr(self.arrival), ptr(self.tmp), torch.cuda.current_stream().cuda_stream)

    def _full_lut(self, weight, root, cin_pad):
        lut = spline.lut_by_spiral_step(weight, self.geom, cin_pad)
        root_t = torch.zeros(cin_pad, weight.shape[2], dtype=weight.dtype)
        root_t[: root.shape[1]] = root.t()
        return torch.cat([lut, (lut[0] + root_t).unsqueeze(0)])                # row `steps` = matrix of the self edge

    def backward(self, ws, N, x, g, weight, root):
        """x f32[N, cin_pad] = the conv's input rows (sorted numbering, device), g f32[N,16] = dL/d(conv output),
        weight f32[25, Cin, 16] / root f32[16, Cin] (CPU or device).  Call transpose() first.
        -> (dx f32[N, cin_pad], dweight f32[25, Cin, 16], droot f32[16, Cin])."""
        cin_pad, steps = int(x.shape[1]), self.geom.steps
        st = torch.cuda.current_stream().cuda_stream
        w_cpu = weight.detach().float().cpu().requires_grad_(True)
        r_cpu = root.detach().float().cpu().requires_grad_(True)
        full = self._full_lut(w_cpu, r_cpu, cin_pad)                             # differentiable, ~50 x 320 floats
        lut_dev = spline.quad_row_order(full.detach()).to(self.dev)
        x, g = x.contiguous(), g.contiguous()
        dx = torch.empty((N, cin_pad), device=self.dev)
        call("eventad_l1_conv_bwd_data", ptr(g), ptr(self.out_start), ptr(self.out_edge), ptr(ws["ell_step"]), ptr(lut_dev), ptr(dx),
             N, steps, cin_pad, st)
        partial = torch.empty(self.parts * (steps + 1) * cin_pad * 16, device=self.dev)
        dlut = torch.empty((steps + 1, cin_pad, 16), device=self.dev)
        call("eventad_l1_conv_bwd_weight", ptr(x), ptr(g), ptr(ws["ell_src"]), ptr(ws["ell_step"]), ptr(ws["deg"]), ptr(partial),
             ptr(dlut), N, steps, cin_pad, st)
        dw, dr = torch.autograd.grad(full, (w_cpu, r_cpu), dlut.cpu())          # the LUT is linear in (weight, root)
        return dx, dw, dr


class EventPoolBackward:
    """Gradients of pool1 (voxel max over [layer-1 output | 64 sampled CNN channels], reference pooling.py:51-97 after
    net.py:200-228) and of the bilinear samplers, on the workspace a NativeBackbone.front() pass left behind."""

    def __init__(self, backbone):
        self.nb = backbone

    def pool1(self, N, B, dfeat, y2=None, fmap=None):
        """dfeat f32[S,80] = dL/d(slot feature table of level 0).  y2 / fmap default to the forward pass's own layer-1
        output and 64-channel NHWC map.  -> (dy2 f32[N,16], dsamp f32[N,64]) in pixel-sorted numbering."""
        nb, ws, g = self.nb, self.nb.ws, self.nb.geom
        L = ws["levels"][0]
        y2 = ws["y2"] if y2 is None else y2.contiguous()
        fmap = fmap.contiguous()
        assert tuple(dfeat.shape) == (L.S, 80) and fmap.shape[0] == B and fmap.shape[3] == 64
        dfeat = dfeat.contiguous().float()
        dy2 = torch.empty((N, 16), device=dfeat.device)
        dsamp = torch.empty((N, 64), device=dfeat.device)
        call("eventad_pool_seg0_bwd", ptr(y2), ptr(fmap), fmap.shape[1], fmap.shape[2], ptr(ws["posn_s"]), ptr(ws["key_s"]),
             ptr(ws["node_slot0"]), ptr(ws["pix_start"]), ptr(nb.vox_tabs[2]), ptr(nb.vox_tabs[3]), ptr(ws["ovf_list"]),
             ptr(ws["counters"]), N, B, g.W, g.H, L.nx, L.ny, ptr(dfeat), ptr(dy2), ptr(dsamp),
             torch.cuda.current_stream().cuda_stream)
        return dy2, dsamp

    def sampler(self, N, B, grad_rows, channels, map_h, map_w, column=0):
        """grad_rows f32[N, ld]: per-event gradient rows (pixel-sorted numbering) holding the sampled channels at
        [column, column + channels).  -> dL/d(feature map) f32[B, map_h, map_w, channels] (NHWC)."""
        ws, g = self.nb.ws, self.nb.geom
        grad_rows = grad_rows.contiguous().float()
        assert grad_rows.shape[0] >= N and grad_rows.shape[1] >= column + channels
        dmap = torch.empty((B, map_h, map_w, channels), device=grad_rows.device)
        call("eventad_sample_bwd", grad_rows.data_ptr() + 4 * column, grad_rows.shape[1], channels, ptr(ws["posn_s"]),
             ptr(ws["pix_start"]), N, B, g.W, g.H, map_h, map_w, ptr(dmap), torch.cuda.current_stream().cuda_stream)
        return dmap
