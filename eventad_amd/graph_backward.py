"""Backward of the event-level LUT spline convolution through the native kernels (csrc/bwd.hip) -- the message-passing
gradient a fine-tuning of the DAGR backbone needs.  The reference gets it from autograd over MySplineConv's gather +
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
