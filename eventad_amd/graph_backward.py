"""Backward of the DAGR backbone through the native kernels: the event-level stage (csrc/bwd.hip -- LUT spline convolution,
pool1's voxel max, the CNN-feature sampler) and the pooled levels (csrc/bwd_pooled.hip -- Layers 2..5, pools 2..4, samplers at
pooled nodes): the gradients a fine-tuning of the backbone needs.  The reference gets it from autograd over MySplineConv's gather +
scatter-add (src/dagr/model/layers/spline_conv.py:39-78); EventAD itself freezes the backbone (models/EventAD.py:148-150),
so nothing here runs in its training step.

Host logic only: workspace, the launch sequence, and the (linear, tiny) map from the per-spiral-step gradient dLUT back to
dL/d weight[25, Cin, 16] and dL/d root: the transpose of the blend that builds the LUT, as one small matrix product."""
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
        cin = int(weight.shape[1])
        full = self._full_lut(weight.detach().float().cpu(), root.detach().float().cpu(), cin_pad)
        lut_dev = spline.quad_row_order(full).to(self.dev)
        x, g = x.contiguous(), g.contiguous()
        dx = torch.empty((N, cin_pad), device=self.dev)
        call("eventad_l1_conv_bwd_data", ptr(g), ptr(self.out_start), ptr(self.out_edge), ptr(ws["ell_step"]), ptr(lut_dev), ptr(dx),
             N, steps, cin_pad, st)
        partial = torch.empty(self.parts * (steps + 1) * cin_pad * 16, device=self.dev)
        dlut = torch.empty((steps + 1, cin_pad, 16), device=self.dev)
        call("eventad_l1_conv_bwd_weight", ptr(x), ptr(g), ptr(ws["ell_src"]), ptr(ws["ell_step"]), ptr(ws["deg"]), ptr(partial),
             ptr(dlut), N, steps, cin_pad, st)
        # The LUT is linear in (weight, root): LUT[s] = sum_c blend[s, c] W_c, row `steps` = LUT[0] + root.  Its transpose as
        # one float64 product (autograd's index-accumulate over the 4 taps of 225 steps was not run-to-run reproducible).
        if getattr(self, "_blend", None) is None:
            self._blend = spline.lut_by_spiral_step(torch.eye(25).view(25, 25, 1), self.geom, 25)[:, :, 0].double()   # [steps, 25]
        dl = dlut.cpu().double()
        d = dl[:steps].clone()
        d[0] += dl[steps]
        dw = (self._blend.t() @ d.reshape(steps, -1)).reshape(25, cin_pad, 16)[:, :cin].float().contiguous()
        dr = dl[steps][:cin].t().float().contiguous()
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


def _wgrad_group(problems, device, stream):
    """C = A^T . B for up to 8 (A f32[K, >=M] lda, B f32[K, >=N] ldb, C f32[M, N], M, N, K) in ONE launch of the tiled grouped
    GEMM: K (= the node or event count) is cut into 256-deep slabs that run on different workgroups, and the slabs are
    summed in slab order by a second small launch (eventad_gemm_group) -- the plain GEMM would walk K serially on the few
    workgroups an [M, N] weight matrix gives it."""
    import ctypes
    desc = (ctypes.c_int64 * (16 * len(problems)))()
    keep = []
    for i, (A, lda, Bm, ldb, C, M, N, K) in enumerate(problems):
        part = torch.empty(((K + 255) // 256) * M * N, dtype=torch.float32, device=device) if K > 256 else None
        keep.append(part)
        desc[16 * i: 16 * i + 16] = [A.data_ptr(), Bm.data_ptr(), part.data_ptr() if part is not None else 0, C.data_ptr(), 0,
                                     lda, ldb, N, M, N, K, 1, 0, 0, 0, 0]
    call("eventad_gemm_group", len(problems), ctypes.cast(desc, ctypes.c_void_p), stream)


class PooledBackward:
    """Backward of NativeBackbone.tail(): Layers 2..5 on the pooled graphs (conv.py:59-72 over MySplineConv,
    spline_conv.py:39-78), pools 2..4 (pooling.py:51-97) and the CNN-feature samplers at pooled nodes (net.py:200-228),
    on the workspace the forward pass left behind.  The reference obtains these gradients from autograd.

    Host logic only (csrc/bwd_pooled.hip + the GEMM of csrc/gemm.hip do the work): the node and edge counts of the levels
    are read back once (a synchronisation -- this is a fine-tuning path, not part of EventAD's captured training step).
    Gradients are not taken with respect to node positions (they do not depend on any parameter) nor with respect to the
    BatchNorm statistics (eval mode)."""

    def __init__(self, backbone):
        load()
        self.nb = backbone

    def _levels(self):
        view = getattr(self.nb, "_tail_view", None)
        if view is None:
            raise RuntimeError("PooledBackward: run NativeBackbone.tail() (or forward()) first")
        return view

    def counts(self):
        """-> ([nodes per level], [edges per level]) of the last forward pass."""
        view = self._levels()
        nd = torch.cat([L.n_dev[:1] for L in view] + [L.e_dev[:1] for L in view]).cpu().tolist()
        return nd[:4], nd[4:]

    def transpose(self, lv, n, E):
        """Out-edge lists of level lv's graph -> (out_start, out_edge, edge_dst)."""
        L, dev = self._levels()[lv], self.nb.device
        i32 = lambda k: torch.empty(max(k, 1), dtype=torch.int32, device=dev)
        out_start, out_edge, edge_dst = i32(n + 1), i32(E), i32(E)
        cnt, scan_tmp, tmp = i32(n + 1), i32(n // 2048 + 4), i32(E)     # named: each stays allocated until the call is issued
        call("eventad_csr_transpose", ptr(L.rowptr), ptr(L.csr_src), n, E, ptr(out_start), ptr(out_edge), ptr(edge_dst),
             ptr(cnt), ptr(scan_tmp), ptr(tmp), torch.cuda.current_stream().cuda_stream)
        return out_start, out_edge, edge_dst

    def layer(self, lv, n, E, g_y):
        """Backward of Layer lv+2 on level lv.  g_y f32[n,64] = dL/d(layer output).
        -> (dX f32[n, cin_pad] = dL/d(layer input rows [features | pos_x | pos_y | 0..]), {state_dict key: gradient})."""
        nb, st = self.nb, torch.cuda.current_stream().cuda_stream
        L, t, dev = self._levels()[lv], nb.tail_w[lv], nb.device
        dvx, dvy = nb.geom.lut_div[lv + 1]
        cin, cin_pad = t["cin"], t["cin_pad"]
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        g_y = g_y.contiguous().float()
        assert tuple(g_y.shape) == (n, 64) and L.ldx == cin_pad
        out_start, out_edge, edge_dst = self.transpose(lv, n, E)
        # block 2: y = ReLU(bn2(blend(Z2) + Z2[root]) + bns(Z1[skip]))
        d2, ds = f32(max(n, 1), 64), f32(max(n, 1), 64)
        call("eventad_relu_scale_bwd", ptr(g_y), 64, ptr(L.y), 64, ptr(t["bn2"][0]), ptr(t["bns"][0]), n, 64, ptr(d2), ptr(ds), st)
        dZ2 = f32(max(n, 1), 26 * 64)
        call("eventad_tail_blend_bwd", ptr(d2), 64, ptr(out_start), ptr(out_edge), ptr(edge_dst), ptr(L.csr_dxy), n, dvx, dvy,
             ptr(dZ2), 26 * 64, 25, None, 0, st)
        dy1, dW2 = f32(max(n, 1), 64), torch.zeros(64, 26 * 64, device=dev)
        call("eventad_gemm_f32_ex", 0, 1, ptr(dZ2), 26 * 64, ptr(t["wc2"]), 26 * 64, ptr(dy1), 64, n, 64, 26 * 64, None, 0, 0, st)
        # block 1: y1 = ReLU(bn1(blend(Z1) + Z1[root]))
        d1 = f32(max(n, 1), 64)
        call("eventad_relu_scale_bwd", ptr(dy1), 64, ptr(L.y1), 64, ptr(t["bn1"][0]), None, n, 64, ptr(d1), None, st)
        dZ1 = f32(max(n, 1), 27 * 64)
        call("eventad_tail_blend_bwd", ptr(d1), 64, ptr(out_start), ptr(out_edge), ptr(edge_dst), ptr(L.csr_dxy), n, dvx, dvy,
             ptr(dZ1), 27 * 64, 25, ptr(ds), 26, st)
        dX, dW1 = f32(max(n, 1), cin_pad), torch.zeros(cin_pad, 27 * 64, device=dev)
        call("eventad_gemm_f32_ex", 0, 1, ptr(dZ1), 27 * 64, ptr(t["wc1"]), 27 * 64, ptr(dX), cin_pad, n, cin_pad, 27 * 64, None, 0, 0, st)
        if n > 0:                                            # dW = X^T . dZ for both blocks
            _wgrad_group([(L.y1, 64, dZ2, 26 * 64, dW2, 64, 26 * 64, n), (L.x, L.ldx, dZ1, 27 * 64, dW1, cin_pad, 27 * 64, n)], dev, st)
        from .weights import LAYER_NAMES
        q1 = "%s%s.conv_block1." % (nb.prefix, LAYER_NAMES[lv + 1])
        q2 = "%s%s.conv_block2." % (nb.prefix, LAYER_NAMES[lv + 1])
        grads = {                                            # columns of [W_0..W_24 | root | skip] back to the reference's tensors
            q1 + "conv.weight": dW1[:cin, :1600].reshape(cin, 25, 64).permute(1, 0, 2).contiguous(),
            q1 + "conv.lin.weight": dW1[:cin, 1600:1664].t().contiguous(),
            q2 + "lin.mlp.weight": dW1[:cin, 1664:1728].t().contiguous(),
            q2 + "conv.weight": dW2[:, :1600].reshape(64, 25, 64).permute(1, 0, 2).contiguous(),
            q2 + "conv.lin.weight": dW2[:, 1600:1664].t().contiguous(),
        }
        return dX[:n], grads

    def pool(self, lv, n_in, n_out, dX):
        """Backward of the pooling INTO level lv (lv = 1..3) and of the sampler that precedes it.  dX f32[n_out, cin_pad] =
        dL/d(level lv's input rows).  -> (dy f32[n_in,64] = dL/d(output of the Layer on level lv-1), dmap f32[B,h,w,64])."""
        nb, g, st = self.nb, self.nb.geom, torch.cuda.current_stream().cuda_stream
        view = self._levels()
        L, Lp, dev, B = view[lv], view[lv - 1], nb.device, nb._B
        fmap = nb._tail_maps_used[lv - 1]
        mean = 1 if g.pool_aggr[lv] == "mean" else 0
        dX = dX.contiguous()
        dy = torch.empty((max(n_in, 1), 64), device=dev)
        dsamp = torch.empty((max(n_in, 1), 64), device=dev)
        argm = torch.empty(max(n_out, 1) * 128, dtype=torch.int32, device=dev)
        call("eventad_pool_max_bwd", ptr(Lp.y), 64, 64, ptr(fmap), fmap.shape[1], fmap.shape[2], 64, ptr(Lp.npos), ptr(Lp.nbatch),
             ptr(Lp.node_slot_in), ptr(L.slot2node), ptr(L.cnt), n_in, n_out, g.W, g.H, mean, ptr(L.x), L.ldx, ptr(dX),
             int(dX.shape[1]), ptr(argm), ptr(dy), ptr(dsamp), st)
        dmap = torch.empty((B, fmap.shape[1], fmap.shape[2], 64), device=dev)
        call("eventad_sample_nodes_bwd", ptr(dsamp), 64, ptr(Lp.npos), ptr(Lp.nbatch), ptr(Lp.occ), Lp.nx, Lp.ny, B, g.W, g.H,
             fmap.shape[1], fmap.shape[2], ptr(dmap), st)
        return dy[:n_in], dmap

    def backward(self, g_out3, g_out4):
        """g_out3 / g_out4 f32[n,64] = dL/d(out3.x) / dL/d(out4.x) (either may be None), rows = the levels' valid nodes.
        -> dict(dx0 = dL/d(level-0 features) f32[n0,80], dmaps = [dL/d map2, map3, map4] (NHWC), weights = {key: grad})."""
        n, E = self.counts()
        dev = self.nb.device
        weights, dmaps = {}, [None, None, None]
        g_y = torch.zeros((n[3], 64), device=dev) if g_out4 is None else g_out4
        for lv in (3, 2, 1, 0):
            dX, gw = self.layer(lv, n[lv], E[lv], g_y)
            weights.update(gw)
            if lv == 0:
                return dict(dx0=dX[:, :80].contiguous(), dmaps=dmaps, weights=weights)
            g_y, dmaps[lv - 1] = self.pool(lv, n[lv - 1], n[lv], dX)
            if lv == 3 and g_out3 is not None:
                g_y = g_y + g_out3


class BackboneBackward:
    """Backward of the whole graph branch (Net.forward, net.py:106-197) on the workspace of the last NativeBackbone.forward():
    from dL/d(out3.x), dL/d(out4.x) to the gradients of the 25 weight tensors of the five Layers and of the five CNN feature
    maps (NHWC; a trainable CNN branch continues from there with its own autograd).  PooledBackward for the pooled levels;
    for the event level the glue around EventConvBackward / EventPoolBackward: ReLU and folded BatchNorm scales, the skip
    projection of conv_block2 (conv.py:47-56) as two small GEMMs."""

    def __init__(self, backbone, state_dict):
        load()
        nb = self.nb = backbone
        pre = nb.prefix
        self.sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if k.startswith(pre + "conv_block1.")}
        self.pooled, self.conv, self.pool1 = PooledBackward(nb), EventConvBackward(nb.geom, nb.device), EventPoolBackward(nb)
        p1, p2 = pre + "conv_block1.conv_block1.", pre + "conv_block1.conv_block2."
        self.p1, self.p2 = p1, p2
        dev = nb.device
        self.sc1 = spline.fold_bn(self.sd, p1 + "norm.module.")[0].contiguous().to(dev)
        self.sc2 = spline.fold_bn(self.sd, p2 + "norm.module.")[0].contiguous().to(dev)
        self.scs = spline.fold_bn(self.sd, p2 + "norm_skip.module.")[0].contiguous().to(dev)
        self.wskip = spline.pad_rows(self.sd[p2 + "lin.mlp.weight"].t().contiguous(), 20).t().contiguous().to(dev)   # [16, 20]

    def backward(self, g_out3, g_out4):
        """-> dict(weights = {state_dict key: gradient} (25 tensors), dmaps = [dL/d map0 .. map4] (NHWC))."""
        nb, st = self.nb, torch.cuda.current_stream().cuda_stream
        ws, g, dev, sd = nb.ws, nb.geom, nb.device, self.sd
        N, B = nb._N, nb._B
        res = self.pooled.backward(g_out3, g_out4)
        weights, dmaps = dict(res["weights"]), [None, None] + res["dmaps"]
        map0, map1 = nb._front_maps
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        # pool1: gradient rows of the level-0 nodes back to their voxel slots, then to the member events
        L0 = ws["levels"][0]
        n0 = res["dx0"].shape[0]
        dfeat = torch.zeros((L0.S, 80), device=dev)
        dfeat[L0.node_slot[:n0].long()] = res["dx0"]
        dy2, dsamp = self.pool1.pool1(N, B, dfeat, fmap=map1)
        dmaps[1] = self.pool1.sampler(N, B, dsamp, 64, map1.shape[1], map1.shape[2])
        # conv_block2: y2 = ReLU(bn2(conv2(y1)) + bns(x0 . Wskip^T))
        d2, ds = f32(max(N, 1), 16), f32(max(N, 1), 16)
        call("eventad_relu_scale_bwd", ptr(dy2), 16, ptr(ws["y2"]), 16, ptr(self.sc2), ptr(self.scs), N, 16, ptr(d2), ptr(ds), st)
        self.conv.transpose(ws, N)
        dy1, dw, dr = self.conv.backward(ws, N, ws["y1"][:N], d2[:N], sd[self.p2 + "conv.weight"], sd[self.p2 + "conv.lin.weight"])
        weights[self.p2 + "conv.weight"], weights[self.p2 + "conv.lin.weight"] = dw.to(dev), dr.to(dev)
        dx_skip, dwskip = f32(max(N, 1), 20), torch.zeros((16, 20), device=dev)
        call("eventad_gemm_f32_ex", 0, 0, ptr(ds), 16, ptr(self.wskip), 20, ptr(dx_skip), 20, N, 20, 16, None, 0, 0, st)
        if N > 0:
            _wgrad_group([(ds, 16, ws["x0s"], 20, dwskip, 16, 20, N)], dev, st)
        weights[self.p2 + "lin.mlp.weight"] = dwskip[:, :19].contiguous()
        # conv_block1: y1 = ReLU(bn1(conv1(x0)))
        d1 = f32(max(N, 1), 16)
        call("eventad_relu_scale_bwd", ptr(dy1), 16, ptr(ws["y1"]), 16, ptr(self.sc1), None, N, 16, ptr(d1), None, st)
        dx0, dw, dr = self.conv.backward(ws, N, ws["x0s"][:N], d1[:N], sd[self.p1 + "conv.weight"], sd[self.p1 + "conv.lin.weight"])
        weights[self.p1 + "conv.weight"], weights[self.p1 + "conv.lin.weight"] = dw.to(dev), dr.to(dev)
        dx0 = dx0 + dx_skip[:N]
        dmaps[0] = self.pool1.sampler(N, B, dx0, 16, map0.shape[1], map0.shape[2], column=1)
        return dict(weights=weights, dmaps=dmaps, dx_events=dx0)
