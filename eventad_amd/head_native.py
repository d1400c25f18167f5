"""Native (HIP) execution of the EventAD head: forward, backward and the clip+AdamW update
(reference models/EventAD.py:178-333 and utils/train.py:100-152), driving csrc/head.hip + csrc/gemm.hip
through the C ABI.  Host logic only: flat parameter / gradient / optimiser-state buffers (the model's
nn.Parameters are re-bound as views into them, so state_dict()/load_state_dict() keep working and a
data-parallel all-reduce is ONE collective on `flat_grads`), workspace, and the launch sequence.

Row convention: R = B*32 rows, row = clip*32 + track slot (slots 0 and 31 are never valid)."""
import ctypes
import os

import torch

from ._lib import call, ptr, stream_handle

NT = 32
from .ddp import PARAMS, flat_layout      # used head parameters in flat-buffer order = the all-reduce bucket layout


def _rp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_TILE_CNT = os.environ.get("EVENTAD_TILE_CNT", "0") == "1"
# chain products with K = 768: leave the split-K slabs for the consuming fused kernel instead of a reduce launch per stage.
# Measured on C2 (same box, A/B): head graph 0.695 ms with, 0.630 ms without -- the extra loads sit on the one-workgroup
# critical path and cost more than the 5 us launch they save.  Off.
_KEEP_SLABS = os.environ.get("EVENTAD_KEEP_SLABS", "0") == "1"
# Which of the head's six batched GEMM launches run on 64x64 tiles over the whole K (eventad_gemm_group64) instead of 32x32
# tiles with K split over the waves; bit i = launch i in program order.  C2, same box, head on its 32 CUs: the input
# projections (bit 0: K = 64 leaves three of the four waves of a 32x32 tile idle) 20 -> 10 us, the recurrent layers' weight
# gradients over all clips (bit 5: 630 small tiles) 50 -> 36 us; the fusion layers (bits 1-4: few tiles, long K) are latency
# bound and lose or stay (step 0.576 -> 0.559 ms with 33, 0.571 with all six).
_WIDE = int(os.environ.get("EVENTAD_GG64", "33"))
# forward recurrence: one fused launch per clip stage (attention of the previous clip + recurrent products + cell,
# csrc/headstage.hip) instead of a grouped GEMM and a cell+attention launch.  0 = the two-launch form (kept for A/B runs).
_STAGE_FWD = os.environ.get("EVENTAD_HEAD_STAGE", "1") == "1"
# backward recurrence likewise (needs the fused forward, which stores the raw recurrent products it reads): one launch per
# clip stage instead of cell+attention backward, a grouped GEMM and the launch that sums its split-K slabs
_STAGE_BWD = _STAGE_FWD and os.environ.get("EVENTAD_HEAD_STAGE_BWD", "1") == "1"


class NativeHead:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0):
        self.model = model
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, grad_clip
        params = dict(model.named_parameters())
        dev = params[PARAMS[0][0]].device
        assert dev.type == "cuda", "the native head needs the model on a GPU (no CPU path)"
        self.dev = dev
        for name, shape in PARAMS:
            assert tuple(params[name].shape) == shape, name
        self.off, off = flat_layout()                         # every tensor 16-byte aligned
        self.n = off
        self.flat = torch.zeros(off, device=dev)
        self.grads = torch.zeros(off, device=dev)
        self.exp_avg = torch.zeros(off, device=dev)
        self.exp_avg_sq = torch.zeros(off, device=dev)
        self.partial = torch.zeros(256, device=dev)
        self.opt_state = torch.zeros(8, device=dev)
        self.P, self.G = {}, {}
        for name, shape in PARAMS:
            p = params[name]
            n, o = p.numel(), self.off[name]
            v = self.flat[o:o + n].view(shape)
            v.copy_(p.data)
            p.data = v                                        # the module parameter IS the flat buffer slice
            g = self.grads[o:o + n].view(shape)            # native-path gradients live here (not in p.grad:
                                                           # autograd would add the bridge's result onto itself)
            self.P[name], self.G[name] = v, g
        self._B = None
        # dropout masks come from a counter-based generator (eventad_dropout_masks): the masks of step k depend on (seed, k)
        # only.  mask_step counts the steps whose masks were drawn; a captured draw reads its step number from a device word
        # that the replaying host sets (next_mask_step).
        self.mask_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        self.mask_step = 0
        self.mask_step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        # launch form of the clip recurrence: one fused launch per stage (default) or the older two / three launches per stage
        self.stage_fwd, self.stage_bwd = _STAGE_FWD, _STAGE_BWD
        self.result_out = None                                # f32[2] that optimizer_step also writes (loss, norm) to
        self.tables = None                                    # input tables filled ahead of forward() (fill_tables)
        self.keep_superseded_workspaces = False               # set once a hipGraph holds pointers into the workspace
        self._retired_ws = []

    def bound(self):
        """True while the model's parameters still alias the flat buffer (a later .to()/.cuda() breaks it)."""
        # (the head's 20 parameters by name: a dict of ALL named parameters walks the frozen backbone's several hundred
        # tensors too -- 0.75 ms per forward of an eager loop that takes 3.5)
        get = self.model.get_parameter
        return all(get(n).data_ptr() == self.P[n].data_ptr() for n, _ in PARAMS)

    # ---------------------------------------------------------------- workspace
    def _alloc(self, B):
        if self._B == B:
            return
        if self._B is not None and self.keep_superseded_workspaces:
            self._retired_ws.append(self.w)                   # a captured graph may still read and write it
        R, dev = B * NT, self.dev
        f = lambda *s: torch.zeros(s, device=dev)
        i = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
        w = self.w = {}
        w["X"], w["CO"], w["target"], w["valid"] = f(R, 64), f(R, 4), i(R), i(R)
        for k, n in (("GI0", 768), ("GH0", 768), ("G0", 768), ("GI1", 768), ("GH1", 768), ("G1", 768),
                     ("GIC", 96), ("GHC", 96), ("GC", 96), ("dGI0", 768), ("dGH0", 768), ("dGI1", 768),
                     ("dGH1", 768), ("dGIC", 96), ("dGHC", 96)):
            w[k] = f(R, n)
        for k, n in (("GHR0", 768), ("GHR1", 768), ("GHRC", 96)):      # raw recurrent products (no bias), for the fused backward stages
            w[k] = f(R, n)
        w["pd"] = f(3, 2, NT * NT)                                # per-workgroup partial row dots, [layer][stage parity]
        for k, n in (("S0", 256), ("S1", 256), ("SC", 32)):
            w[k] = f(R + NT, n)
        for k, n in (("H0n", 256), ("X1", 256), ("H1n", 256), ("HCn", 32), ("F1r", 256), ("dPre", 256),
                     ("dH1o", 256), ("dHCo", 32)):
            w[k] = f(R, n)
        w["ZCAT"], w["dZCAT"] = f(R, 512), f(R, 512)
        w["AL0"], w["AL1"], w["ALC"] = f(R), f(R), f(R)
        w["logits"], w["dlog"], w["loss_rows"], w["loss"] = f(R, 2), f(R, 2), f(R), f(1)
        for k, n in (("dS0a", 256), ("dS0b", 256), ("dS1a", 256), ("dS1b", 256), ("dSCa", 32), ("dSCb", 32),
                     ("dHn0", 256), ("dHn1", 256), ("dHnC", 32), ("dSp0", 256), ("dSp1", 256), ("dSpC", 32), ("dX1", 256)):
            w[k] = f(NT, n)
        # slabs for products with K > 256 (one scratch per grouped problem): ceil(K/256) x M x N floats; the largest are
        # the weight gradients (K = R rows, M x N = 768 x 256) and the chain's [32,768]x[768,256] (3 slabs of 32 x 256)
        w["part"] = [f(max(3 * 32 * 768, ((R + 255) // 256) * 768 * 256)) for _ in range(8)]
        w["cnt"] = [i(24 * 24) for _ in range(8)]                               # tile counters of those products (<= 768 x 768 outputs)
        # accumulators of d soft_attention.weight (one per GRU layer) and d soft_attention_cor.weight: [l0 | cor | l1 | 0]
        w["dwatt"] = f(2 * 288)
        w["dwatt0"], w["dwattc"], w["dwatt1"] = w["dwatt"][:256], w["dwatt"][256:288], w["dwatt"][288:544]
        w["zero256"], w["zero32"] = f(NT, 256), f(NT, 32)
        w["D"] = f(2 * R, 256)                                                  # dropout masks (GRU inter-layer | fusion)
        self._B = B

    # ------------------------------------------------------------ launch helpers
    def _gg(self, *probs, wide=False):
        """One grouped-GEMM launch (wide: on 64x64 tiles over the whole K instead of 32x32 tiles with K split over the waves).  prob = (ta, tb, A, lda, B, ldb, C, ldc, M, N, K[, bias[, relu[, acc[, keep]]]]);
        keep = leave the split-K slabs of a K > 256 product un-summed in self.w["part"][i] for the consumer."""
        d = (ctypes.c_int64 * (16 * len(probs)))()
        for i, pr in enumerate(probs):
            ta, tb, A, lda, Bm, ldb, C, ldc, M, N, K = pr[:11]
            bias = pr[11] if len(pr) > 11 else None
            relu = pr[12] if len(pr) > 12 else 0
            acc = pr[13] if len(pr) > 13 else 0
            keep = (pr[14] if len(pr) > 14 else 0) and _KEEP_SLABS
            if K > 256 and ((K + 255) // 256) * M * N > self.w["part"][i].numel():
                raise RuntimeError("grouped GEMM: split-K scratch too small for M=%d N=%d K=%d" % (M, N, K))
            if K > 256 and ((M + 31) // 32) * ((N + 31) // 32) > self.w["cnt"][i].numel():
                raise RuntimeError("grouped GEMM: tile-counter buffer too small for M=%d N=%d" % (M, N))
            d[16 * i:16 * i + 16] = [A.data_ptr(), Bm.data_ptr(), self.w["part"][i].data_ptr() if K > 256 else 0, C.data_ptr(),
                                     bias.data_ptr() if bias is not None else 0, lda, ldb, ldc, M, N, K, ta, tb, acc, relu,
                                     1 if (keep and K > 256) else (self.w["cnt"][i].data_ptr() if (K > 256 and _TILE_CNT) else 0)]
        call("eventad_gemm_group64" if wide else "eventad_gemm_group", len(probs), ctypes.cast(d, ctypes.c_void_p),
             stream_handle())

    @staticmethod
    def _desc(fn, width, rows, *extra):
        d = (ctypes.c_int64 * (width * len(rows)))()
        for i, row in enumerate(rows):
            for j, v in enumerate(row):
                d[width * i + j] = v if isinstance(v, int) else (v.data_ptr() if v is not None else 0)
        call(fn, len(rows), ctypes.cast(d, ctypes.c_void_p), *extra, stream_handle())

    def _att(self, fn, valid, *probs):
        """Grouped attentions: prob = dict(Hn, Sprev, w, alpha, Snext, dSnext, dHn_out, dHn, dSpass, dw, H)."""
        keys = ("Hn", "Sprev", "w", "alpha", "Snext", "dSnext", "dHn_out", "dHn", "dSpass", "dw", "H")
        self._desc(fn, 12, [[pr.get(k) for k in keys] for pr in probs], _rp(valid))

    @staticmethod
    def _gemm(ta, tb, A, lda, Bm, ldb, C, ldc, M, N, K, bias=None, relu=0, acc=0):
        call("eventad_gemm_f32_ex", ta, tb, _rp(A), lda, _rp(Bm), ldb, _rp(C), ldc, M, N, K, _rp(bias), relu, acc,
             stream_handle())

    # ------------------------------------------------------------------ forward
    def fill_masks(self, masks, step_dev=None):
        """Dropout masks of one step (GRU inter-layer | fusion MLP, keep probability 0.7, pre-scaled) into `masks`
        f32[2*B*32, 256]: one launch that depends on nothing, so a pipelined caller can issue it ahead of the head.
        step_dev: device int64[1] holding the step number (for launches that are captured and replayed: the host sets it
        with next_mask_step before every replay); default: this head's own step count -- under capture its device word."""
        if step_dev is None and torch.cuda.is_current_stream_capturing():
            step_dev = self.mask_step_dev
        step = 0
        if step_dev is None:
            step, self.mask_step = self.mask_step, self.mask_step + 1
        call("eventad_dropout_masks", ptr(masks), masks.numel(), 0.7, self.mask_seed, step, ptr(step_dev),
             stream_handle())
        return masks

    def next_mask_step(self, step_dev=None):
        """Before replaying a graph that holds a captured fill_masks: hand it the next step number (on the current stream)."""
        (self.mask_step_dev if step_dev is None else step_dev).fill_(self.mask_step)
        self.mask_step += 1

    def new_tables(self, B):
        """A set of the head's input tables (track features, box coordinates, targets, validity) for fill_tables."""
        R, dev = B * NT, self.dev
        return {"X": torch.zeros((R, 64), device=dev), "CO": torch.zeros((R, 4), device=dev),
                "target": torch.zeros(R, dtype=torch.int32, device=dev), "valid": torch.zeros(R, dtype=torch.int32, device=dev)}

    def fill_tables(self, tables, feats, bbox, bbox_batch, labels, B, width, height):
        """The first launch of forward() on its own, into `tables`: it depends on the batch and its ROI features only, not on
        the weights, so a pipelined caller issues it behind the backbone and hands the set to forward() through self.tables."""
        self._alloc(B)
        self._launch_tables(feats, bbox, bbox_batch, labels, B, width, height, tables)

    def _launch_tables(self, feats, bbox, bbox_batch, labels, B, width, height, dst):
        M = int(bbox.shape[0]) if bbox is not None else 0
        bbox_c = bbox.contiguous().float() if M else None
        bb_c = bbox_batch.contiguous().long() if M else None
        lab_c, label_col = None, -1
        if M and labels is not None:
            if (labels.dtype == torch.float32 and labels.dim() == 1 and labels.shape[0] == M and labels.stride(0) == 6
                    and labels.data_ptr() - bbox_c.data_ptr() in range(0, 24, 4)):
                label_col = (labels.data_ptr() - bbox_c.data_ptr()) // 4      # a column of bbox itself: read in place
            else:
                lab_c = labels.contiguous().float()
        call("eventad_head_tables", _rp(bbox_c), _rp(bb_c), M, _rp(lab_c), label_col, ptr(feats), B, int(width), int(height),
             ptr(dst["X"]), ptr(dst["CO"]), ptr(dst["target"]), ptr(dst["valid"]), stream_handle())

    def forward(self, feats, bbox, bbox_batch, labels, B, width, height, train, lean=False, masks=None):
        """feats f32[B,2,31,64] (ROI features), bbox f32[M,6], bbox_batch i64[M], labels f32[M] or None.
        masks: this step's dropout masks, already drawn by fill_masks (default: drawn here, when train).
        Returns (loss 0-d, logits [B,32,2], valid bool[B,32], targets i64[B,32]); keeps what backward needs."""
        self._alloc(B)
        P = self.P
        st = stream_handle()
        R = B * NT
        if self.tables is None:
            self._launch_tables(feats, bbox, bbox_batch, labels, B, width, height, self.w)
            w = self.w
        else:
            # the tables of this batch were filled ahead of time (fill_tables, e.g. at the end of the backbone's graph): the
            # head's launches below read them in place of the workspace's own
            w = dict(self.w)
            w.update(self.tables)
        self._w_fw = w
        self.D1 = self.D2 = None
        if train:                                              # nn.GRU(dropout=.3) between layers; fusion Dropout(.3):
            D = masks if masks is not None else self.fill_masks(w["D"])      # both masks in two launches
            assert tuple(D.shape) == (2 * R, 256) and D.is_contiguous()
            self.D1, self.D2 = D[:R], D[R:]
        ev, co, fu = "gru_net_event.gru.", "gru_net_cor.gru.", "fusion_module."
        watt, wattc = P["soft_attention.weight"], P["soft_attention_cor.weight"]
        # input projections of every clip at once: they ride in the first stage's launch (independent problems)
        input_proj = [(0, 1, w["X"], 64, P[ev + "weight_ih_l0"], 64, w["GI0"], 768, R, 768, 64, P[ev + "bias_ih_l0"]),
                      (0, 1, w["CO"], 4, P[co + "weight_ih_l0"], 4, w["GIC"], 96, R, 96, 4, P[co + "bias_ih_l0"])]
        # rows [0, NT) of S0/S1/SC are the initial hidden state: zero at allocation and never written
        # The clip-to-clip hidden-state chain (EventAD.py:211-323) as a wavefront: stage b runs GRU layer 0 and the
        # coordinate GRU on clip b and GRU layer 1 on clip b-1 (its input, the dropout'd layer-0 output of clip
        # b-1, and its state, the layer-1 attention of clip b-2, are both a stage old), so one stage is two
        # dependent launches -- a grouped GEMM and a fused cell+attention -- instead of five.
        fused_fwd = self.stage_fwd
        self._fw_fused = fused_fwd
        if fused_fwd:
            self._gg(*input_proj, wide=_WIDE & 1)
            self._forward_chain_fused(B, train)
        for b in range(0 if fused_fwd else B + 1):
            gemms, steps = (list(input_proj) if b == 0 else []), []
            if b < B:
                r0, r1 = b * NT, (b + 1) * NT
                gemms += [(0, 1, w["S0"][r0:r1], 256, P[ev + "weight_hh_l0"], 256, w["GH0"][r0:r1], 768, NT, 768, 256, P[ev + "bias_hh_l0"]),
                          (0, 1, w["SC"][r0:r1], 32, P[co + "weight_hh_l0"], 32, w["GHC"][r0:r1], 96, NT, 96, 32, P[co + "bias_hh_l0"])]
                vb = w["valid"][r0:r1]
                steps += [[w["GI0"][r0:r1], w["GH0"][r0:r1], w["S0"][r0:r1], w["G0"][r0:r1], w["H0n"][r0:r1],
                           self.D1[r0:r1] if train else None, w["X1"][r0:r1], watt, w["AL0"][r0:r1], w["S0"][r1:r1 + NT], vb, 256],
                          [w["GIC"][r0:r1], w["GHC"][r0:r1], w["SC"][r0:r1], w["GC"][r0:r1], w["HCn"][r0:r1],
                           None, None, wattc, w["ALC"][r0:r1], w["SC"][r1:r1 + NT], vb, 32]]
            if b >= 1:
                q0, q1 = (b - 1) * NT, b * NT
                gemms += [(0, 1, w["S1"][q0:q1], 256, P[ev + "weight_hh_l1"], 256, w["GH1"][q0:q1], 768, NT, 768, 256, P[ev + "bias_hh_l1"]),
                          (0, 1, w["X1"][q0:q1], 256, P[ev + "weight_ih_l1"], 256, w["GI1"][q0:q1], 768, NT, 768, 256, P[ev + "bias_ih_l1"])]
                steps += [[w["GI1"][q0:q1], w["GH1"][q0:q1], w["S1"][q0:q1], w["G1"][q0:q1], w["H1n"][q0:q1],
                           None, None, watt, w["AL1"][q0:q1], w["S1"][q1:q1 + NT], w["valid"][q0:q1], 256]]
            self._gg(*gemms)
            self._desc("eventad_gru_att_fwd_group", 12, steps)
        # fusion MLP on all rows at once (Fuse_GRUNet.forward, EventAD.py:30-59)
        self._gg((0, 1, w["H1n"], 256, P[fu + "event_proj.weight"], 256, w["ZCAT"], 512, R, 256, 256, P[fu + "event_proj.bias"]),
                 (0, 1, w["HCn"], 32, P[fu + "coord_proj.weight"], 32, w["ZCAT"][:, 256:], 512, R, 256, 32, P[fu + "coord_proj.bias"]),
                 wide=_WIDE & 2)
        self._gg((0, 1, w["ZCAT"], 512, P[fu + "fusion.0.weight"], 512, w["F1r"], 256, R, 256, 512, P[fu + "fusion.0.bias"], 1), wide=_WIDE & 4)
        # a training loop (lean) reads the loss after the step: its rows are then added by the first launch of backward()
        # instead of a launch of their own here
        self._loss_deferred = bool(lean and train)
        call("eventad_fuse_out", ptr(w["F1r"]), _rp(self.D2), _rp(P[fu + "fusion.3.weight"]), _rp(P[fu + "fusion.3.bias"]),
             ptr(w["target"]), ptr(w["valid"]), R, ptr(w["logits"]), ptr(w["dlog"]), ptr(w["loss_rows"]),
             None if self._loss_deferred else ptr(w["loss"]),
             ptr(w["dPre"]), st)                               # dPre: the first step of backward(), while the rows are in registers
        self._fw = (B, train)
        if lean:                                               # training loop: no dtype conversions (3 launches less)
            return (w["loss"][0], w["logits"].view(B, NT, 2), w["valid"].view(B, NT), w["target"].view(B, NT))
        return (w["loss"][0], w["logits"].view(B, NT, 2), w["valid"].view(B, NT).bool(), w["target"].view(B, NT).long())

    def _forward_chain_fused(self, B, train):
        """The clip-to-clip recurrence with one launch per stage (eventad_head_stage_fwd): stage b runs GRU layer 0 and the
        coordinate GRU on clip b and GRU layer 1 on clip b-1; the attention that turns a clip's cell outputs into the next
        clip's state is computed at the start of the stage that consumes it.  Fills the same buffers as the two-launch form
        (the state after the LAST clip and its attention weights, which nothing reads, are not computed)."""
        w, P = self._w_fw, self.P                              # (the forward's view of the workspace: its input tables may be a set of their own)
        ev, co = "gru_net_event.gru.", "gru_net_cor.gru."
        watt, wattc = P["soft_attention.weight"], P["soft_attention_cor.weight"]
        st = stream_handle()
        rows = lambda t, b: t[b * NT:(b + 1) * NT].data_ptr()
        for b in range(B + 1):
            d = (ctypes.c_int64 * 57)()
            if b < B:
                prev = b > 0
                d[0:19] = [1, rows(w["H0n"], b - 1) if prev else 0, rows(w["S0"], b - 1) if prev else 0,
                           rows(w["valid"], b - 1) if prev else 0, watt.data_ptr(), rows(w["AL0"], b - 1) if prev else 0,
                           rows(w["S0"], b), P[ev + "weight_hh_l0"].data_ptr(), P[ev + "bias_hh_l0"].data_ptr(),
                           rows(w["GI0"], b), 0, 0, 0, rows(w["GH0"], b), rows(w["G0"], b), rows(w["H0n"], b),
                           rows(self.D1, b) if train else 0, rows(w["X1"], b), rows(w["GHR0"], b)]
                d[38:57] = [1, rows(w["HCn"], b - 1) if prev else 0, rows(w["SC"], b - 1) if prev else 0,
                            rows(w["valid"], b - 1) if prev else 0, wattc.data_ptr(), rows(w["ALC"], b - 1) if prev else 0,
                            rows(w["SC"], b), P[co + "weight_hh_l0"].data_ptr(), P[co + "bias_hh_l0"].data_ptr(),
                            rows(w["GIC"], b), 0, 0, 0, rows(w["GHC"], b), rows(w["GC"], b), rows(w["HCn"], b), 0, 0,
                            rows(w["GHRC"], b)]
            if b >= 1:
                q = b - 1
                prev = q > 0
                d[19:38] = [1, rows(w["H1n"], q - 1) if prev else 0, rows(w["S1"], q - 1) if prev else 0,
                            rows(w["valid"], q - 1) if prev else 0, watt.data_ptr(), rows(w["AL1"], q - 1) if prev else 0,
                            rows(w["S1"], q), P[ev + "weight_hh_l1"].data_ptr(), P[ev + "bias_hh_l1"].data_ptr(),
                            0, rows(w["X1"], q), P[ev + "weight_ih_l1"].data_ptr(), P[ev + "bias_ih_l1"].data_ptr(),
                            rows(w["GH1"], q), rows(w["G1"], q), rows(w["H1n"], q), 0, 0, rows(w["GHR1"], q)]
            call("eventad_head_stage_fwd", ctypes.cast(d, ctypes.c_void_p), st)

    # ----------------------------------------------------------------- backward
    def backward(self):
        """Fills `self.grads` (d loss / d every used parameter) for the last forward."""
        B, train = self._fw
        w, P, G = self._w_fw, self.P, self.G
        st = stream_handle()
        R = B * NT
        ev, co, fu = "gru_net_event.gru.", "gru_net_cor.gru.", "fusion_module."
        # every used gradient entry is written (not accumulated) below; the attention vectors accumulate over the clips in
        # small buffers of their own (cleared here), the alignment pads of the flat buffer are zero from allocation
        deferred = getattr(self, "_loss_deferred", False)
        call("eventad_out_bwd", ptr(w["dlog"]), _rp(P[fu + "fusion.3.weight"]), ptr(w["F1r"]), _rp(self.D2), R, None,
             _rp(G[fu + "fusion.3.weight"]), _rp(G[fu + "fusion.3.bias"]), ptr(w["dwatt"]), w["dwatt"].numel(),
             ptr(w["loss_rows"]) if deferred else None, ptr(w["loss"]) if deferred else None, st)
        self._loss_deferred = False
        dE, dC = w["dZCAT"], w["dZCAT"][:, 256:]
        self._gg((1, 0, w["dPre"], 256, w["ZCAT"], 512, G[fu + "fusion.0.weight"], 512, 256, 512, R),
                 (0, 0, w["dPre"], 256, P[fu + "fusion.0.weight"], 512, w["dZCAT"], 512, R, 512, 256), wide=_WIDE & 8)
        self._gg((1, 0, dE, 512, w["H1n"], 256, G[fu + "event_proj.weight"], 256, 256, 256, R),
                 (0, 0, dE, 512, P[fu + "event_proj.weight"], 256, w["dH1o"], 256, R, 256, 256),
                 (1, 0, dC, 512, w["HCn"], 32, G[fu + "coord_proj.weight"], 32, 256, 32, R),
                 (0, 0, dC, 512, P[fu + "coord_proj.weight"], 32, w["dHCo"], 32, R, 32, 256), wide=_WIDE & 16)
        # (the bias gradients of these three layers are column sums of dPre / dZCAT, which stay as they are until the end
        # of backward(): they ride in its last column-sum launch)
        # dS of the state after the last clip is zero: a constant zero buffer read by the first stage of each chain
        # (z* are never written), then the a/b buffers alternate
        d0, d1, dc = w["zero256"], w["zero256"], w["zero32"]
        n0, n1, nc = w["dS0a"], w["dS1a"], w["dSCa"]
        o0, o1, oc = w["dS0b"], w["dS1b"], w["dSCb"]
        watt, wattc = P["soft_attention.weight"], P["soft_attention_cor.weight"]
        gattc = w["dwattc"]
        # reverse wavefront: stage b runs layer 1 + the coordinate GRU on clip b and layer 0 on clip b+1 (whose
        # dX1 and dS0 come out of the previous stage's GEMMs): fused attention+cell backward, then one grouped GEMM
        # The [32,768]x[768,256] products of the chain have 3 split-K slabs; with _KEEP_SLABS the NEXT stage's fused kernel
        # adds them while it loads (same order as the reduce kernel: same bits) instead of a reduce launch per stage.
        SL = 3                                                  # ceil(768 / 256)
        p1 = p0 = px = None                                    # slabs of dS1 / dS0 / dX1 from the previous stage
        fused_bwd = self.stage_bwd and self._fw_fused         # the fused backward reads what only the fused forward stores
        if fused_bwd:
            self._backward_chain_fused(B, train)
        for b in range(B - 1, -2 if not fused_bwd else B - 1, -1):
            steps, gemms = [], []
            i1 = i0 = ix = None
            if b >= 0:
                r0, r1 = b * NT, (b + 1) * NT
                vb = w["valid"][r0:r1]
                steps += [[w["H1n"][r0:r1], watt, w["AL1"][r0:r1], d1, w["dH1o"][r0:r1], vb, w["dwatt1"], None, None,
                           w["G1"][r0:r1], w["GH1"][r0:r1], w["S1"][r0:r1], w["dGI1"][r0:r1], w["dGH1"][r0:r1], n1, 256,
                           p1, SL if p1 is not None else 0, None, 0],
                          [w["HCn"][r0:r1], wattc, w["ALC"][r0:r1], dc, w["dHCo"][r0:r1], vb, gattc, None, None,
                           w["GC"][r0:r1], w["GHC"][r0:r1], w["SC"][r0:r1], w["dGIC"][r0:r1], w["dGHC"][r0:r1], nc, 32,
                           None, 0, None, 0]]
                if b > 0:                                      # dS of the initial state (clip 0) is never used
                    i1 = len(gemms)
                    gemms += [(0, 0, w["dGH1"][r0:r1], 768, P[ev + "weight_hh_l1"], 256, n1, 256, NT, 256, 768, None, 0, 1, 1),
                              (0, 0, w["dGHC"][r0:r1], 96, P[co + "weight_hh_l0"], 32, nc, 32, NT, 32, 96, None, 0, 1)]
                ix = len(gemms)
                gemms += [(0, 0, w["dGI1"][r0:r1], 768, P[ev + "weight_ih_l1"], 256, w["dX1"], 256, NT, 256, 768, None, 0, 0, 1)]
            if b + 1 < B:
                q0, q1 = (b + 1) * NT, (b + 2) * NT
                steps += [[w["H0n"][q0:q1], watt, w["AL0"][q0:q1], d0, None, w["valid"][q0:q1], w["dwatt0"], w["dX1"],
                           self.D1[q0:q1] if train else None, w["G0"][q0:q1], w["GH0"][q0:q1], w["S0"][q0:q1],
                           w["dGI0"][q0:q1], w["dGH0"][q0:q1], n0, 256,
                           p0, SL if p0 is not None else 0, px, SL if px is not None else 0]]
                if b + 1 > 0:
                    i0 = len(gemms)
                    gemms += [(0, 0, w["dGH0"][q0:q1], 768, P[ev + "weight_hh_l0"], 256, n0, 256, NT, 256, 768, None, 0, 1, 1)]
            self._desc("eventad_gru_att_bwd_group", 20, steps)
            if gemms:
                self._gg(*gemms)
            # slabs are per problem slot of the launch; the next stage reads them before its own launch overwrites them
            p1 = w["part"][i1] if (i1 is not None and _KEEP_SLABS) else None
            px = w["part"][ix] if (ix is not None and _KEEP_SLABS) else None
            p0 = w["part"][i0] if (i0 is not None and _KEEP_SLABS) else None
            if b >= 0:
                d1, n1, o1 = n1, o1, n1
                dc, nc, oc = nc, oc, nc
            if b + 1 < B:
                d0, n0, o0 = n0, o0, n0
        # soft_attention.weight (256) and soft_attention_cor.weight (32) are neighbours in the flat gradient buffer:
        # [dwatt0 | dwattc] + [dwatt1 | 0] lands on both with one launch
        o = self.off["soft_attention.weight"]
        assert self.off["soft_attention_cor.weight"] == o + 256
        # weight gradients of the recurrent layers, batched over all clips: ONE launch (+ one reduce)
        self._gg((1, 0, w["dGI0"], 768, w["X"], 64, G[ev + "weight_ih_l0"], 64, 768, 64, R),
                 (1, 0, w["dGH0"], 768, w["S0"], 256, G[ev + "weight_hh_l0"], 256, 768, 256, R),
                 (1, 0, w["dGI1"], 768, w["X1"], 256, G[ev + "weight_ih_l1"], 256, 768, 256, R),
                 (1, 0, w["dGH1"], 768, w["S1"], 256, G[ev + "weight_hh_l1"], 256, 768, 256, R),
                 (1, 0, w["dGIC"], 96, w["CO"], 4, G[co + "weight_ih_l0"], 4, 96, 4, R),
                 (1, 0, w["dGHC"], 96, w["SC"], 32, G[co + "weight_hh_l0"], 32, 96, 32, R), wide=_WIDE & 32)
        # every bias gradient of the head (a column sum over the R rows) in one launch, together with the attention vectors:
        # [dwatt0 | dwattc] + [dwatt1 | 0] is the column sum of the accumulator seen as a 2-row matrix
        self._desc("eventad_colsum_group", 5,
                   [[w["dGI0"], G[ev + "bias_ih_l0"], 768, 768, 0], [w["dGH0"], G[ev + "bias_hh_l0"], 768, 768, 0],
                    [w["dGI1"], G[ev + "bias_ih_l1"], 768, 768, 0], [w["dGH1"], G[ev + "bias_hh_l1"], 768, 768, 0],
                    [w["dGIC"], G[co + "bias_ih_l0"], 96, 96, 0], [w["dGHC"], G[co + "bias_hh_l0"], 96, 96, 0],
                    [w["dPre"], G[fu + "fusion.0.bias"], 256, 256, 0], [dE, G[fu + "event_proj.bias"], 512, 256, 0],
                    [dC, G[fu + "coord_proj.bias"], 512, 256, 0], [w["dwatt"], self.grads[o:o + 288], 288, 288, 2]], R)

    def _backward_chain_fused(self, B, train):
        """The reverse recurrence with one launch per stage (eventad_head_stage_bwd): stage b runs GRU layer 1 and the
        coordinate GRU on clip b and GRU layer 0 on clip b+1.  A stage first forms, for its columns, the products that
        carry the previous stage's gate gradients to this clip's state (and, for layer 0, to its dropout'd output), then
        does attention + cell backward; the attention's row dots come from partial dots the previous stage left."""
        w, P = self._w_fw, self.P                              # (the forward's view of the workspace: its input tables may be a set of their own)
        ev, co = "gru_net_event.gru.", "gru_net_cor.gru."
        watt, wattc = P["soft_attention.weight"], P["soft_attention_cor.weight"]
        st = stream_handle()
        rows = lambda t, b: t[b * NT:(b + 1) * NT].data_ptr()
        dS1, dS0, dSC = (w["dS1a"], w["dS1b"]), (w["dS0a"], w["dS0b"]), (w["dSCa"], w["dSCb"])
        pd = w["pd"]

        def slot(c, hn, wa, al, dhn_out, mask, gates, gh, hprev, dgi, dgh, dS, dw, whh, li, ghr, dgi_x=None, wih_x=None):
            nxt, prv = c + 1 < B, c >= 1
            return [1, rows(hn, c), wa.data_ptr(), rows(al, c), rows(w["valid"], c), rows(dhn_out, c) if dhn_out is not None else 0,
                    rows(mask, c) if mask is not None else 0, rows(gates, c), rows(gh, c), rows(hprev, c), rows(dgi, c), rows(dgh, c),
                    dS[c % 2].data_ptr(), dw.data_ptr(),
                    int(nxt), dS[(c + 1) % 2].data_ptr() if nxt else 0, rows(dgh, c + 1) if nxt else 0, whh.data_ptr() if nxt else 0,
                    pd[li, (c + 1) % 2].data_ptr() if nxt else 0,
                    int(dgi_x is not None), rows(dgi_x, c) if dgi_x is not None else 0, wih_x.data_ptr() if wih_x is not None else 0,
                    int(prv), rows(ghr, c) if prv else 0, rows(hn, c - 1) if prv else 0, rows(al, c - 1) if prv else 0,
                    rows(w["valid"], c - 1) if prv else 0, pd[li, c % 2].data_ptr() if prv else 0]

        for b in range(B - 1, -2, -1):
            d = (ctypes.c_int64 * 84)()
            if b >= 0:
                d[0:28] = slot(b, w["H1n"], watt, w["AL1"], w["dH1o"], None, w["G1"], w["GH1"], w["S1"], w["dGI1"], w["dGH1"], dS1,
                               w["dwatt1"], P[ev + "weight_hh_l1"], 0, w["GHR1"])
                d[56:84] = slot(b, w["HCn"], wattc, w["ALC"], w["dHCo"], None, w["GC"], w["GHC"], w["SC"], w["dGIC"], w["dGHC"], dSC,
                                w["dwattc"], P[co + "weight_hh_l0"], 2, w["GHRC"])
            if b + 1 < B:
                d[28:56] = slot(b + 1, w["H0n"], watt, w["AL0"], None, self.D1 if train else None, w["G0"], w["GH0"], w["S0"],
                                w["dGI0"], w["dGH0"], dS0, w["dwatt0"], P[ev + "weight_hh_l0"], 1, w["GHR0"],
                                dgi_x=w["dGI1"], wih_x=P[ev + "weight_ih_l1"])
            call("eventad_head_stage_bwd", ctypes.cast(d, ctypes.c_void_p), st)

    # ---------------------------------------------------------------- optimiser
    def optimizer_step(self):
        """clip_grad_norm_(grad_clip) + AdamW on the flat buffers; returns the (pre-clip) gradient norm."""
        # result_out (set by a caller that keeps several steps' results, TrainStep.capture_dual): the update's own launch
        # leaves (loss, norm) there
        res = self.result_out
        call("eventad_clip_adamw_out", ptr(self.flat), ptr(self.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n,
             float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), float(self.clip),
             ptr(self.partial), ptr(self.opt_state), ptr(self.w["loss"]) if res is not None else None, ptr(res), stream_handle())
        return self.opt_state[1]

    def nonfinite_steps(self):
        return int(self.opt_state[5].item())
