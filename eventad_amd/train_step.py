"""One optimisation step of EventAD on the GPU, data-parallel over clips.

Reference step body (utils/train.py:100-152, setup_optimizer :27-53): zero_grad -> forward -> backward (head only:
the DAGR backbone is frozen and under no_grad) -> clip_grad_norm_(1.0) -> NaN scan -> AdamW(lr 1e-3, wd 1e-5).
Here: no `detect_anomaly`, no per-parameter host syncs (a non-finite gradient norm skips the update on the device and
is counted in a flag the host reads lazily), and -- the one addition -- with world_size > 1 the gradients of the used
head parameters are summed over ranks in one flat-bucket all-reduce before clipping (eventad_amd/ddp.py).

Launch forms, all leaving bit-identical weights:
  step(batch)                               eager, one stream
  capture(batch) / replay()                 the same step as one hipGraph
  prime(b0); pipelined_step(b_k, b_k+1)     eager, two streams: backbone + ROI of batch k+1 || head update of batch k
  capture_dual(b0); replay_dual(b_k+1)      that pipeline as two hipGraphs on two event-coupled streams
The backbone is frozen (models/EventAD.py:148-150), so the features of batch k+1 do not depend on the head update of
batch k: overlapping them changes nothing but the wall clock.  There is no CPU path: a CPU model raises."""
import os

import torch
import torch.distributed as dist

from ._lib import call
from .ddp import GradSync


class _HeadInputs:
    """What the head reads of a batch besides its features: the current-frame boxes (labels = bbox[:, 4])."""

    def __init__(self, data):
        self.num_graphs = int(data.num_graphs)
        self.bbox = data.bbox.detach().clone().float().contiguous()
        self.bbox_batch = data.bbox_batch.detach().clone().long().contiguous()

    def copy_from(self, data):
        self.bbox.copy_(data.bbox, non_blocking=True)
        self.bbox_batch.copy_(data.bbox_batch, non_blocking=True)


class TrainStep:
    def __init__(self, model, lr=1e-3, weight_decay=1e-5, grad_clip=1.0, process_group=None):
        self.model = model
        self.grad_clip = grad_clip
        if next(model.gru_net_event.parameters()).device.type != "cuda":
            raise RuntimeError("TrainStep: the training step runs on the GPU only (no CPU fallback); move the model to cuda")
        self.nh = model.native_head(lr=lr, weight_decay=weight_decay, grad_clip=grad_clip)
        # EVENTAD_FORCE_ALLREDUCE=1: issue the collective on a single rank too (tests of its capture into a hipGraph)
        self.sync = GradSync(self.nh.grads, process_group, force=os.environ.get("EVENTAD_FORCE_ALLREDUCE") == "1")
        self.world, self.bucket, self.bucket_bytes = self.sync.world, self.sync.bucket, self.sync.bytes
        self.graph = None
        self._retired = []                # superseded graphs stay alive: buffers first allocated under them stay mapped
        self._dual = None
        # two-graph pipeline: CUs set aside for the head's stream.  The head is a chain of ~70 small dependent launches;
        # sharing CUs with the backbone's chip-wide grids, each of them queues behind whatever grid is being dispatched and
        # the chain stretches from 0.55 to 1.05 ms (C2) -- longer than the backbone graph.  -1 (default): decided at
        # capture from the two graphs' stand-alone times (a head that takes less than 0.3 of its backbone's time gets no CUs of its
        # own: they would only be taken from the backbone); 0: never; n > 0: always n CUs.
        self.head_cus = int(os.environ.get("EVENTAD_HEAD_CUS", "-1"))
        # two-graph pipeline: number of backbone chains (each with its own workspace set, input buffers, feature buffer and
        # stream; batch k runs on chain k % n).  The backbone graph is ~45 launches, half of them latency bound: with three
        # chains the backbone passes of consecutive batches overlap and fill each other's gaps (C2: 0.84 -> 0.62 ms per
        # batch for two passes in flight), while the head updates stay strictly ordered.  0 (default): 3 chains when the
        # CNN features come precomputed with the batch and the head gets CUs of its own, else 2 (= plain double buffering:
        # with the live CNN trunk the backbone's kernels fill the chip and overlapping them gains nothing).
        self.backbone_chains = int(os.environ.get("EVENTAD_BACKBONE_CHAINS", "0"))

    # ------------------------------------------------------------------ pieces
    def _head_update(self, feats, head_in, labels=None, masks=None):
        """Head forward + backward + gradient all-reduce + clip/AdamW on `feats`; -> (loss, grad_norm) device scalars.
        masks: dropout masks drawn ahead of time (NativeHead.fill_masks), else they are drawn here."""
        if labels is None:
            labels = head_in.bbox[:, 4]
        loss, _, _, _ = self.model._head_native(feats, head_in, labels, lean=True, masks=masks)
        self.nh.backward()
        norm = self.sync.reduce_and_update(self.nh.optimizer_step)
        return loss, norm

    def _pin_workspaces(self):
        """A captured graph holds raw pointers into the native workspaces: from now on a workspace that has to grow is
        retired (kept alive) instead of freed, so earlier captures stay replayable."""
        nb = getattr(self.model.dagr_model.backbone, "_native", None)
        if nb is not None:
            nb.keep_superseded_workspaces = True
        self.nh.keep_superseded_workspaces = True

    def snapshot(self):
        """Head weights + optimiser state (flat buffers, ~14 MB) and the dropout-mask step count, e.g. to undo warm-up steps."""
        nh = self.nh
        return [t.clone() for t in (nh.flat, nh.exp_avg, nh.exp_avg_sq, nh.opt_state)] + [nh.mask_step]

    def restore(self, snap):
        nh = self.nh
        for dst, src in zip((nh.flat, nh.exp_avg, nh.exp_avg_sq, nh.opt_state), snap):
            dst.copy_(src)
        if len(snap) > 4:
            nh.mask_step = snap[4]

    # -------------------------------------------------------------------- eager
    def step(self, data, labels=None):
        """data: formatted batch on the device.  Returns (loss, grad_norm) as device tensors."""
        with torch.no_grad():
            feats = self.model.extract_features(data, frames=(1,))      # frame 0 features are never read by the head (EventAD.py:216)
            return self._head_update(feats, data, labels)

    # ---- hipGraph: the whole step (backbone + ROI + head fwd/bwd + all-reduce + clip/AdamW) has no host sync, so it
    # can be captured once and replayed.  The captured buffers are static: `data` itself is the graph's input -- copy
    # each new batch of the same shapes INTO it (TrainStep.load_into(data, batch)) on the replaying stream before replay().
    # Capturing runs warm-up steps and then restores weights, optimiser state and the dropout-mask step count: it has no
    # side effect on the model or on the masks of the steps that follow.
    def capture(self, data, labels=None, warmup=3):
        body = lambda: self.step(data, labels)
        snap = self.snapshot()                                 # the warm-up steps are real steps: undone below, so that
        side = torch.cuda.Stream()                             # capturing leaves weights and optimiser state untouched
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                body()
        torch.cuda.current_stream().wait_stream(side)
        self.restore(snap)
        torch.cuda.synchronize()
        self._pin_workspaces()
        if self.graph is not None:
            self._retired.append(self.graph)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread polls events while the capture is open
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._g_loss, self._g_norm = body()
        return self.graph

    def replay(self):
        if self.model.training:
            self.nh.next_mask_step()                           # the captured mask draw reads its step number from the device
        self.graph.replay()
        return self._g_loss, self._g_norm

    # ---- software pipeline across steps, eager:   stream G: backbone + ROI pooling of batch k+1   (-> feats_next)
    #                                                stream H: head update of batch k                 (on feats_cur)
    def _streams(self, masked=False, chains=1):
        """The streams of the pipelined modes, all at the default priority (work on streams of different priorities has
        hung on this platform): (backbone stream, head stream), or ([backbone streams], head stream) for chains > 1.
        masked=True: disjoint CU sets -- `head_cus` CUs for the head's stream, the rest for the backbone's streams -- so
        the head's small dependent launches do not queue behind the backbone's chip-wide grids."""
        if chains > 1:
            key = "_chain_streams_m" if masked else "_chain_streams"
            have = getattr(self, key, None)
            if have is None or len(have[0]) < chains:
                sg0, sh = self._streams(masked)
                extra = [self._new_backbone_stream(masked) for _ in range(chains - 1)]
                setattr(self, key, ([sg0] + extra, sh))
            lst, sh = getattr(self, key)
            return lst[:chains], sh
        if not masked:
            if getattr(self, "_sg", None) is None:
                self._sg, self._sh = torch.cuda.Stream(), torch.cuda.Stream()
            return self._sg, self._sh
        if getattr(self, "_sgm", None) is None:
            import ctypes
            from ._lib import call
            total = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
            n_head = self.head_cus if self.head_cus > 0 else total // 8
            words = (total + 31) // 32
            hm, bm = (ctypes.c_uint32 * words)(), (ctypes.c_uint32 * words)()
            # mask bit i is CU i/8 of XCD i%8 on this part: the first n bits take n/8 CUs from every XCD (measured at C2:
            # 32 CUs taken from ONE XCD instead run 3x slower; 16 / 24 / 28 / 32 / 40 CUs: 7.8k / 8.7k / 8.8k / 9.0k / 8.1k clips/s)
            for cu in range(total):
                (hm if cu < n_head else bm)[cu // 32] |= 1 << (cu % 32)
            self._raw_streams, out = [], []
            self._backbone_mask = (bm, words)
            for m in (bm, hm):
                h = ctypes.c_void_p()
                call("eventad_stream_create_cu_mask", ctypes.cast(m, ctypes.c_void_p), words, ctypes.byref(h))
                self._raw_streams.append(h)
                out.append(torch.cuda.ExternalStream(h.value))
            self._sgm, self._shm = out
            import atexit
            atexit.register(self.close)                        # HIP streams made outside torch: destroyed before the runtime unloads
        return self._sgm, self._shm

    def _new_backbone_stream(self, masked):
        """One more stream for a further backbone chain (same CU set as the first backbone stream)."""
        if not masked:
            return torch.cuda.Stream()
        import ctypes
        from ._lib import call
        bm, words = self._backbone_mask
        h = ctypes.c_void_p()
        call("eventad_stream_create_cu_mask", ctypes.cast(bm, ctypes.c_void_p), words, ctypes.byref(h))
        self._raw_streams.append(h)
        return torch.cuda.ExternalStream(h.value)

    def caller_stream(self):
        """A non-blocking stream for the caller's own loop (batch preparation, replay_dual calls): work on the legacy default
        stream synchronises implicitly with the pipeline's CU-masked streams, which are blocking streams."""
        if getattr(self, "_caller_stream", None) is None:
            self._caller_stream = torch.cuda.Stream()
        return self._caller_stream

    def close(self):
        """Destroy the CU-masked streams (idempotent).  Graphs captured by this object must not be replayed afterwards."""
        raw, self._raw_streams = getattr(self, "_raw_streams", None) or [], []
        if raw:
            from ._lib import call
            torch.cuda.synchronize()
            self._sgm = self._shm = None
            self._chain_streams_m = None
            for h in raw:
                call("eventad_stream_destroy", h)

    def prime(self, data):
        """Fill the pipeline: features of the first batch."""
        with torch.no_grad():
            f = self.model.extract_features(data, frames=(1,))
        if getattr(self, "_feats_cur", None) is None or self._feats_cur.shape != f.shape:
            self._feats_cur = torch.empty_like(f)
        self._feats_cur.copy_(f)
        self._streams()

    def pipelined_step(self, data_cur, labels_cur=None, data_next=None):
        """Head update of `data_cur` (whose features prime() / the previous call produced) overlapped with the backbone
        of `data_next` (default: data_cur again).  Returns (loss, grad_norm) of data_cur."""
        data_next = data_next if data_next is not None else data_cur
        main = torch.cuda.current_stream()
        sg, sh = self._streams()
        sg.wait_stream(main); sh.wait_stream(main)
        with torch.no_grad():
            with torch.cuda.stream(sg):
                feats_next = self.model.extract_features(data_next, frames=(1,))
            with torch.cuda.stream(sh):
                loss, norm = self._head_update(self._feats_cur, data_cur, labels_cur)
        main.wait_stream(sg); main.wait_stream(sh)
        self._feats_cur.copy_(feats_next)
        return loss, norm

    # ---- the same pipeline as single-branch hipGraphs on streams coupled only by events, buffered per CHAIN c = k % n so
    # that nothing is copied between the stages:
    #     backbone graph c  reads its own input batch in_b[c], works in workspace set c, writes its own feature tensor
    #                       feats[c]                                                           (stream G[c])
    #     head graph c      reads feats[c] and the boxes / labels in_h[c]                      (stream H)
    # Per replay k (c = k % n, c' = (k+1) % n):
    #     stream H:     wait(backbone k done); head graph c; record(head k done)
    #     stream G[c']: wait(head k+1-n done: it was the last reader of feats[c'] and in_h[c']); in_b[c'], in_h[c'] <- batch
    #                   k+1 (if given); backbone graph c'; record(backbone k+1 done)
    # The head updates run strictly in order on one stream.  With n = 2 this is double buffering: the backbone of batch k+1
    # overlaps the head update of batch k.  With n = 3 -- and a caller that issues steps without a per-step join -- the
    # backbone passes of batches k+1 and k+2 are in flight together (different streams, different workspace sets) and fill
    # each other's launch gaps.  The step object owns the static inputs, so the head always sees the boxes and labels of
    # the batch that produced its features.
    def capture_dual(self, data, warmup=3, prefetch=1):
        """data: the first batch (formatted, on the device).  Captures the graphs on copies of it and starts the backbone
        pass of that batch.  Every later batch must have the same shapes (events N, boxes M, clips B).
        prefetch: how many batches the backbone runs ahead of the head (1 .. chains - 1; "max" = chains - 1).  With p > 1
        the first p - 1 calls of replay_dual only start backbone passes (and return None), and the loss a call returns
        belongs to the batch passed p calls earlier; the steady state is the same, but a caller that synchronises now and
        then (an epoch end, a benchmark's timer) restarts with p finished passes in hand instead of one."""
        d = self._dual = {}
        in_b = d["in_b"] = [data.clone()]
        in_h = d["in_h"] = [_HeadInputs(in_b[0])]
        self.prime(in_b[0])
        snap = self.snapshot()                                 # warm-up steps are undone: the first replay consumes `data`
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.pipelined_step(in_h[0], None, in_b[0])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._pin_workspaces()
        d["gb"], d["gh"], d["feats"], d["out"], d["mask_step"], d["masks"] = [], [], [], [], [], []
        native = getattr(self.model.dagr_model.backbone, "_native", None)

        def add_chain():
            c = len(d["gb"])
            if c > 0:
                in_b.append(data.clone())
                in_h.append(_HeadInputs(in_b[c]))
            with torch.no_grad():
                if native is not None:
                    native.select_workspace(c)
                    if c > 0:                                  # allocate workspace set c outside the capture
                        self.model.extract_features(in_b[c], frames=(1,))
                        torch.cuda.synchronize()
                gb, gh = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                # the step's dropout masks depend on nothing: they are drawn at the end of the chain's backbone graph, off the
                # head's chain of dependent launches
                masks = step_dev = None
                if self.model.training:
                    masks = torch.empty((2 * in_h[c].num_graphs * 32, 256), device=in_h[c].bbox.device)
                    step_dev = torch.zeros(1, dtype=torch.int64, device=masks.device)   # set before each replay (_issue_backbone)
                d["mask_step"].append(step_dev)
                d["masks"].append(masks)                       # the graphs hold raw pointers: the buffer must outlive this call
                # so do the head's input tables (track features, box coordinates, targets, validity: a function of the batch
                # and its ROI features, not of the weights): filled behind the backbone into a set of the chain's own, which
                # the head graph then reads in place of the shared workspace's
                hi = in_h[c]
                tabs = self.nh.new_tables(hi.num_graphs)
                W_, H_ = int(self.model.dagr_model.width), int(self.model.dagr_model.height)
                with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                    feats = self.model.extract_features(in_b[c], frames=(1,))
                    self.nh.fill_tables(tabs, feats, hi.bbox, hi.bbox_batch, hi.bbox[:, 4], hi.num_graphs, W_, H_)
                    if masks is not None:
                        self.nh.fill_masks(masks, step_dev)
                # the head's loss and gradient norm live in the ONE head workspace every chain shares: the update's own launch
                # also leaves them in two floats of the chain's, so a step's results stay readable until the chain's next turn
                res = torch.zeros(2, device=hi.bbox.device)
                self.nh.tables, self.nh.result_out = tabs, res
                try:
                    with torch.cuda.graph(gh, capture_error_mode="thread_local"):
                        self._head_update(feats, hi, masks=masks)
                finally:
                    self.nh.tables, self.nh.result_out = None, None
                out = (res[0], res[1])
                d.setdefault("res", []).append(res)
                d.setdefault("tabs", []).append(tabs)        # (the graphs hold raw pointers)
                if native is not None:
                    native.select_workspace(0)
            d["gb"].append(gb); d["gh"].append(gh); d["feats"].append(feats); d["out"].append(out)

        add_chain(); add_chain()
        d["masked"] = self.head_cus > 0
        if self.head_cus < 0:                                  # stand-alone times of the two graphs decide
            t = []
            for g_ in (d["gb"][0], d["gh"][0]):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g_.replay()
                a.record()
                for _ in range(3):
                    g_.replay()
                b.record()
                torch.cuda.synchronize()
                t.append(a.elapsed_time(b) / 3)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                # every rank must take the same decision (it fixes the number of chains): use the slowest rank's times
                tt = torch.tensor(t, device=in_h[0].bbox.device, dtype=torch.float32)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t = [float(v) for v in tt.cpu()]
            d["graph_ms"] = tuple(t)
            # C2: head 0.41 ms against 0.82 ms (ratio 0.5; the partition + three chains give 0.68 ms per step against 0.88
            # without); C3 / C4 / C5: ratios 0.16 / 0.02 / 0.03, where taking CUs from the backbone only costs
            d["masked"] = t[1] > 0.3 * t[0]
        n = self.backbone_chains
        if n <= 0:
            n = 3 if (d["masked"] and native is not None and getattr(data, "image_feats", None) is not None) else 2
        n = max(2, n)
        if n > 2 and not d["masked"]:
            # More than two chains without the CU partition (every stream on the whole chip): at 32 M events a run in this
            # mode once ended in a GPU memory fault.  A cause that fits was found and fixed since (the chains' mask buffers
            # were freed while their graphs still wrote them; see add_chain), but the mode has not been run again -- it has
            # no use where two chains already hide a small head -- so it stays refused at every size.
            raise RuntimeError("TrainStep: %d backbone chains need the CU-partitioned head (EVENTAD_HEAD_CUS > 0, or a head that "
                               "takes more than 0.3 of its backbone's time); without it at most 2 chains are allowed" % n)
        while len(d["gb"]) < n:
            add_chain()
        d["n"] = n
        d["in_b_id"], d["in_h_id"] = [0] * n, [0] * n          # which loaded batch each buffer holds
        d["ev_b"] = [torch.cuda.Event() for _ in range(n)]
        d["ev_h"] = [torch.cuda.Event() for _ in range(n)]
        d["k"], d["issued"], d["drained"] = 0, 0, False        # head updates done / backbone passes started
        p = n - 1 if prefetch == "max" else int(prefetch)
        if not 1 <= p <= n - 1:
            raise ValueError("capture_dual: prefetch must be between 1 and chains - 1 = %d" % (n - 1))
        d["prefetch"] = p
        self.restore(snap)                                     # warm-up and timing replays were real steps: undone here
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        sgs, sh = self._streams(d["masked"], n)
        for sg in sgs:
            sg.wait_stream(main)
        sh.wait_stream(main)
        with torch.cuda.stream(sgs[0]):                        # the first batch enters the pipeline
            self._replay_backbone(d, 0)
            d["ev_b"][0].record(sgs[0])
        d["issued"] = 1
        self._g_loss, self._g_norm = d["out"][0]
        return d["gb"], d["gh"]

    @staticmethod
    def _copy_pairs(pairs):
        """[(dst, src)] of equal shapes, on the current stream: every pair of contiguous same-dtype tensors on this device goes
        into ONE launch (eventad_copy_group, 16 per call) -- a batch is ~10 fields, and ten copy launches in front of the
        backbone graph hold the chain's stream longer than the data they move; anything else (host memory, a dtype change)
        takes torch's copy."""
        import ctypes
        fast, host = [], []
        for dst, src in pairs:
            if (src.is_cuda and dst.is_cuda and src.device == dst.device and src.dtype == dst.dtype and src.is_contiguous()
                    and dst.is_contiguous()):
                if dst.numel():
                    fast.append((dst, src))
            elif (not src.is_cuda and dst.is_cuda and src.dtype == dst.dtype and src.is_pinned() and dst.numel()
                  and (src.is_contiguous() and dst.is_contiguous() or (src.dim() == 4 and src.stride() == dst.stride()
                       and src.is_contiguous(memory_format=torch.channels_last) and dst.is_contiguous(memory_format=torch.channels_last)))):
                # pinned host memory: copied by the library on the current stream's raw handle -- not by torch, whose host
                # allocator would record events on this (possibly CU-masked, externally owned) stream and touch them again when
                # the buffer is freed, after the stream may be gone
                host.append((dst, src))
            else:
                dst.copy_(src, non_blocking=True)
        st = torch.cuda.current_stream().cuda_stream
        if host:
            n = len(host)
            srcs = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in host])
            dsts = (ctypes.c_void_p * n)(*[d_.data_ptr() for d_, _ in host])
            sizes = (ctypes.c_uint64 * n)(*[d_.numel() * d_.element_size() for d_, _ in host])
            call("eventad_copy_to_device", n, ctypes.cast(srcs, ctypes.c_void_p), ctypes.cast(dsts, ctypes.c_void_p),
                 ctypes.cast(sizes, ctypes.c_void_p), st)
        for i in range(0, len(fast), 16):
            part = fast[i: i + 16]
            n = len(part)
            srcs = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in part])
            dsts = (ctypes.c_void_p * n)(*[d_.data_ptr() for d_, _ in part])
            sizes = (ctypes.c_uint64 * n)(*[d_.numel() * d_.element_size() for d_, _ in part])
            call("eventad_copy_group", n, ctypes.cast(srcs, ctypes.c_void_p), ctypes.cast(dsts, ctypes.c_void_p),
                 ctypes.cast(sizes, ctypes.c_void_p), st)

    @staticmethod
    def _load_pairs(in_b, batch):
        pairs = []
        for k in ("pos", "x", "batch", "image", "bbox", "bbox_batch", "bbox0", "bbox0_batch"):
            src, dst = getattr(batch, k, None), getattr(in_b, k, None)
            if dst is None:
                continue
            if src is None or src.shape != dst.shape:
                raise ValueError("replay_dual: batch field %r has shape %s, the captured graphs need %s (one capture per shape bucket)"
                                 % (k, None if src is None else tuple(src.shape), tuple(dst.shape)))
            pairs.append((dst, src))
        if getattr(in_b, "image_feats", None) is not None:
            pairs.extend(zip(in_b.image_feats, batch.image_feats))
        return pairs

    @staticmethod
    def _any_on_device(batch):
        """Does any tensor that a step loads from this batch live on the GPU (i.e. may still be in the making on the
        caller's stream)?"""
        for k in ("pos", "x", "batch", "image", "bbox", "bbox_batch", "bbox0", "bbox0_batch"):
            v = getattr(batch, k, None)
            if isinstance(v, torch.Tensor) and v.is_cuda:
                return True
        return any(m.is_cuda for m in (getattr(batch, "image_feats", None) or ()))

    @classmethod
    def load_into(cls, in_b, batch):
        """Copy a batch into static (captured) input buffers of the same shapes, on the current stream."""
        cls._copy_pairs(cls._load_pairs(in_b, batch))

    def _issue_backbone(self, d, c, next_batch, sg):
        """Backbone-stream part of a step: load the next batch into chain c's buffers (if given; None = chain c runs the
        batch it holds again), backbone graph of chain c."""
        with torch.cuda.stream(sg):
            sg.wait_event(d["ev_h"][c])                        # the last reader of feats[c] / in_h[c] is done
            if next_batch is not None:
                # the head's copy of the boxes comes from the same batch in the same launch
                pairs = self._load_pairs(d["in_b"][c], next_batch)
                pairs += [(d["in_h"][c].bbox, next_batch.bbox), (d["in_h"][c].bbox_batch, next_batch.bbox_batch)]
                self._copy_pairs(pairs)
                d["in_b_id"][c] += 1
                d["in_h_id"][c] = d["in_b_id"][c]
            if d["in_h_id"][c] != d["in_b_id"][c]:
                d["in_h"][c].copy_from(d["in_b"][c])
                d["in_h_id"][c] = d["in_b_id"][c]
            self._replay_backbone(d, c)
            d["ev_b"][c].record(sg)

    def _replay_backbone(self, d, c):
        """Backbone graph of chain c on the current stream; it ends with the mask draw of the step it belongs to."""
        if d["mask_step"][c] is not None:
            self.nh.next_mask_step(d["mask_step"][c])
        d["gb"][c].replay()

    def replay_dual(self, next_batch=None, join=True):
        """One pipelined step: head update of the batch loaded by the previous call (or capture_dual) || backbone of
        `next_batch` (formatted; on the device or in pinned host memory; None = the chain whose turn it is runs the batch it
        holds again).  Returns (loss, grad_norm) of the batch the head consumed -- two floats owned by that chain, written at
        the end of its head graph: valid after a synchronisation and until the chain's next turn (n calls later).  join=False: no per-step barrier with the
        caller's stream (what lets more than one backbone pass be in flight); call flush() -- or synchronise -- before
        reading results or re-using the memory of a `next_batch` passed earlier."""
        d = self._dual
        n = d["n"]
        main = torch.cuda.current_stream()
        sgs, sh = self._streams(d["masked"], n)
        c, c_next = d["k"] % n, d["issued"] % n                # chain of the head update / of the backbone pass to start
        # A batch on the device was produced on the caller's stream: the chain waits for that stream.  (A batch in pinned
        # host memory is ready as it is.)  NOTE for loops that hand over batches: the CU-masked streams are blocking streams --
        # hipExtStreamCreateWithCUMask takes no flags -- so ANY operation on the legacy default stream, the event of this wait
        # included, is a barrier across all of them and serialises the pipeline (C2: 1.35 instead of 0.58 ms per step): run the
        # loop under a stream of its own (`with torch.cuda.stream(ts.caller_stream()):`).
        wait_main = join or (next_batch is not None and self._any_on_device(next_batch))
        if d["drained"] or d["issued"] - d["k"] < d["prefetch"]:
            # after flush(), or while the backbone is not yet `prefetch` batches ahead: only start this batch's backbone pass
            d["drained"] = False
            if wait_main:
                sgs[c_next].wait_stream(main)
            self._issue_backbone(d, c_next, next_batch, sgs[c_next])
            d["issued"] += 1
            if join:
                main.wait_stream(sgs[c_next])
            return None
        if wait_main:
            sgs[c_next].wait_stream(main)                      # next_batch was produced on the caller's stream
        if join:
            sh.wait_stream(main)
        with torch.cuda.stream(sh):
            sh.wait_event(d["ev_b"][c])
            d["gh"][c].replay()
            d["ev_h"][c].record(sh)
        self._issue_backbone(d, c_next, next_batch, sgs[c_next])
        d["k"] += 1
        d["issued"] += 1
        if join:
            main.wait_stream(sgs[c_next]); main.wait_stream(sh)
        self._g_loss, self._g_norm = d["out"][c]
        return d["out"][c]

    def flush(self):
        """Drain the pipeline: head updates of every batch whose backbone pass was started, no new pass.  -> (loss, grad_norm)
        of the last one.  The next replay_dual(batch) restarts the pipeline with `batch` (and returns None: no head update
        in that call)."""
        d = self._dual
        n = d["n"]
        main = torch.cuda.current_stream()
        sgs, sh = self._streams(d["masked"], n)
        sh.wait_stream(main)
        c = d["k"] % n
        while d["k"] < d["issued"]:
            c = d["k"] % n
            with torch.cuda.stream(sh):
                sh.wait_event(d["ev_b"][c])
                d["gh"][c].replay()
                d["ev_h"][c].record(sh)
            d["k"] += 1
        d["drained"] = True
        main.wait_stream(sh)
        for sg in sgs:
            main.wait_stream(sg)
        self._g_loss, self._g_norm = d["out"][c]
        return d["out"][c]

    def check(self):
        """Lazy, synchronising: raises if a step saw a non-finite gradient norm (the reference raises at
        utils/train.py:147-149; here that step's update was skipped on the device) or a data-dependent backbone error
        (event outside the sensor, unsupported pooling geometry)."""
        bad = self.nh.nonfinite_steps()
        if bad != 0:
            raise ValueError("non-finite gradient norm seen in %d training step(s); their updates were skipped" % bad)
        nb = getattr(self.model.dagr_model.backbone, "_native", None)
        if nb is not None and getattr(nb, "ws", None) is not None:
            nb.check_flags()
