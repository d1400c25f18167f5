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
        # capture from the two graphs' stand-alone times (a head that is short against its backbone gets no CUs of its
        # own: they would only be taken from the backbone); 0: never; n > 0: always n CUs.
        self.head_cus = int(os.environ.get("EVENTAD_HEAD_CUS", "-1"))

    # ------------------------------------------------------------------ pieces
    def _head_update(self, feats, head_in, labels=None):
        """Head forward + backward + gradient all-reduce + clip/AdamW on `feats`; -> (loss, grad_norm) device scalars."""
        if labels is None:
            labels = head_in.bbox[:, 4]
        loss, _, _, _ = self.model._head_native(feats, head_in, labels, lean=True)
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
        """Head weights + optimiser state (flat buffers, ~14 MB), e.g. to undo warm-up steps."""
        nh = self.nh
        return [t.clone() for t in (nh.flat, nh.exp_avg, nh.exp_avg_sq, nh.opt_state)]

    def restore(self, snap):
        nh = self.nh
        for dst, src in zip((nh.flat, nh.exp_avg, nh.exp_avg_sq, nh.opt_state), snap):
            dst.copy_(src)

    # -------------------------------------------------------------------- eager
    def step(self, data, labels=None):
        """data: formatted batch on the device.  Returns (loss, grad_norm) as device tensors."""
        with torch.no_grad():
            feats = self.model.extract_features(data, frames=(1,))      # frame 0 features are never read by the head (EventAD.py:216)
            return self._head_update(feats, data, labels)

    # ---- hipGraph: the whole step (backbone + ROI + head fwd/bwd + all-reduce + clip/AdamW) has no host sync, so it
    # can be captured once and replayed.  The captured buffers are static: `data` itself is the graph's input -- copy
    # each new batch of the same shapes INTO it (TrainStep.load_into(data, batch)) on the replaying stream before replay().
    # Capturing runs warm-up steps and then restores weights and optimiser state: it has no side effect on the model.
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
        self.graph.replay()
        return self._g_loss, self._g_norm

    # ---- software pipeline across steps, eager:   stream G: backbone + ROI pooling of batch k+1   (-> feats_next)
    #                                                stream H: head update of batch k                 (on feats_cur)
    def _streams(self, masked=False):
        """The two streams of the pipelined modes, both at the default priority (work on streams of different priorities
        has hung on this platform).  masked=True: the pair with disjoint CU sets -- `head_cus` CUs for the head's stream,
        the rest for the backbone's -- so the head's small dependent launches do not queue behind the backbone's
        chip-wide grids."""
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
            for m in (bm, hm):
                h = ctypes.c_void_p()
                call("eventad_stream_create_cu_mask", ctypes.cast(m, ctypes.c_void_p), words, ctypes.byref(h))
                self._raw_streams.append(h)
                out.append(torch.cuda.ExternalStream(h.value))
            self._sgm, self._shm = out
            import atexit
            atexit.register(self.close)                        # HIP streams made outside torch: destroyed before the runtime unloads
        return self._sgm, self._shm

    def close(self):
        """Destroy the CU-masked streams (idempotent).  Graphs captured by this object must not be replayed afterwards."""
        raw, self._raw_streams = getattr(self, "_raw_streams", None) or [], []
        if raw:
            from ._lib import call
            torch.cuda.synchronize()
            self._sgm = self._shm = None
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

    # ---- the same pipeline as single-branch hipGraphs on two streams coupled only by events, double-buffered by step
    # parity p = k % 2 so that nothing is copied between the stages and the backbone of batch k+1 starts the moment the
    # backbone of batch k ends:
    #     backbone graph p  writes its own feature tensor feats[p]            (stream G)
    #     head graph p      reads feats[p] and the boxes / labels in_h[p]     (stream H)
    # Per replay k (p = k % 2, p' = (k+1) % 2):
    #     stream H: wait(backbone k done); head graph p; record(head k done)
    #     stream G: wait(head k-1 done: it was the last reader of feats[p'] and in_h[p']); in_b, in_h[p'] <- batch k+1 (if
    #               given); backbone graph p'; record(backbone k+1 done)
    # The step object owns the static inputs: `in_b` (a whole batch: the backbone graphs' input) and `in_h[0..1]` (the
    # boxes of the batch whose features the head graph of that parity consumes), so the head always sees the boxes and
    # labels of the batch that produced its features.
    def capture_dual(self, data, warmup=3):
        """data: the first batch (formatted, on the device).  Captures the graphs on copies of it and starts the backbone
        pass of that batch.  Every later batch must have the same shapes (events N, boxes M, clips B)."""
        d = self._dual = {}
        in_b = d["in_b"] = data.clone()
        in_h = d["in_h"] = [_HeadInputs(in_b), _HeadInputs(in_b)]
        d["in_b_id"], d["in_h_id"] = 0, [0, 0]                 # which loaded batch each buffer holds
        self.prime(in_b)
        snap = self.snapshot()                                 # warm-up steps are undone: the first replay consumes `data`
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.pipelined_step(in_h[0], None, in_b)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._pin_workspaces()
        d["gb"], d["gh"], d["feats"], d["out"] = [], [], [], []
        with torch.no_grad():
            for p in range(2):
                gb, gh = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                    feats = self.model.extract_features(in_b, frames=(1,))
                with torch.cuda.graph(gh, capture_error_mode="thread_local"):
                    out = self._head_update(feats, in_h[p])
                d["gb"].append(gb); d["gh"].append(gh); d["feats"].append(feats); d["out"].append(out)
        d["ev_b"] = [torch.cuda.Event(), torch.cuda.Event()]
        d["ev_h"] = [torch.cuda.Event(), torch.cuda.Event()]
        d["k"], d["drained"] = 0, False
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
            d["graph_ms"] = tuple(t)
            d["masked"] = t[1] > 0.5 * t[0]
        self.restore(snap)                                     # warm-up and timing replays were real steps: undone here
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        sg, sh = self._streams(d["masked"])
        sg.wait_stream(main); sh.wait_stream(main)
        with torch.cuda.stream(sg):                            # the first batch enters the pipeline
            d["gb"][0].replay()
            d["ev_b"][0].record(sg)
        self._g_loss, self._g_norm = d["out"][0]
        return d["gb"], d["gh"]

    @staticmethod
    def load_into(in_b, batch):
        """Copy a batch into static (captured) input buffers of the same shapes, on the current stream."""
        for k in ("pos", "x", "batch", "image", "bbox", "bbox_batch", "bbox0", "bbox0_batch"):
            src, dst = getattr(batch, k, None), getattr(in_b, k, None)
            if dst is None:
                continue
            if src is None or src.shape != dst.shape:
                raise ValueError("replay_dual: batch field %r has shape %s, the captured graphs need %s (one capture per shape bucket)"
                                 % (k, None if src is None else tuple(src.shape), tuple(dst.shape)))
            dst.copy_(src, non_blocking=True)
        if getattr(in_b, "image_feats", None) is not None:
            for dst, src in zip(in_b.image_feats, batch.image_feats):
                dst.copy_(src, non_blocking=True)

    def _issue_backbone(self, d, p_next, next_batch, sg):
        """Stream G part of a step: load the next batch (if any), backbone graph of parity p_next."""
        with torch.cuda.stream(sg):
            sg.wait_event(d["ev_h"][p_next])                   # the last reader of feats[p_next] / in_h[p_next] is done
            if next_batch is not None:
                self.load_into(d["in_b"], next_batch)
                d["in_b_id"] += 1
            if d["in_h_id"][p_next] != d["in_b_id"]:
                d["in_h"][p_next].copy_from(d["in_b"])
                d["in_h_id"][p_next] = d["in_b_id"]
            d["gb"][p_next].replay()
            d["ev_b"][p_next].record(sg)

    def replay_dual(self, next_batch=None, join=True):
        """One pipelined step: head update of the batch loaded by the previous call (or capture_dual) || backbone of
        `next_batch` (formatted; on the device or in pinned host memory; None = run the resident batch again).
        Returns (loss, grad_norm) of the batch the head consumed -- static device tensors of that step parity, valid after
        a synchronisation and until the call after next.  join=False: no per-step barrier with the caller's stream; call
        flush() -- or synchronise -- before reading results or re-using the memory of a `next_batch` passed earlier."""
        d = self._dual
        main = torch.cuda.current_stream()
        sg, sh = self._streams(d["masked"])
        if join or next_batch is not None:
            sg.wait_stream(main)                               # next_batch was produced on the caller's stream
        if join:
            sh.wait_stream(main)
        if d["drained"]:                                       # after flush(): restart the pipeline with this batch
            d["drained"] = False
            self._issue_backbone(d, d["k"] % 2, next_batch, sg)
            if join:
                main.wait_stream(sg)
            return None
        p = d["k"] % 2
        with torch.cuda.stream(sh):
            sh.wait_event(d["ev_b"][p])
            d["gh"][p].replay()
            d["ev_h"][p].record(sh)
        self._issue_backbone(d, 1 - p, next_batch, sg)
        d["k"] += 1
        if join:
            main.wait_stream(sg); main.wait_stream(sh)
        self._g_loss, self._g_norm = d["out"][p]
        return d["out"][p]

    def flush(self):
        """Drain the pipeline: head update of the last loaded batch, no new backbone pass.  -> (loss, grad_norm).
        The next replay_dual(batch) restarts the pipeline with `batch` (and returns None: no head update in that call)."""
        d = self._dual
        main = torch.cuda.current_stream()
        sg, sh = self._streams(d["masked"])
        sh.wait_stream(main)
        p = d["k"] % 2
        with torch.cuda.stream(sh):
            sh.wait_event(d["ev_b"][p])
            d["gh"][p].replay()
            d["ev_h"][p].record(sh)
        d["k"] += 1
        d["drained"] = True
        main.wait_stream(sh); main.wait_stream(sg)
        self._g_loss, self._g_norm = d["out"][p]
        return d["out"][p]

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
