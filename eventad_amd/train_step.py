"""One optimisation step of EventAD, data-parallel over clips.

Reference step body (utils/train.py:100-152, setup_optimizer :27-53): zero_grad -> forward ->
backward (head only: the DAGR backbone is frozen and under no_grad) -> clip_grad_norm_(1.0) ->
NaN scan -> AdamW(lr 1e-3, wd 1e-5).  Here: no `detect_anomaly`, no per-parameter host syncs (a
non-finite gradient norm is recorded in a device flag and read lazily), and -- the one addition --
with world_size > 1 each rank runs the reference step on its own clips and the gradients of the
USED head parameters (852,066 floats = 3.41 MB; fusion_module.attention.* never gets a gradient)
are summed over ranks in one flat-bucket all-reduce (RCCL over xGMI) before clipping.  Loss is the
reference's SUM over objects, so summing gradients over ranks equals one large-batch reference step
with per-rank hidden-state chains."""
import os

import torch
import torch.distributed as dist


class TrainStep:
    def __init__(self, model, lr=1e-3, weight_decay=1e-5, grad_clip=1.0, process_group=None, fused=None):
        self.model = model
        self.grad_clip = grad_clip
        self.native = next(model.gru_net_event.parameters()).device.type == "cuda"
        if self.native:
            # product path: HIP head + fused clip/AdamW on flat buffers; ONE all-reduce on the flat gradient
            self.nh = model.native_head(lr=lr, weight_decay=weight_decay, grad_clip=grad_clip)
            self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
            if os.environ.get("EVENTAD_FORCE_ALLREDUCE") == "1" and dist.is_initialized():
                self.world = max(self.world, 2)        # test hook: exercise the collective (and its capture) on one rank
            self.group = process_group
            self.bucket = self.nh.grads
            self.bucket_bytes = 4 * self.nh.n
            return
        self.params = [p for n, p in model.named_parameters()
                       if p.requires_grad and not n.startswith("dagr_model.")]
        self.used = [p for n, p in model.named_parameters()
                     if p.requires_grad and not n.startswith("dagr_model.") and not n.startswith("fusion_module.attention.")]
        dev = self.used[0].device
        kw = {}
        if fused is None:
            fused = dev.type == "cuda"
        if fused:
            kw["fused"] = True
        # the reference hands every parameter to AdamW (utils/train.py:38-42); parameters without a
        # gradient are skipped by the optimiser, so only the used ones ever move
        self.opt = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay, **kw)
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.group = process_group
        n = sum(p.numel() for p in self.used)
        self.bucket = torch.zeros(n, dtype=torch.float32, device=dev)
        self.nonfinite = torch.zeros((), dtype=torch.int32, device=dev)
        self.bucket_bytes = 4 * n

    def _allreduce(self):
        if self.world == 1:
            return
        off = 0
        views = []
        for p in self.used:
            n = p.numel()
            v = self.bucket[off:off + n].view_as(p)
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
            views.append(v)
            off += n
        dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
        for p, v in zip(self.used, views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

    def backward_and_update(self, loss):
        loss.backward()
        self._allreduce()
        norm = torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
        self.nonfinite += (~torch.isfinite(norm)).to(torch.int32)
        self.opt.step()
        return norm

    def step(self, data, labels=None):
        """data: formatted batch on the device.  Returns (loss, grad_norm) as device tensors."""
        if labels is None:
            labels = data.bbox[:, 4]
        if self.native:
            with torch.no_grad():
                loss, logits, valid, targets = self.model.loss_only(data, labels, lean=True)
                self.nh.backward()
                if self.world > 1:
                    dist.all_reduce(self.nh.grads, op=dist.ReduceOp.SUM, group=self.group)
                norm = self.nh.optimizer_step()
            return loss, norm
        self.opt.zero_grad(set_to_none=True)
        loss, logits, valid, targets = self.model.loss_only(data, labels)
        norm = self.backward_and_update(loss)
        return loss.detach(), norm

    # ---- hipGraph: the whole step (backbone + ROI + head fwd/bwd + all-reduce + clip/AdamW) has no host
    # sync, so it can be captured once and replayed -- the launch-bound head chain then costs graph-node
    # boundaries (~1.5 us) instead of Python/ctypes launches.  The captured batch buffers are static:
    # copy a new batch of the same shape into `data` before replay().
    def capture(self, data, labels=None, warmup=3, pipelined=False, data_next=None):
        """Capture one step into a hipGraph.  pipelined: False = plain step, True/2 = two-stage pipeline
        (backbone || head), 3 = three-stage pipeline (front || tail+ROI || head)."""
        assert self.native, "graph capture is for the GPU path"
        if labels is None:
            labels = data.bbox[:, 4]
        if pipelined == 3:
            self.prime3(data)
            body = lambda: self.pipelined_step3(data, labels, data_next or data, data_next or data)
        elif pipelined:
            self.prime(data)
            body = lambda: self.pipelined_step(data, labels, data_next or data)
        else:
            body = lambda: self.step(data, labels)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if getattr(self, "graph", None) is not None:          # keep an earlier capture (and its private pool) alive:
            self._retired = getattr(self, "_retired", []) + [self.graph]   # buffers first allocated under it stay mapped
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread polls events while the capture is open
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._g_loss, self._g_norm = body()
        return self.graph

    def replay(self):
        self.graph.replay()
        return self._g_loss, self._g_norm

    # ---- software pipeline across steps.  The DAGR backbone is frozen (models/EventAD.py:148-150), so the
    # features of batch k+1 do not depend on the head update of batch k: one pipelined step runs
    #     stream G:  backbone + ROI pooling of batch k+1            (-> feats_next)
    #     stream H:  head forward/backward + all-reduce + clip/AdamW of batch k   (on feats_cur)
    # concurrently and then hands feats_next over.  Every batch still gets exactly one backbone pass and one
    # head update with the weights left by the previous batch -- results are identical to step(); the
    # launch-bound head chain and the backbone's small kernels simply share the 256 CUs.
    def prime(self, data):
        """Fill the pipeline: features of the first batch."""
        with torch.no_grad():
            f = self.model.extract_features(data, frames=(1,))
        if getattr(self, "_feats_cur", None) is None or self._feats_cur.shape != f.shape:
            self._feats_cur = torch.empty_like(f)
            # both streams at the default priority: work on streams of different priorities has hung on this platform
            self._sg, self._sh = torch.cuda.Stream(), torch.cuda.Stream()
        self._feats_cur.copy_(f)

    def pipelined_step(self, data_cur, labels_cur=None, data_next=None):
        """Head update of `data_cur` (whose features prime()/the previous call produced) overlapped with the
        backbone of `data_next`.  Returns (loss, grad_norm) of data_cur."""
        if labels_cur is None:
            labels_cur = data_cur.bbox[:, 4]
        data_next = data_next or data_cur
        main = torch.cuda.current_stream()
        sg, sh = self._sg, self._sh
        sg.wait_stream(main); sh.wait_stream(main)
        with torch.no_grad():
            with torch.cuda.stream(sg):
                feats_next = self.model.extract_features(data_next, frames=(1,))
            with torch.cuda.stream(sh):
                loss, logits, valid, targets = self.model._head_native(self._feats_cur, data_cur, labels_cur, lean=True)
                self.nh.backward()
                if self.world > 1:
                    dist.all_reduce(self.nh.grads, op=dist.ReduceOp.SUM, group=self.group)
                norm = self.nh.optimizer_step()
        main.wait_stream(sg); main.wait_stream(sh)
        self._feats_cur.copy_(feats_next)
        return loss, norm

    # ---- the same two-stage pipeline as TWO single-branch hipGraphs replayed on two streams.  A single graph with
    # parallel branches costs ~1.3 ms of HOST time per replay on this runtime (single-branch graphs enqueue in
    # ~0.04 ms), but measured end to end the two-graph form is no faster (C2: 1.50 vs 1.43 ms per step): with two
    # queues of small dependent kernels the step is bound by the command processor's per-dispatch latency
    # (~4.5 us), not by the host.  Kept as an option.  Per step:   stream H: wait(backbone k done); feats_cur <- feats_next; head graph(k)
    #                                   stream G: wait(copy done);       backbone graph(k+1)
    def capture_dual(self, data, labels=None, warmup=3, data_next=None):
        assert self.native, "graph capture is for the GPU path"
        if labels is None:
            labels = data.bbox[:, 4]
        data_next = data_next or data
        self.prime(data)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.pipelined_step(data, labels, data_next)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._retired = getattr(self, "_retired", []) + [g for g in (getattr(self, "_gb", None), getattr(self, "_gh", None)) if g is not None]
        self._gb, self._gh = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.no_grad():
            with torch.cuda.graph(self._gb, capture_error_mode="thread_local"):
                self._feats_next = self.model.extract_features(data_next, frames=(1,))
            with torch.cuda.graph(self._gh, capture_error_mode="thread_local"):
                loss, logits, valid, targets = self.model._head_native(self._feats_cur, data, labels, lean=True)
                self.nh.backward()
                if self.world > 1:
                    dist.all_reduce(self.nh.grads, op=dist.ReduceOp.SUM, group=self.group)
                norm = self.nh.optimizer_step()
        self._g_loss, self._g_norm = loss, norm
        self._ev_b, self._ev_c = torch.cuda.Event(), torch.cuda.Event()
        main = torch.cuda.current_stream()
        self._sg.wait_stream(main)
        with torch.cuda.stream(self._sg):                      # features of the first batch, as prime() left them
            self._gb.replay()
            self._ev_b.record(self._sg)
        self.graph = None
        return self._gb, self._gh

    def replay_dual(self, join=True):
        """join=False: the two streams are coupled only by the feature hand-over events (no per-step barrier with the
        caller's stream); call once with join=True -- or synchronise -- before reading the results."""
        main = torch.cuda.current_stream()
        sg, sh = self._sg, self._sh
        if join:
            sg.wait_stream(main); sh.wait_stream(main)
        with torch.cuda.stream(sh):
            sh.wait_event(self._ev_b)
            self._feats_cur.copy_(self._feats_next)
            self._ev_c.record(sh)
            self._gh.replay()
        with torch.cuda.stream(sg):
            sg.wait_event(self._ev_c)
            self._gb.replay()
            self._ev_b.record(sg)
        if join:
            main.wait_stream(sg); main.wait_stream(sh)
        return self._g_loss, self._g_norm

    # ---- three-stage pipeline: front(batch k+2) || tail + ROI(batch k+1) || head update(batch k).
    # The backbone itself is split at its flat hand-off (eventad_amd/backbone.py): the event-level stage and
    # the pooled stage are both chains of small dependent launches, so overlapping them hides launch latency.
    def prime3(self, data):
        net = self.model.dagr_model.backbone
        with torch.no_grad():
            net.front(data)
            hands = net._native.ws["hand"]
            hands[1].flat.copy_(hands[0].flat)
            out4 = net.tail(hands[1])[0][1]
            f = self.model.roi_features(out4, data, frames=(1,))
        if getattr(self, "_feats_cur", None) is None or self._feats_cur.shape != f.shape:
            self._feats_cur = torch.empty_like(f)
        self._feats_cur.copy_(f)
        if getattr(self, "_s3", None) is None:
            self._s3 = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())

    def pipelined_step3(self, data_cur, labels_cur=None, data_mid=None, data_next=None):
        """One step of the 3-stage pipeline; returns (loss, grad_norm) of data_cur."""
        if labels_cur is None:
            labels_cur = data_cur.bbox[:, 4]
        data_mid, data_next = data_mid or data_cur, data_next or data_cur
        net = self.model.dagr_model.backbone
        hands = net._native.ws["hand"]
        main = torch.cuda.current_stream()
        sa, sb, sh = self._s3
        sa.wait_stream(main); sb.wait_stream(main); sh.wait_stream(main)
        with torch.no_grad():
            with torch.cuda.stream(sa):
                net.front(data_next)
            with torch.cuda.stream(sb):
                out4 = net.tail(hands[1])[0][1]
                feats_next = self.model.roi_features(out4, data_mid, frames=(1,))
            with torch.cuda.stream(sh):
                loss, logits, valid, targets = self.model._head_native(self._feats_cur, data_cur, labels_cur, lean=True)
                self.nh.backward()
                if self.world > 1:
                    dist.all_reduce(self.nh.grads, op=dist.ReduceOp.SUM, group=self.group)
                norm = self.nh.optimizer_step()
        main.wait_stream(sa); main.wait_stream(sb); main.wait_stream(sh)
        hands[1].flat.copy_(hands[0].flat)
        self._feats_cur.copy_(feats_next)
        return loss, norm

    # ---- the three-stage pipeline as THREE single-branch graphs on three event-coupled streams:
    #     stream A: front(k+2)                      -> hand-off set 0
    #     stream B: hand 1 <- hand 0; tail + ROI(k+1) -> feats_next
    #     stream H: feats_cur <- feats_next; head update(k)
    def capture_triple(self, data, labels=None, warmup=2):
        assert self.native, "graph capture is for the GPU path"
        if labels is None:
            labels = data.bbox[:, 4]
        net = self.model.dagr_model.backbone
        self.prime3(data)
        self._s3 = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.pipelined_step3(data, labels, data, data)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        hands = net._native.ws["hand"]
        self._retired = getattr(self, "_retired", []) + [g for g in getattr(self, "_g3", ()) if g is not None]
        ga, gb, gh = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.no_grad():
            with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                net.front(data)
            with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                out4 = net.tail(hands[1])[0][1]
                self._feats_next = self.model.roi_features(out4, data, frames=(1,))
            with torch.cuda.graph(gh, capture_error_mode="thread_local"):
                loss, logits, valid, targets = self.model._head_native(self._feats_cur, data, labels, lean=True)
                self.nh.backward()
                if self.world > 1:
                    dist.all_reduce(self.nh.grads, op=dist.ReduceOp.SUM, group=self.group)
                norm = self.nh.optimizer_step()
        self._g3, self._g_loss, self._g_norm = (ga, gb, gh), loss, norm
        self._ev3 = {k: torch.cuda.Event() for k in ("a", "h", "b", "c")}
        sa, sb, sh = self._s3
        main = torch.cuda.current_stream()
        for s_ in (sa, sb, sh):
            s_.wait_stream(main)
        # state of a primed pipeline: hand 0 = front(batch), feats_next = tail(hand 1) of the batch before it
        with torch.cuda.stream(sa):
            ga.replay(); self._ev3["a"].record(sa)
        with torch.cuda.stream(sb):
            gb.replay(); self._ev3["b"].record(sb)
        with torch.cuda.stream(sh):
            self._ev3["c"].record(sh)
        self.graph = None
        return self._g3

    def replay_triple(self, join=False):
        ga, gb, gh = self._g3
        sa, sb, sh = self._s3
        ev = self._ev3
        net = self.model.dagr_model.backbone
        hands = net._native.ws["hand"]
        main = torch.cuda.current_stream()
        if join:
            for s_ in (sa, sb, sh):
                s_.wait_stream(main)
        with torch.cuda.stream(sh):                            # head(k): features of tail(k), handed over first
            sh.wait_event(ev["b"])
            self._feats_cur.copy_(self._feats_next)
            ev["c"].record(sh)
            gh.replay()
        with torch.cuda.stream(sb):                            # tail(k+1): level-0 graph of front(k+1), handed over first
            sb.wait_event(ev["a"]); sb.wait_event(ev["c"])
            hands[1].flat.copy_(hands[0].flat)
            ev["h"].record(sb)
            gb.replay()
            ev["b"].record(sb)
        with torch.cuda.stream(sa):                            # front(k+2)
            sa.wait_event(ev["h"])
            ga.replay()
            ev["a"].record(sa)
        if join:
            for s_ in (sa, sb, sh):
                main.wait_stream(s_)
        return self._g_loss, self._g_norm

    def step_from_feats(self, feats, data, labels=None):
        if labels is None:
            labels = data.bbox[:, 4]
        self.opt.zero_grad(set_to_none=True)
        loss, logits, valid, targets = self.model.head(feats, data, labels)
        norm = self.backward_and_update(loss)
        return loss.detach(), norm

    def check(self):
        if self.native:
            bad = self.nh.nonfinite_steps()
        else:
            bad = int(self.nonfinite.item())
        if bad != 0:
            raise ValueError("non-finite gradient norm seen in a training step (reference raises at utils/train.py:147-149)")
