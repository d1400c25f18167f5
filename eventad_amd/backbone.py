"""Native (HIP) execution of the frozen DAGR backbone -- the product path behind
`dagr.model.networks.net.Net.forward` (reference src/dagr/model/networks/net.py:106-197).

Host logic only: weight preparation (LUT / concatenated kernel matrices / folded BatchNorm),
workspace ownership (torch tensors = device memory plumbing) and the sequence of C-ABI calls.
No host<->device synchronisation happens inside `forward`; node and edge counts of the pooled
graphs stay on the device (`n_dev`), so the whole forward is hipGraph-capturable."""
import os
import ctypes
import torch

from . import _lib, spline
from ._lib import call, ptr, stream_handle
from .geometry import Geometry
from .weights import LAYER_NAMES, LAYER_DIMS

FEATURE_CHANNELS = [16, 64, 64, 64, 64]


def _stream():
    return stream_handle()


def _rp(t):
    """Raw pointer of a (possibly strided-view) tensor; the C ABI gets the leading dimension separately."""
    import ctypes
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Level:
    """Node/edge arrays of one pooled graph level (padded to n_max; valid count in n_dev)."""
    pass


class NativeBackbone:
    def __init__(self, state_dict, width, height, device, radius=0.01, max_neighbors=16,
                 pooling_dim_at_output="5x7", time_window=1000000, prefix="backbone."):
        _lib.load()
        self.geom = Geometry(width, height, radius, max_neighbors, pooling_dim_at_output, time_window)
        self.device = torch.device(device)
        self.prefix = prefix
        self._ws_key = None
        self.ws = None
        self._front_zero_words = None
        self._slot, self._slots = 0, {}             # workspace sets (select_workspace): passes that overlap in time use different ones
        self.keep_superseded_workspaces = False     # set once a hipGraph holds raw pointers into the workspace: a workspace
        self._retired_ws = []                       # that has to grow is then retired (kept alive) instead of freed
        # edge-search kernel: -1 = thread per event, ELL row in registers (fastest on C2: 130 us); >0 = LDS-staged strips of
        # that many pixels (139 us at 64); 0 = batched prefetch (175 us).  All bit-identical (tools/microbench_ell.py).
        # edge-search kernel (see eventad_build_ell); None = by expected window occupancy: the row scan (1) for sparse
        # windows, the register-row spiral walk (0, early exit after 16 neighbours) for dense ones
        v = os.environ.get("EVENTAD_ELL_VARIANT")
        self.ell_variant = int(v) if v else None
        # who fills pool1's neighbour bitmap: True = the wave-per-voxel pooling kernel (no atomics; default),
        # "edge_search" = the edge search while the ELL row is in registers (atomics), False = a pass of its own over the ELL
        self.fused_mark = True
        # the last blend of a pooled Layer also scatters its rows into the next level's voxel table (one launch per level less)
        self.fused_blend_pool = os.environ.get("EVENTAD_FUSED_BLEND_POOL", "1") != "0"
        # the pooled levels' GEMMs compute only the kernel cells that the level's edges use (bit-identical outputs)
        self.tail_live_cells = os.environ.get("EVENTAD_TAIL_LIVE_CELLS", "1") != "0"
        self._tail_arena_clean = False
        # asynchronous (event-by-event) mode: rows reserved behind a dense pass for events appended later (append_events)
        self.async_capacity = 0
        self.async_graph = True                              # append_events replays its pooled stage from a recorded graph
        self._async_tail = None
        self._n_dense = self._n_cur = 0
        self.timing = None            # set to {} to record (start, end) HIP events around every C-ABI call
        with torch.cuda.device(self.device):
            call("eventad_spiral_prepare", self.geom.radius_px)      # spiral tables on this GPU, before any capture
        self.load_weights(state_dict)

    # ------------------------------------------------------------------ weights
    def load_weights(self, sd):
        g, dev, pre = self.geom, self.device, self.prefix
        sd = {k: v.detach().float().cpu() for k, v in sd.items() if k.startswith(pre)}
        w = {}
        # event-level Layer (conv_block1): LUT per spiral step, Cin padded 19 -> 20.  Every node's first ELL slot is its
        # self edge, whose LUT cell is spiral step 0: W_root x_i + LUT[0] x_i is one mat-vec with LUT[0] + W_root, appended
        # as LUT row `steps` (other edges from the same pixel keep LUT[0]).  The eval-mode BatchNorm that follows each conv
        # is an affine map per output channel: its scale is multiplied into the weights' output columns here, its shift is
        # added in the kernel's epilogue (conv 2: conv shift + skip shift), so a node's output is one accumulation.
        p1, p2 = pre + "conv_block1.conv_block1.", pre + "conv_block1.conv_block2."
        sc1, sh1 = spline.fold_bn(sd, p1 + "norm.module.")
        sc2, sh2 = spline.fold_bn(sd, p2 + "norm.module.")
        scs, shs = spline.fold_bn(sd, p2 + "norm_skip.module.")
        lut1 = spline.lut_by_spiral_step(sd[p1 + "conv.weight"], g, 20)
        root1 = spline.pad_rows(sd[p1 + "conv.lin.weight"].t().contiguous(), 20)
        lut2 = spline.lut_by_spiral_step(sd[p2 + "conv.weight"], g, 16)
        root2 = sd[p2 + "conv.lin.weight"].t().contiguous()
        w["lut1"] = spline.quad_row_order(torch.cat([lut1, (lut1[0] + root1).unsqueeze(0), torch.zeros(1, 20, 16)]) * sc1.view(1, 1, 16))
        w["lut2"] = spline.quad_row_order(torch.cat([lut2, (lut2[0] + root2).unsqueeze(0), torch.zeros(1, 16, 16)]) * sc2.view(1, 1, 16))
        w["shift1"] = sh1.contiguous()
        w["wskip"] = spline.quad_row_order(spline.pad_rows(sd[p2 + "lin.mlp.weight"].t().contiguous(), 20) * scs.view(1, 16))
        w["shift2"] = (sh2 + shs).contiguous()
        # pooled Layers: X.[W_0..W_24 | root | skip]  and  Y1.[W_0..W_24 | root]
        self.tail_w = []
        for name, (cin, cout) in list(zip(LAYER_NAMES, LAYER_DIMS))[1:]:
            q1, q2 = "%s%s.conv_block1." % (pre, name), "%s%s.conv_block2." % (pre, name)
            cin_pad = (cin + 3) // 4 * 4
            wc1 = torch.cat([sd[q1 + "conv.weight"].permute(1, 0, 2).reshape(cin, 25 * cout),
                             sd[q1 + "conv.lin.weight"].t(), sd[q2 + "lin.mlp.weight"].t()], dim=1)
            wc2 = torch.cat([sd[q2 + "conv.weight"].permute(1, 0, 2).reshape(cout, 25 * cout),
                             sd[q2 + "conv.lin.weight"].t()], dim=1)
            t = dict(cin=cin, cin_pad=cin_pad, wc1=spline.pad_rows(wc1, cin_pad).to(dev), wc2=wc2.contiguous().to(dev),
                     bn1=[v.to(dev) for v in spline.fold_bn(sd, q1 + "norm.module.")],
                     bn2=[v.to(dev) for v in spline.fold_bn(sd, q2 + "norm.module.")],
                     bns=[v.to(dev) for v in spline.fold_bn(sd, q2 + "norm_skip.module.")])
            self.tail_w.append(t)
        self.w = {k: ([t.to(dev) for t in v] if isinstance(v, tuple) else v.to(dev)) for k, v in w.items()}
        self.vox_tabs = [t.to(dev).contiguous() for t in g.pixel_voxel_tables()]     # px->cx, py->cy, vx_lo, vy_lo

    # ---------------------------------------------------------------- workspace
    def _alloc(self, N, B):
        if self._ws_key is not None and self._ws_key[1] == B and N <= self._ws_key[0]:
            return self.ws                                   # event-indexed arrays only need >= N rows: a loader's varying
                                                             # batch sizes re-use the workspace of the largest batch seen
        if self._ws_key is not None and self.keep_superseded_workspaces:
            self._retired_ws.append(self.ws)
        g, dev = self.geom, self.device
        P = B * g.H * g.W
        K = g.max_neighbors
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        # zero-initialised scratch of the event-level stage, back to back so eventad_sort_events clears it with ONE
        # fill: [counters(4) | pixel histogram(P+1) | pool1 neighbour bitmap(3*S0) | work counters of the two convs (2 x 256)]
        S0_ = g.grid[0][0] * g.grid[0][1] * (B + 1)
        work_at = (4 + P + 1 + 3 * S0_ + 31) // 32 * 32        # 128-byte aligned
        # the pooled stage's zero-initialised scratch (slot tables + neighbour bitmaps of pools 2..4, error flags) lies right
        # behind it, so that a front() pass clears both with the same fill; tail() alone (a hand-off set from elsewhere)
        # clears its part itself
        live_words = [(g.grid[lv][0] * g.grid[lv][1] * (B + 1) + 63) // 64 + 3 & ~3 for lv in range(4)]     # live-cell masks
        tail_words = sum(g.grid[lv][0] * g.grid[lv][1] * (B + 1) * (128 + 8 + 3) + 8 for lv in range(1, 4)) + sum(live_words) + 4
        tail_at = (work_at + 512 + 3) // 4 * 4
        self._front_zero_words = tail_at + tail_words - (4 + P + 1)
        both = torch.zeros(tail_at + tail_words, dtype=torch.int32, device=dev)
        front_arena = both[:work_at + 512]
        ws = dict(
            front_arena=front_arena, counters=front_arena[:4], pix_count=front_arena[4:4 + P + 1],
            bitmap0=front_arena[4 + P + 1:4 + P + 1 + 3 * S0_].view(S0_, 3), l1_work=front_arena[work_at:work_at + 512],
            slot_vd0=i32(S0_),                               # pool1: occupied << 31 | pooled in-edges per slot
            key=i32(N), tq=i32(N), bev=i32(N), pix_start=i32(P + 1),
            scan_tmp=i32(2 * (max(P, 1 << 16) // 2048 + 4)), sid_tmp=i32(N), sid=i32(N),
            key_s=i32(N), ts_s=i32(N), ev2=i32(N, 2), posn_s=f32(N, 3), x0s=f32(N, 20),
            ell_src=i32(N, K), ell_step=torch.empty((N, K), dtype=torch.int16, device=dev), deg=i32(N),
            y1=f32(N, 16), y2=f32(N, 16), node_slot0=i32(N), node_vox0=i32(N), ovf_list=i32(max(N, 1)),
        )
        nx0, ny0 = g.grid[0]
        S0 = nx0 * ny0 * (B + 1)
        from .synthetic import feature_map_sizes
        msz = feature_map_sizes(g.W, g.H)[2:]                        # maps 2..4 feed pools 2..4 (tail stage)
        layout = [("x", S0 * 84, True), ("npos", S0 * 3, True), ("npix", S0 * 2, False), ("nbatch", S0, False),
                  ("n_dev", 4, False), ("e_dev", 4, False), ("rowptr", S0 + 4, False), ("csr_src", S0 * 75, False),
                  ("csr_dxy", S0 * 75, False)] + [("map%d" % (k + 2), B * h * w * 64, True) for k, (h, w) in enumerate(msz)]
        total = sum((n + 3) // 4 * 4 for _, n, _ in layout)
        hands = []
        for _ in range(2):
            flat = torch.zeros(total, dtype=torch.int32, device=dev)
            H_ = Level(); H_.flat = flat; off = 0
            for name, n, is_f in layout:
                v = flat[off:off + n]
                setattr(H_, name, v.view(torch.float32) if is_f else v)
                off += (n + 3) // 4 * 4
            H_.x = H_.x.view(S0, 84); H_.npos = H_.npos.view(S0, 3); H_.npix = H_.npix.view(S0, 2)
            H_.n_dev = H_.n_dev[:1]; H_.e_dev = H_.e_dev[:1]
            H_.maps = [getattr(H_, "map%d" % (k + 2)).view(B, h, w, 64) for k, (h, w) in enumerate(msz)]
            hands.append(H_)
        ws["hand"] = hands
        ws["capacity"] = N
        if self.async_capacity > 0:                          # per-pixel lists of the events appended after a dense pass
            ws["pix_new_head"], ws["new_next"] = i32(P), i32(self.async_capacity)
        ws["scan_tmp_tail"] = i32(1 << 10)
        # zero-initialised scratch of the pooled stage (slot tables + neighbour bitmaps of pools 2..4, error flags): ONE
        # fill at the start of tail() instead of two memsets per level.  (Separate from the front arena: the two
        # stages may run concurrently on different batches.)
        tail_arena = both[tail_at:]
        ws["tail_arena"], ws["flags_tail"] = tail_arena, tail_arena[tail_words - 4:]
        toff = 0
        levels = []
        for lv in range(4):
            nx, ny = g.grid[lv]
            S = nx * ny * (B + 1)
            C = 80 if lv == 0 else 128
            ldx = C + 4                                      # [features | pos_x | pos_y | 0 0]
            L = Level()
            L.nx, L.ny, L.S, L.C, L.ldx, L.n_max = nx, ny, S, C, ldx, S
            # slot tables back to back (feat | sums | tsum | cnt | maxidx) so the scatter path clears them with ONE memset
            if lv == 0:
                tab = torch.empty(S * (C + 8), dtype=torch.int32, device=dev)     # pool_seg0 writes every slot: never cleared
            else:
                tab = tail_arena[toff: toff + S * (C + 8)]; toff += (S * (C + 8) + 3) // 4 * 4
            L.feat = tab[: S * C].view(S, C)
            L.sums = tab[S * C: S * C + 4 * S].view(torch.int64).view(S, 2)
            L.tsum = tab[S * C + 4 * S: S * C + 6 * S].view(torch.float64)
            L.cnt = tab[S * C + 6 * S: S * C + 7 * S]; L.maxidx = tab[S * C + 7 * S: S * C + 8 * S]
            L.occ = i32(S + 1); L.slot2node = i32(S)
            if lv == 0:
                L.bitmap = ws["bitmap0"]
            else:
                L.bitmap = tail_arena[toff: toff + 3 * S].view(S, 3); toff += (3 * S + 3) // 4 * 4
            L.live = tail_arena[toff: toff + live_words[lv]]; toff += live_words[lv]
            L.node_slot = i32(S)
            if lv == 0:                                      # graph of level 0 = the front stage's hand-off (set A)
                A_ = hands[0]
                L.x, L.npos, L.npix, L.nbatch, L.n_dev, L.e_dev = A_.x, A_.npos, A_.npix, A_.nbatch, A_.n_dev, A_.e_dev
                L.rowptr, L.csr_src, L.csr_dxy = A_.rowptr, A_.csr_src, A_.csr_dxy
            else:
                L.x = f32(S, ldx); L.npos = f32(S, 3); L.npix = i32(S, 2); L.nbatch = i32(S)
                L.n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
                L.e_dev = torch.zeros(1, dtype=torch.int32, device=dev)
                L.rowptr = i32(S + 2); L.csr_src = i32(S * 75); L.csr_dxy = i32(S * 75)
            L.z1 = f32(S, 27 * 64); L.z2 = f32(S, 26 * 64); L.y1 = f32(S, 64); L.y = f32(S, 64)
            L.node_slot_in = i32(S)                          # slot of each of this level's nodes in the NEXT pooling
            levels.append(L)
        ws["levels"] = levels
        self.ws, self._ws_key = ws, (N, B)
        return ws

    def select_workspace(self, slot):
        """Every following pass runs in workspace set `slot` (allocated on first use).  Two passes may overlap in time --
        on different streams, e.g. the backbone graphs of consecutive batches in TrainStep -- only if they were issued under
        different sets; weights are shared."""
        if slot == self._slot:
            return
        self._slots[self._slot] = (self.ws, self._ws_key, self._front_zero_words)
        self.ws, self._ws_key, self._front_zero_words = self._slots.get(slot, (None, None, None))
        self._slot = slot

    def _c(self, name, *args):
        if self.timing is None:
            return call(name, *args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        call(name, *args)
        b.record()
        self._seq[name] = self._seq.get(name, 0) + 1
        self.timing.setdefault("%s#%d" % (name, self._seq[name]), []).append((a, b))

    def timing_summary(self):
        """-> {call label: (n_launch_groups, mean ms)} from the recorded HIP events (synchronises)."""
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in (self.timing or {}).items()}

    # ------------------------------------------------------------------ forward
    def forward(self, pos, polarity, batch, num_graphs, image_feats, feats_are_nhwc=False):
        """pos f32[N,3] normalised (format_data output), polarity f32[N] or [N,1], batch i64[N],
        image_feats: the five CNN feature maps ([B,C,h,w], or [B,h,w,C] if feats_are_nhwc).
        Returns a dict with out3/out4 node arrays (padded) and their device-side counts."""
        self.front(pos, polarity, batch, num_graphs, image_feats, feats_are_nhwc)
        self._last_result = self.tail(self.ws["hand"][0])
        return self._last_result

    # The forward is two stages with a flat hand-off between them, so that a training loop can overlap
    # stage `front` of batch k+2, stage `tail` of batch k+1 and the head update of batch k (TrainStep).
    def front(self, pos, polarity, batch, num_graphs, image_feats, feats_are_nhwc=False):
        """Event-level stage: sort, edge search, Layer 1, pool1 -> level-0 graph in hand-off set A."""
        g, st = self.geom, _stream()
        self._seq = {}
        N, B = int(pos.shape[0]), int(num_graphs)
        self._pass_id = getattr(self, "_pass_id", 0) + 1
        if self.async_capacity > 0 and self.ws is not None and "pix_new_head" not in self.ws:
            self._ws_key = None                              # allocated before the asynchronous mode was switched on
        ws = self._alloc(N + self.async_capacity, B)
        self._B = B
        hand = ws["hand"][0]
        pos = pos.contiguous().float()
        polarity = polarity.reshape(-1).contiguous().float()
        batch = batch.contiguous().long()
        # The kernels read the CNN maps as NHWC.  Maps that already are -- the live trunk's output (feats_are_nhwc), or
        # [B,C,h,w] tensors in torch's channels_last memory format, e.g. precomputed features stored that way -- are used
        # in place; plain NCHW maps go through one grouped layout-change launch (maps 2..4 straight into the hand-off set).
        maps, todo = [], []
        in_place = True
        for i, m in enumerate(image_feats):
            m = m.float()
            if not feats_are_nhwc and m.dim() == 4 and m.is_contiguous(memory_format=torch.channels_last) and m.shape[1] > 1:
                maps.append(m.permute(0, 2, 3, 1))           # a contiguous NHWC view of the same memory
                continue
            m = m.contiguous()
            if feats_are_nhwc:
                maps.append(m)
                continue
            in_place = False
            dst = hand.maps[i - 2] if i >= 2 else None
            if dst is not None:
                assert tuple(dst.shape) == (m.shape[0], m.shape[2], m.shape[3], m.shape[1]), "unexpected CNN map size"
            o = dst if dst is not None else torch.empty((m.shape[0], m.shape[2], m.shape[3], m.shape[1]),
                                                        dtype=torch.float32, device=m.device)
            todo.append((m, o))
            maps.append(o)
        assert all(m.is_contiguous() for m in maps)
        # maps 2..4 feed the pooled stage: tail() on this pass's hand-off set reads them where they are
        self._tail_maps = maps[2:] if in_place else None
        self._front_maps, self._N = maps[:2], N             # graph_backward.BackboneBackward reads them
        if not in_place:
            for i in range(2, 5):                            # mixed input: make the hand-off set complete
                if maps[i].data_ptr() != hand.maps[i - 2].data_ptr():
                    hand.maps[i - 2].copy_(maps[i])
                    maps[i] = hand.maps[i - 2]
        if todo:                                            # every NCHW -> NHWC layout change of the step in one launch
            d = (ctypes.c_int64 * (6 * len(todo)))()
            for i, (m, o) in enumerate(todo):
                d[6 * i:6 * i + 6] = [m.data_ptr(), o.data_ptr(), m.shape[0], m.shape[1], m.shape[2], m.shape[3]]
            self._c("eventad_nchw_to_nhwc_group", len(todo), ctypes.cast(d, ctypes.c_void_p), st)
        assert [m.shape[3] for m in maps] == FEATURE_CHANNELS and all(m.shape[0] == B for m in maps)
        w, K = self.w, g.max_neighbors
        self._tail_arena_clean = True
        self._c("eventad_sort_events", ptr(pos), ptr(batch), ptr(polarity), ptr(maps[0]), maps[0].shape[1], maps[0].shape[2],
             N, B, g.W, g.H, g.T, ptr(ws["key"]), ptr(ws["tq"]), ptr(ws["bev"]), ptr(ws["pix_count"]), ptr(ws["pix_start"]),
             ptr(ws["scan_tmp"]), ptr(ws["sid_tmp"]), ptr(ws["ell_src"]), ptr(ws["sid"]), ptr(ws["counters"]),   # record scratch (32 B per event) = the first half of ell_src, which the edge search only writes afterwards
             ptr(ws["key_s"]), ptr(ws["ts_s"]), ptr(ws["ev2"]), ptr(ws["posn_s"]), ptr(ws["x0s"]), g.voxel[0][0], g.voxel[0][1],
             g.grid[0][0], g.grid[0][1], ptr(self.vox_tabs[0]), ptr(self.vox_tabs[1]), ptr(ws["node_slot0"]),
             ptr(ws["node_vox0"]), ptr(ws["ovf_list"]), self._front_zero_words, st)   # also clears counters, histogram, pool1 bitmap, work counters
        sw = self.ell_variant
        if sw is None:
            window = (2 * g.radius_px + 1) ** 2 * N / max(B * g.W * g.H, 1)         # mean events per search window
            sw = 1 if (g.radius_px in (1, 3) and window < 64) else 0
        self._c("eventad_build_ell", ptr(ws["sid"]), ptr(ws["ts_s"]), ptr(ws["ev2"]),
             ptr(ws["key_s"]), ptr(ws["pix_start"]), ptr(ws["counters"]),
             N, B, g.W, g.H, g.radius_px, g.delta_t_us, K, g.Q, ptr(ws["ell_src"]), ptr(ws["ell_step"]), ptr(ws["deg"]),
             ptr(ws["node_slot0"]), ptr(ws["node_vox0"]), g.grid[0][0], g.grid[0][1],
             ptr(ws["levels"][0].bitmap) if self.fused_mark == "edge_search" else None, sw, 1, st)
        self._c("eventad_l1_conv1", ptr(ws["x0s"]), ptr(ws["ell_src"]), ptr(ws["ell_step"]), ptr(ws["deg"]), ptr(w["lut1"]),
             ptr(w["shift1"]), ptr(ws["y1"]), ptr(ws["l1_work"][:256]), N, g.steps, K, st)
        self._c("eventad_l1_conv2", ptr(ws["y1"]), ptr(ws["x0s"]), ptr(ws["ell_src"]), ptr(ws["ell_step"]), ptr(ws["deg"]),
             ptr(w["lut2"]), ptr(w["wskip"]), ptr(w["shift2"]), ptr(ws["y2"]), ptr(ws["l1_work"][256:]), N, g.steps, K, st)
        L = ws["levels"][0]
        fmap = maps[1]
        # wave-per-voxel segmented max over the pixel CSR (no atomics, no table memsets)
        self._c("eventad_pool_seg0", ptr(ws["y2"]), ptr(fmap), fmap.shape[1], fmap.shape[2], ptr(ws["posn_s"]),
                ptr(ws["key_s"]), ptr(ws["sid"]), ptr(ws["node_slot0"]), ptr(ws["pix_start"]), ptr(self.vox_tabs[2]),
                ptr(self.vox_tabs[3]), ptr(ws["ovf_list"]), ptr(ws["counters"]), N, B, g.W, g.H, L.nx, L.ny,
                ptr(L.feat), ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx), ptr(ws["ell_src"]), ptr(ws["deg"]),
                ptr(ws["node_vox0"]), ptr(L.bitmap) if self.fused_mark is True else None,
                ptr(ws["slot_vd0"]) if self.fused_mark is True else None, st)
        small = L.S <= 131072                               # both scans of the level fit the single-workgroup kernel
        if not self.fused_mark:
            self._c("eventad_pool_edges_mark_ell", ptr(ws["ell_src"]), ptr(ws["deg"]), ptr(ws["node_slot0"]), N, K,
                 L.nx, L.ny, B, ptr(L.bitmap), ptr(ws["counters"]), st)
        if small:
            self._c("eventad_pool_finalize_edges", ptr(L.feat), ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx), ptr(ws["bev"]),
                 ptr(L.bitmap), L.nx, L.ny, B, L.C, 2, 1, g.W, g.H, g.wh_inv[0], g.wh_inv[1], ptr(L.occ), _rp(hand.x), L.ldx,
                 _rp(hand.npos), _rp(hand.npix), _rp(hand.nbatch), ptr(L.node_slot), ptr(L.slot2node), _rp(hand.n_dev),
                 _rp(hand.rowptr), _rp(hand.csr_src), _rp(hand.csr_dxy), _rp(hand.e_dev),
                 ptr(ws["slot_vd0"]) if self.fused_mark is True else None, st)
        else:
            self._c("eventad_pool_finalize", ptr(L.feat), ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx), ptr(ws["bev"]),
                 L.nx, L.ny, B, L.C, 2, 1, g.W, g.H, g.wh_inv[0], g.wh_inv[1], ptr(L.occ),
                 ptr(ws["scan_tmp"]), _rp(hand.x), L.ldx, _rp(hand.npos), _rp(hand.npix), _rp(hand.nbatch), ptr(L.node_slot),
                 ptr(L.slot2node), _rp(hand.n_dev), st)
            self._c("eventad_pool_edges_emit", ptr(L.bitmap), ptr(L.node_slot), ptr(L.slot2node), _rp(hand.npix), L.n_max,
                 _rp(hand.n_dev), L.nx, L.ny, _rp(hand.rowptr), ptr(ws["scan_tmp"]), _rp(hand.csr_src), _rp(hand.csr_dxy),
                 _rp(hand.e_dev), st)
        self._n_dense = self._n_cur = N
        if self.async_capacity > 0:
            self._c("eventad_incr_reset", ptr(ws["pix_new_head"]), B * g.H * g.W, st)
        return hand

    def append_events(self, pos, polarity, batch):
        """Asynchronous update (reference src/dagr/asynchronous/: make_model_asynchronous __init__.py:41-110, conv.py:96-190,
        max_pool.py:245-252, over the incremental graph builder ev_graph.py:45-103): k new events, newer than every event of
        their clip so far, are appended to the graph of the last dense pass -- their ELL rows, their layer-1 rows and the
        pool1 voxels they touch -- at O(k) cost; the pooled stage (<= 2240 nodes per clip) is then run on the updated level-0
        graph.  Same result dict as forward().  pos f32[k,3] normalised, polarity f32[k], batch i64[k]."""
        ws, g, st = self.ws, self.geom, _stream()
        if ws is None or "pix_new_head" not in ws or self._n_dense == 0:
            raise RuntimeError("append_events needs a dense pass made with async_capacity > 0")
        k, n_cur, K = int(pos.shape[0]), self._n_cur, g.max_neighbors
        if n_cur + k > ws["capacity"] or n_cur + k - self._n_dense > ws["new_next"].numel():
            raise RuntimeError("append_events: %d appended events exceed the reserved capacity (%d); run a dense pass"
                               % (n_cur + k - self._n_dense, ws["new_next"].numel()))
        self._seq = {}
        hand, L, w, B = ws["hand"][0], ws["levels"][0], self.w, self._B
        if k > 0:
            pos = pos.contiguous().float()
            polarity = polarity.reshape(-1).contiguous().float()
            batch = batch.contiguous().long()
            m0, m1 = self._front_maps
            self._c("eventad_incr_insert", ptr(pos), ptr(polarity), ptr(batch), k, n_cur, self._n_dense, ws["capacity"], B, g.W, g.H,
                    g.T, ptr(m0), m0.shape[1], m0.shape[2], g.voxel[0][0], g.voxel[0][1], g.grid[0][0], g.grid[0][1], ptr(ws["key_s"]),
                    ptr(ws["ts_s"]), ptr(ws["ev2"]), ptr(ws["sid"]), ptr(ws["bev"]), ptr(ws["posn_s"]), ptr(ws["x0s"]),
                    ptr(ws["node_slot0"]), ptr(ws["node_vox0"]), ptr(ws["pix_new_head"]), ptr(ws["new_next"]), ptr(ws["counters"]), st)
            self._c("eventad_incr_edges", ptr(ws["ev2"]), ptr(ws["key_s"]), ptr(ws["pix_start"]), ptr(ws["pix_new_head"]),
                    ptr(ws["new_next"]), ptr(ws["counters"]), n_cur, k, self._n_dense, g.W, g.H, g.radius_px, g.delta_t_us, K, g.Q,
                    ptr(ws["ell_src"]), ptr(ws["ell_step"]), ptr(ws["deg"]), st)
            self._c("eventad_l1_conv1_rows", ptr(ws["x0s"]), ptr(ws["ell_src"]), ptr(ws["ell_step"]), ptr(ws["deg"]), ptr(w["lut1"]),
                    ptr(w["shift1"]), ptr(ws["y1"]), ptr(ws["l1_work"][:256]), n_cur, n_cur + k, g.steps, K, st)
            self._c("eventad_l1_conv2_rows", ptr(ws["y1"]), ptr(ws["x0s"]), ptr(ws["ell_src"]), ptr(ws["ell_step"]), ptr(ws["deg"]),
                    ptr(w["lut2"]), ptr(w["wskip"]), ptr(w["shift2"]), ptr(ws["y2"]), ptr(ws["l1_work"][256:]), n_cur, n_cur + k,
                    g.steps, K, st)
            self._c("eventad_incr_pool0", ptr(ws["y2"]), ptr(m1), m1.shape[1], m1.shape[2], ptr(ws["posn_s"]), ptr(ws["key_s"]),
                    ptr(ws["node_slot0"]), ptr(ws["node_vox0"]), ptr(ws["ell_src"]), ptr(ws["deg"]), n_cur, k, g.W, g.H, ptr(L.feat),
                    ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx), ptr(L.bitmap), ptr(ws["counters"]), st)
            self._n_cur = self._N = n_cur + k
        # level-0 nodes, features and pooled edges from the updated slot table, then the pooled stage: some 25 small launches
        # whose arguments do not change between two dense passes -- recorded once per dense pass and replayed
        if not self.async_graph or self.timing is not None:
            self._async_tail = None
            self._finalize0_updated(hand)
            self._tail_arena_clean = False
            return self.tail(hand)
        if self._async_tail is None or self._async_tail[0] != self._pass_id:
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self._finalize0_updated(hand)
                self._tail_arena_clean = False
                out = self.tail(hand)
            self._async_tail = (self._pass_id, graph, out)
        self._async_tail[1].replay()
        return self._async_tail[2]

    def _finalize0_updated(self, hand):
        """Level-0 nodes and pooled edges again from pool1's slot table (the scans read counts and bitmap directly)."""
        ws, g, st, B = self.ws, self.geom, _stream(), self._B
        L = ws["levels"][0]
        if L.S <= 131072:
            self._c("eventad_pool_finalize_edges", ptr(L.feat), ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx), ptr(ws["bev"]),
                    ptr(L.bitmap), L.nx, L.ny, B, L.C, 2, 1, g.W, g.H, g.wh_inv[0], g.wh_inv[1], ptr(L.occ), _rp(hand.x), L.ldx,
                    _rp(hand.npos), _rp(hand.npix), _rp(hand.nbatch), ptr(L.node_slot), ptr(L.slot2node), _rp(hand.n_dev),
                    _rp(hand.rowptr), _rp(hand.csr_src), _rp(hand.csr_dxy), _rp(hand.e_dev), None, st)
        else:
            self._c("eventad_pool_finalize", ptr(L.feat), ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx), ptr(ws["bev"]),
                    L.nx, L.ny, B, L.C, 2, 1, g.W, g.H, g.wh_inv[0], g.wh_inv[1], ptr(L.occ), ptr(ws["scan_tmp"]), _rp(hand.x), L.ldx,
                    _rp(hand.npos), _rp(hand.npix), _rp(hand.nbatch), ptr(L.node_slot), ptr(L.slot2node), _rp(hand.n_dev), st)
            self._c("eventad_pool_edges_emit", ptr(L.bitmap), ptr(L.node_slot), ptr(L.slot2node), _rp(hand.npix), L.n_max,
                    _rp(hand.n_dev), L.nx, L.ny, _rp(hand.rowptr), ptr(ws["scan_tmp"]), _rp(hand.csr_src), _rp(hand.csr_dxy),
                    _rp(hand.e_dev), st)

    def tail(self, hand):
        """Pooled stage: Layers 2..5 and pools 2..4 on the level-0 graph found in `hand` (set A or B)."""
        g, st, ws = self.geom, _stream(), self.ws
        B = self._B
        levels = ws["levels"]
        L0 = levels[0]
        own = hand is ws["hand"][0] and getattr(self, "_tail_maps", None) is not None
        tail_maps = self._tail_maps if own else hand.maps       # NHWC inputs of this pass are read in place
        # level-0 graph arrays come from the hand-off; everything this stage writes for level 0 is its own
        G0 = Level()
        G0.__dict__.update(L0.__dict__)
        G0.x, G0.npos, G0.npix, G0.nbatch, G0.n_dev, G0.e_dev = hand.x, hand.npos, hand.npix, hand.nbatch, hand.n_dev, hand.e_dev
        G0.rowptr, G0.csr_src, G0.csr_dxy = hand.rowptr, hand.csr_src, hand.csr_dxy
        view = [G0] + levels[1:]
        self._tail_view, self._tail_maps_used = view, tail_maps     # what graph_backward.PooledBackward walks back through
        if self._tail_arena_clean:
            self._tail_arena_clean = False                   # front() cleared it together with its own scratch
        else:
            ws["tail_arena"].zero_()
        out3 = None
        fused = self.fused_blend_pool                        # a Layer's last blend also scatters into the next level's voxel table
        for lv in range(4):
            L = view[lv]
            if lv > 0:
                Lp = view[lv - 1]
                vsx, vsy = g.voxel[lv]
                mean = 1 if g.pool_aggr[lv] == "mean" else 0
                fmap = tail_maps[lv - 1]
                if fused and mean:
                    self._c("eventad_pool_mean_gather", _rp(Lp.y), 64, 64, _rp(fmap), fmap.shape[1], fmap.shape[2], 64, _rp(Lp.npos),
                            _rp(Lp.nbatch), ptr(Lp.node_slot_in), Lp.n_max, _rp(Lp.n_dev), g.W, g.H, L.nx, L.ny, B, ptr(L.feat), st)
                if not fused:
                  self._c("eventad_pool_scatter", _rp(Lp.y), 64, 64, _rp(fmap), fmap.shape[1], fmap.shape[2], 64,
                     _rp(Lp.npos), _rp(Lp.npix), _rp(Lp.nbatch), None, None, Lp.n_max, _rp(Lp.n_dev), g.W, g.H, vsx, vsy,
                     L.nx, L.ny, B, mean, ptr(L.feat), ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx),
                     ptr(Lp.node_slot_in), ptr(ws["flags_tail"]), 1, _rp(Lp.rowptr), _rp(Lp.csr_src), ptr(L.bitmap), st)   # + edge marks
                if L.S <= 131072:
                    self._c("eventad_pool_finalize_edges", ptr(L.feat), ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx),
                         _rp(Lp.nbatch), ptr(L.bitmap), L.nx, L.ny, B, L.C, mean, 0, g.W, g.H, g.wh_inv[0], g.wh_inv[1], ptr(L.occ),
                         ptr(L.x), L.ldx, ptr(L.npos), ptr(L.npix), ptr(L.nbatch), ptr(L.node_slot), ptr(L.slot2node), ptr(L.n_dev),
                         ptr(L.rowptr), ptr(L.csr_src), ptr(L.csr_dxy), ptr(L.e_dev), None, st)
                else:
                    self._c("eventad_pool_finalize", ptr(L.feat), ptr(L.sums), ptr(L.tsum), ptr(L.cnt), ptr(L.maxidx), _rp(Lp.nbatch),
                         L.nx, L.ny, B, L.C, mean, 0, g.W, g.H, g.wh_inv[0], g.wh_inv[1], ptr(L.occ),
                         ptr(ws["scan_tmp_tail"]), ptr(L.x), L.ldx, ptr(L.npos), ptr(L.npix), ptr(L.nbatch), ptr(L.node_slot),
                         ptr(L.slot2node), ptr(L.n_dev), st)
                    self._c("eventad_pool_edges_emit", ptr(L.bitmap), ptr(L.node_slot), ptr(L.slot2node), ptr(L.npix), L.n_max,
                         ptr(L.n_dev), L.nx, L.ny, ptr(L.rowptr), ptr(ws["scan_tmp_tail"]), ptr(L.csr_src), ptr(L.csr_dxy),
                         ptr(L.e_dev), st)
            # Layer lv+2 on this level's graph
            t = self.tail_w[lv]
            dvx, dvy = g.lut_div[lv + 1]
            # Z only where a blend of this level reads it; on the small levels the GEMMs are latency bound and the
            # launch that gathers the masks costs more than the skipped tiles
            live = self.tail_live_cells and L.n_max >= 4096
            if live:
                self._c("eventad_tail_live_cells", _rp(L.csr_src), _rp(L.csr_dxy), _rp(L.e_dev), L.csr_src.numel(), dvx, dvy,
                        ptr(L.live), (L.n_max + 63) // 64, st)
                self._c("eventad_tail_gemm", _rp(L.x), L.ldx, ptr(t["wc1"]), 27 * 64, ptr(L.z1), 27 * 64, L.n_max, _rp(L.n_dev),
                        27 * 64, t["cin_pad"], ptr(L.live), st)
            else:
                self._c("eventad_gemm_f32", _rp(L.x), L.ldx, ptr(t["wc1"]), 27 * 64, ptr(L.z1), 27 * 64, L.n_max, _rp(L.n_dev),
                        27 * 64, t["cin_pad"], None, None, None, None, 0, 0, st)
            self._c("eventad_tail_blend", ptr(L.z1), 27 * 64, 25, None, 0, 0, _rp(L.rowptr), _rp(L.csr_src), _rp(L.csr_dxy),
                 L.n_max, _rp(L.n_dev), dvx, dvy, ptr(t["bn1"][0]), ptr(t["bn1"][1]), None, None, ptr(L.y1), 64, st)
            if live:
                self._c("eventad_tail_gemm", ptr(L.y1), 64, ptr(t["wc2"]), 26 * 64, ptr(L.z2), 26 * 64, L.n_max, _rp(L.n_dev),
                        26 * 64, 64, ptr(L.live), st)
            else:
                self._c("eventad_gemm_f32", ptr(L.y1), 64, ptr(t["wc2"]), 26 * 64, ptr(L.z2), 26 * 64, L.n_max, _rp(L.n_dev),
                        26 * 64, 64, None, None, None, None, 0, 0, st)
            if fused and lv < 3:
                Ln = view[lv + 1]
                nvx, nvy = g.voxel[lv + 1]
                nmap = tail_maps[lv]
                self._c("eventad_tail_blend_pool", ptr(L.z2), 26 * 64, 25, ptr(L.z1), 27 * 64, 26, _rp(L.rowptr), _rp(L.csr_src),
                     _rp(L.csr_dxy), L.n_max, _rp(L.n_dev), dvx, dvy, ptr(t["bn2"][0]), ptr(t["bn2"][1]), ptr(t["bns"][0]),
                     ptr(t["bns"][1]), ptr(L.y), 64, _rp(nmap), nmap.shape[1], nmap.shape[2], _rp(L.npos), _rp(L.npix),
                     _rp(L.nbatch), g.W, g.H, nvx, nvy, Ln.nx, Ln.ny, B, 1 if g.pool_aggr[lv + 1] == "mean" else 0, ptr(Ln.feat),
                     ptr(Ln.sums), ptr(Ln.tsum), ptr(Ln.cnt), ptr(Ln.maxidx), ptr(L.node_slot_in), ptr(ws["flags_tail"]),
                     ptr(Ln.bitmap), st)
            else:
                self._c("eventad_tail_blend", ptr(L.z2), 26 * 64, 25, ptr(L.z1), 27 * 64, 26, _rp(L.rowptr), _rp(L.csr_src),
                     _rp(L.csr_dxy), L.n_max, _rp(L.n_dev), dvx, dvy, ptr(t["bn2"][0]), ptr(t["bn2"][1]), ptr(t["bns"][0]),
                     ptr(t["bns"][1]), ptr(L.y), 64, st)
            if lv == 2:
                out3 = dict(x=L.y, pos=L.npos, batch=L.nbatch, n_dev=L.n_dev, level=L)
        L = view[3]
        out4 = dict(x=L.y, pos=L.npos, batch=L.nbatch, n_dev=L.n_dev, level=L)
        return dict(out3=out3, out4=out4, ws=ws, flags=ws["counters"])

    def check_flags(self):
        """Lazy, synchronising check of the data-dependent error flags (fails loudly)."""
        f = int(self.ws["counters"][0].item()) | int(self.ws["flags_tail"][0].item())
        if f & 2:
            raise RuntimeError("eventad_amd: an event lies outside the sensor / batch range")
        if f & 4:
            raise RuntimeError("eventad_amd: pooled edge outside the 5x5x3 voxel window (unsupported geometry)")
        if f & 8:
            raise RuntimeError("eventad_amd: normalised event time outside [0,1]")
        if f & 16:
            raise RuntimeError("eventad_amd: asynchronous update: more than 16 appended events in one pixel (run a dense pass)")
        return f
