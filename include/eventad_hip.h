/*
 * include/eventad_hip.h -- C ABI of libeventad_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the EventAD hot path.  The reference has no C plugin ABI:
 * its only native piece is the pybind11 torch extension `ev_graph_cuda`
 * (reference src/dagr/graph/ev_graph.cu:279-283).  Group (A) below are exactly those three
 * functions with torch tensors replaced by raw device pointers + sizes + a stream; groups (B)-(E)
 * are the native stage functions the Python host mirror (eventad_amd/) drives for the rest of the
 * path, each citing the reference code it replaces.
 *
 * Conventions: every pointer is DEVICE memory owned by the caller unless it says "host"; nothing
 * is allocated or freed per call (one small cached spiral table excepted); every launch goes on
 * `stream` (a hipStream_t passed as void*); no call synchronises the stream, so all of them can be
 * captured into a hipGraph after one warm-up call.  Return value: 0 = launched, -1 = bad argument
 * (the reference raises RuntimeError via AT_ASSERTM, ev_graph.cu:9-12), other = hipError_t.
 * Data-dependent problems (timestamps not monotone, event outside the sensor, pooled edge outside
 * the neighbour window) are reported through a caller-owned `flags` word, see EVENTAD_FLAG_*.
 */
#ifndef EVENTAD_HIP_H
#define EVENTAD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define EVENTAD_FLAG_T_NOT_MONOTONE 1u
#define EVENTAD_FLAG_POS_OUT_OF_FOV 2u
#define EVENTAD_FLAG_POOL_WINDOW    4u
#define EVENTAD_FLAG_T_OUT_OF_RANGE 8u
/* return code of the *_static entry points when the input exceeds the library's fixed workspace */
#define EVENTAD_ERR_TOO_LARGE (-3)

/* ---- (A) reference-compatible graph builder: replaces pybind module ev_graph_cuda ------------- */
/* insert_in_queue_cuda(indices, unique_coords, cumsum_counts, queue)   ev_graph.cu:241-276 (kernel :169-212)
 * indices i32[N] ids sorted by pixel, unique_coords i32[K], cumsum_counts i32[K], queue i32[B,Q,H,W] in place. */
int eventad_insert_in_queue(const int32_t* indices, const int32_t* unique_coords, const int32_t* cumsum_counts,
                            int32_t* queue, int B, int Q, int H, int W, int K, void* stream);
/* insert_in_queue_single_cuda(indices, events, queue)                  ev_graph.cu:215-238 (kernel :130-166) */
int eventad_insert_in_queue_single(const int32_t* indices, const int32_t* events, int32_t* queue,
                                   int B, int Q, int H, int W, void* stream);
/* fill_edges_cuda(batch, pos, all_timestamps, event_queue, indices, max_num_neighbors, radius, delta_t_us,
 *                 edges, min_index)                                     ev_graph.cu:82-128 (kernel :15-80)
 * edges i64[2,K] prefilled with -1 by the caller (ev_graph.py:89), K >= N*max_num_neighbors. */
int eventad_fill_edges(const int32_t* batch, const int32_t* pos, const int32_t* all_timestamps,
                       const int32_t* event_queue, const int32_t* indices, int max_num_neighbors, float radius,
                       float delta_t_us, int64_t* edges, int min_index, int B, int Q, int H, int W, int N, int64_t K,
                       void* stream);
/* spiral.h:1-15 as a table; dxdy is HOST memory int16[(2r+1)^2][2]; returns the number of steps. */
int eventad_spiral_offsets(int radius, int16_t* dxdy);
/* Put the spiral tables for `radius` on the current GPU (constant-memory tables of radii 0..13 once per GPU, a device
 * table per larger radius).  No counterpart in the reference (spiral.h computes the walk inline).  The edge-search entry
 * points do this themselves on first use -- with a synchronous copy, which is not allowed inside a stream capture: call
 * this once per GPU and radius before capturing.  Tables are never changed or freed afterwards. */
int eventad_spiral_prepare(int radius);

/* ---- (A') format_data for the events of a loader batch (src/dagr/utils/buffers.py:33-44) in one launch:
 * pos_out f32[N,3] = (pos_x / W, pos_y / H, t / T) with the int -> fp32 conversion and IEEE division torch applies,
 * x_out f32[N] = float(x).  Inputs as collated by the loader: pos i16[N,2], t i32[N], x i8[N]; width / height /
 * time_window are the batch's i64 DEVICE tensors (element 0 is used, as the reference does): no host sync. */
int eventad_format_events(const int16_t* pos, const int32_t* t, const int8_t* x, int N, const int64_t* width,
                          const int64_t* height, const int64_t* time_window, float* pos_out, float* x_out, void* stream);

/* Loading the next batch into the static input buffers of a captured step (the reference's loop hands each batch to the
 * model as fresh tensors, utils/train.py:120-131; a hipGraph reads fixed addresses instead): n <= 16 device-to-device
 * copies src[i] -> dst[i] of bytes[i] bytes in ONE launch.  src / dst / bytes are HOST arrays (read before the call
 * returns); segments must not overlap each other's destinations. */
int eventad_copy_group(int n, const void* const* src, void* const* dst, const uint64_t* bytes, void* stream);
/* The same hand-over for a batch in PINNED host memory (a loader's collated batch, utils/train.py:120-122 does `data.to(device)`):
 * n asynchronous host-to-device copies on `stream`.  The host buffers stay alive and unchanged until the stream has passed
 * the copies. */
int eventad_copy_to_device(int n, const void* const* src_host, void* const* dst, const uint64_t* bytes, void* stream);

/* ---- (B) native graph build: replaces EV_TGN.forward + SlidingWindowGraph.forward ------------- */
/* (ev_tgn.py:39-58, ev_graph.py:62-103, graph/utils.py:6-23) for one fresh batch.
 * eventad_sort_events: denormalise (ev_tgn.py:11-16), counting-sort the events by pixel, and emit
 * sorted-order records plus the 19(+1 pad)-channel layer-1 input row x0s[r] =
 * [polarity, 16 bilinear samples of CNN map 0 (net.py:116,200-228), pos_x, pos_y, 0].
 *   in : pos f32[N,3] normalised, batch i64[N], polarity f32[N], map0 f32[B,h,w,16] (NHWC)
 *   out: key/tq/bev i32[N] (event order: pixel key, t in us, clip), pix_start u32[B*H*W+1],
 *        sid i32[N] (event id at sorted position), key_s/ts_s i32[N], ev2 i32[N,2] = {id, t} pairs (nullable), posn_s f32[N,3], x0s f32[N,20]
 *   ws : pix_count u32[B*H*W+1], scan_tmp u32[2*(B*H*W/2048+2)], sid_tmp i32[N], rec (nullable, 16-byte aligned)
 *        i32[N,8]: with it every event is placed as a 32-byte record {id, key, t, polarity | x, y, t, 0} and the gather
 *        pass reads records instead of chasing event ids (same results; sid_tmp is then unused),
 *        counters u32[4], zeroed by this call -- the first of a step ([0] = EVENTAD_FLAG_* bits, accumulated by the
 *        later stages; [1] unused; [2] = length of ovf_list; [3] = longest pixel segment of the batch).
 *        When pix_count == counters + 4 the two are cleared by ONE memset, which then also covers the
 *        zero_tail_words u32 words behind pix_count (the caller's other zero-initialised scratch of the step,
 *        e.g. the pool1 bitmap); zero_tail_words must be 0 otherwise.
 *   pool1 prep (node_slot nullable = skip): node_slot i32[N] = voxel slot of each sorted event computed from its
 *        float position as grid_cluster does (pooling.py:55-56); px_to_cx i32[W] / py_to_cy i32[H] = voxel of a
 *        pixel; node_vox i32[N] = the slot decomposed as cx | cy<<8 | plane<<16; events whose slot differs from their pixel's (t == time_window, off-grid positions) -> ovf_list */
int eventad_sort_events(const float* pos, const int64_t* batch, const float* polarity, const float* map0_nhwc,
                        int map_h, int map_w, int N, int B, int W, int H, int T, int32_t* key, int32_t* tq,
                        int32_t* bev, uint32_t* pix_count, uint32_t* pix_start, uint32_t* scan_tmp, int32_t* sid_tmp,
                        int32_t* rec, int32_t* sid, uint32_t* counters, int32_t* key_s, int32_t* ts_s, int32_t* ev2,
                        float* posn_s, float* x0s, float vsx, float vsy, int nx, int ny, const int32_t* px_to_cx,
                        const int32_t* py_to_cy, int32_t* node_slot, int32_t* node_vox, int32_t* ovf_list, int zero_tail_words,
                        void* stream);
/* eventad_sort_events without workspace arguments, for callers that cannot allocate device memory (code under hipGraph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same inputs,
 * outputs and results as eventad_sort_events; pix_count / scan_tmp / sid_tmp live in a fixed 8.1 MiB array inside the
 * library (one copy per GPU, held until the process ends), so B*H*W <= 1048576 pixels and N <= 1048576 events:
 * anything larger returns EVENTAD_ERR_TOO_LARGE (and says so on stderr).  Allocates nothing.  counters u32[4] stays
 * caller-owned (the later stages read it).  All calls on one GPU share the array: calls that overlap on one GPU race. */
int eventad_sort_events_static(const float* pos, const int64_t* batch, const float* polarity, const float* map0_nhwc,
                               int map_h, int map_w, int N, int B, int W, int H, int T, int32_t* key, int32_t* tq,
                               int32_t* bev, uint32_t* pix_start, int32_t* sid, uint32_t* counters, int32_t* key_s,
                               int32_t* ts_s, int32_t* ev2, float* posn_s, float* x0s, float vsx, float vsy, int nx, int ny,
                               const int32_t* px_to_cx, const int32_t* py_to_cy, int32_t* node_slot, int32_t* node_vox,
                               int32_t* ovf_list, void* stream);
/* eventad_build_ell: the edge search of ev_graph.cu:15-80 over the pixel CSR.  ELL adjacency in
 * pixel-sorted numbering: ell_src i32[N,K] (slot 0 = self edge), ell_step u16[N,K] (spiral step =
 * LUT cell of the source pixel), deg i32[N].  bitmap (nullable): also set, for every edge,
 * the source-voxel bit in the destination voxel's pool1 neighbour bitmap u32[S,3] (= eventad_pool_edges_mark_ell).
 * variant (needs ev2 and max_num_neighbors == 16, else the reference-style walk runs and bitmap must be NULL):
 * 0 = by density: from 0.4 events per pixel on average and no bitmap, one wave per destination PIXEL at a time (lanes =
 * spiral steps, the neighbour pixels' newest entries held in registers and shared by all events of the pixel);
 * otherwise thread per event walking the spiral with the row in registers and an early exit after 16 neighbours;
 * 1 = row scan, radius 1 or 3 only (2R+1 contiguous range look-ups per event instead of (2R+1)^2 pixel
 * probes, candidates fetched 8 at a time -- sparse windows; other radii fall back to 0); 2 = wave per pixel whatever the
 * density (bitmap must be NULL, else as 3); 3 = thread per event whatever the density; 4 = the wave-per-pixel search in its
 * ring-band form (lanes take the pixels of a band of whole rings in row-major order, so bounds and entries are coalesced
 * loads, and source counts are put into spiral order through an LDS table; non-empty pixels only; bitmap must be NULL; not
 * picked by 0: it measures equal to 2 at 1280x720 and slower at 640x480); 5 = wave per pixel with the first pass in its
 * deep-pixel form (a lane keeps two positions in its neighbour's segment -- newest entry older than the event, oldest entry
 * inside its time window -- that move up from event to event of the pixel, instead of the newest entries in registers;
 * bitmap must be NULL; picked by 0 from 20 events per pixel on average); 6 = wave per pixel with a first pass of four events
 * at a time on 16 spiral steps each, entries in registers (picked by 0 from 5 events per pixel).  Identical rows in every case.
 * bitmap_prezeroed != 0: the caller already cleared the bitmap. */
int eventad_build_ell(const int32_t* sid, const int32_t* ts_s, const int32_t* ev2, const int32_t* key_s,
                      const uint32_t* pix_start, uint32_t* counters, int N, int B, int W, int H, int radius_px, int delta_t_us,
                      int max_num_neighbors, int queue_size, int32_t* ell_src, uint16_t* ell_step, int32_t* deg,
                      const int32_t* node_slot, const int32_t* node_vox, int nx, int ny, uint32_t* bitmap, int variant,
                      int bitmap_prezeroed, void* stream);
/* ELL -> the reference's edge buffer layout (i64[2,K], event numbering, -1 padded). */
int eventad_ell_to_edges(const int32_t* ell_src, const int32_t* deg, const int32_t* sid, int N, int max_num_neighbors,
                         int64_t K, int64_t* edges, void* stream);

/* ---- (C) first Layer on the event graph: replaces Layer.forward for conv_block1 --------------- */
/* (conv.py:59-72 = ConvBlock :10-28 + ConvBlockWithSkip :31-56 over MySplineConv.message_lut,
 * spline_conv.py:39-78).  lut = MySplineConv.lut_weights re-ordered by spiral step, f32[steps+2][Cin_pad][16], every
 * output column pre-multiplied by the folded eval-mode BatchNorm scale of that channel (components.py:9-12); row `steps`
 * = (lut[0] + conv.lin.weight^T) * scale, the matrix of the self edge that opens every ELL row (root weight,
 * spline_conv.py:72-73); row `steps + 1` = zeros (idle lanes multiply by it instead of branching).  shift = folded BatchNorm shift (conv 2: of the conv plus of the skip branch); wskip =
 * lin.mlp.weight^T [20][16] times the skip BatchNorm's scale.  The input-channel rows of every matrix are stored in the
 * order [i][j] (row i*4 + j = channel 4j+i for i < 4, channel 16+j for i = 4): the 4 lanes of a destination split the
 * input channels.  work u32[256]: per-XCD work counters, one per 128-byte line, ZERO on entry (the waves pull groups of
 * 64 destinations from them), left non-zero; each call needs its own zeroed 256 words.  max_num_neighbors must be 16;
 * N <= 53,687,091 (32-bit byte offsets into the 80-byte rows), else EVAD_ERR_ARG. */
int eventad_l1_conv1(const float* x0s, const int32_t* ell_src, const uint16_t* ell_step, const int32_t* deg,
                     const float* lut, const float* shift, float* y1, uint32_t* work, int N, int steps,
                     int max_num_neighbors, void* stream);
int eventad_l1_conv2(const float* y1, const float* x0s, const int32_t* ell_src, const uint16_t* ell_step,
                     const int32_t* deg, const float* lut, const float* wskip, const float* shift, float* y2,
                     uint32_t* work, int N, int steps, int max_num_neighbors, void* stream);

/* ---- (D) sampling_skip + Pooling.forward: replaces net.py:15-17,200-228 + pooling.py:51-97 ------ */
/* Slot table over S = nx*ny*(B+1) voxel slots: feat u32[S,Ca+Cm], sums u64[S,2], tsum 8 bytes[S] (2^-40 fixed point),
 * cnt u32[S], maxidx i32[S].  Level 0 passes key_s (pixel keys) and nidx = sid; pooled levels pass
 * npix/nbatch and n_dev (device-side node count).  The tables are cleared first unless prezeroed != 0 (the caller
 * cleared them, e.g. with one memset over the tables of every level).  bitmap (nullable, pooled levels only, already
 * cleared): also mark, for every incoming CSR edge (rowptr/csr_src) of every node, the source voxel in the destination
 * voxel's neighbour bitmap (= eventad_pool_edges_mark_csr in the same launch). */
int eventad_pool_scatter(const float* xa, int lda, int Ca, const float* map_nhwc, int map_h, int map_w, int Cm,
                         const float* npos, const int32_t* npix, const int32_t* nbatch, const int32_t* key_s,
                         const int32_t* nidx, int n_max, const int32_t* n_dev, int W, int H, float vsx, float vsy,
                         int nx, int ny, int B, int mean_aggr, uint32_t* feat, uint64_t* sums, double* tsum,
                         uint32_t* cnt, int32_t* maxidx, int32_t* node_slot, uint32_t* flags, int prezeroed,
                         const uint32_t* rowptr, const int32_t* csr_src, uint32_t* bitmap, void* stream);
/* Level-0 pooling without atomics: one wave per voxel slot walks the slot's pixel rows in the pixel CSR
 * (events of a voxel row are contiguous in sorted order), samples the 64-channel CNN map per member and
 * keeps running maxima in registers.  vx_lo i32[nx+1] / vy_lo i32[ny+1] = first pixel of each voxel column
 * / row.  Writes feat f32[S,80] (plain floats: finalize mode 2), sums, tsum, cnt, maxidx for EVERY slot.
 * bitmap != NULL: the wave also ORs the marks of all in-edges of the slot's members (ELL rows ell_src / deg, source voxels
 * node_vox as written by eventad_sort_events) into the slot's three neighbour-bitmap words and stores them -- the
 * cluster[edge_index] pass of pooling.py:57-61 without atomics and without a pre-zeroed table (replaces the bitmap
 * argument of eventad_build_ell / eventad_pool_edges_mark_ell); counters[0] receives the window-overflow flag.
 * slot_vd != NULL (needs bitmap): u32[S], receives per slot (occupied << 31 | pooled in-edges), which lets
 * eventad_pool_finalize_edges scan with one workgroup per 1024 slots instead of a single one. */
int eventad_pool_seg0(const float* y2, const float* map_nhwc, int map_h, int map_w, const float* posn_s,
                      const int32_t* key_s, const int32_t* sid, const int32_t* node_slot, const uint32_t* pix_start,
                      const int32_t* vx_lo, const int32_t* vy_lo, const int32_t* ovf_list, uint32_t* counters,
                      int N, int B, int W, int H, int nx, int ny, float* feat, uint64_t* sums, double* tsum,
                      uint32_t* cnt, int32_t* maxidx, const int32_t* ell_src, const int32_t* deg, const int32_t* node_vox,
                      uint32_t* bitmap, uint32_t* slot_vd, void* stream);
int eventad_pool_finalize(const uint32_t* feat, const uint64_t* sums, const double* tsum, const uint32_t* cnt,
                          const int32_t* maxidx, const int32_t* batch_of_index, int nx, int ny, int B, int C,
                          int feat_mode /* 0 ordered-uint max, 1 mean, 2 float max */, int level0, int W, int H, float wh_inv_x, float wh_inv_y, uint32_t* occ_scan,
                          uint32_t* scan_tmp, float* xout, int ldx, float* npos_out, int32_t* npix_out,
                          int32_t* nbatch_out, int32_t* node_slot_out, int32_t* slot2node, int32_t* n_out_dev,
                          void* stream);
/* eventad_pool_finalize + eventad_pool_edges_emit in two launches (tables of up to 131072 slots): one single-workgroup
 * pass does both exclusive scans (slot occupancy = node numbering; popcount of the occupied slots' bitmap = CSR row
 * pointers), one launch writes node attributes, features and CSR columns.  The bitmap must be complete (marked)
 * before the call.  rowptr u32[S+2].  Outputs are bit-identical to the separate calls. */
int eventad_pool_finalize_edges(const uint32_t* feat, const uint64_t* sums, const double* tsum, const uint32_t* cnt,
                                const int32_t* maxidx, const int32_t* batch_of_index, const uint32_t* bitmap, int nx, int ny,
                                int B, int C, int feat_mode, int level0, int W, int H, float wh_inv_x, float wh_inv_y,
                                uint32_t* occ_scan, float* xout, int ldx, float* npos_out, int32_t* npix_out,
                                int32_t* nbatch_out, int32_t* node_slot_out, int32_t* slot2node, int32_t* n_out_dev,
                                uint32_t* rowptr, int32_t* csr_src, int32_t* csr_dxy, int32_t* e_out_dev, const uint32_t* slot_vd /* NULL, or eventad_pool_seg0's per-slot words */, void* stream);
int eventad_pool_edges_mark_ell(const int32_t* ell_src, const int32_t* deg, const int32_t* node_slot, int N,
                                int max_num_neighbors, int nx, int ny, int B, uint32_t* bitmap, uint32_t* flags,
                                void* stream);
int eventad_pool_edges_mark_csr(const uint32_t* rowptr, const int32_t* csr_src, const int32_t* node_slot, int n_max,
                                const int32_t* n_dev, int nx, int ny, int B, uint32_t* bitmap, uint32_t* flags,
                                int prezeroed, void* stream);
int eventad_pool_edges_emit(const uint32_t* bitmap, const int32_t* node_slot_out, const int32_t* slot2node,
                            const int32_t* npix_out, int n_max_out, const int32_t* n_out_dev, int nx, int ny,
                            uint32_t* rowptr, uint32_t* scan_tmp, int32_t* csr_src, int32_t* csr_dxy,
                            int32_t* e_out_dev, void* stream);

/* ---- (E) dense pieces: tail Layers (conv.py:59-72 on pooled graphs), 1x1 reducers ------------- */
/* C = epilogue(A.B): exact-fp32 MFMA GEMM; m_dev (nullable) = device-side row count <= M. */
int eventad_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, const int32_t* m_dev,
                     int N, int K, const float* bias, const float* scale, const float* shift, const float* residual,
                     int ldr, int relu, void* stream);
/* Which of the 25 kernel cells the blends of a pooled level read, per 64 source nodes (spline_conv.py:27-36: an edge's LUT
 * entry is a 4-tap blend of the cells around its offset): live u32[n_words >= ceil(n/64)], zero on entry, bit c of word i = some
 * out-edge of nodes 64i..64i+63 has a tap in cell c.  e_dev = device-side edge count <= e_max. */
int eventad_tail_live_cells(const int32_t* csr_src, const int32_t* csr_dxy, const int32_t* e_dev, int e_max, float div_x,
                            float div_y, uint32_t* live, int n_words, void* stream);
/* Z = X.[W_0|..|W_24|W_root(|W_skip)] (N = 26*64 or 27*64 columns) without the 64x64 tiles of cell blocks that `live` does
 * not name: those parts of Z are left untouched and must not be read (eventad_tail_blend on the same edges does not). */
int eventad_tail_gemm(const float* X, int ldx, const float* Wc, int ldw, float* Z, int ldz, int n_max, const int32_t* n_dev,
                      int N, int K, const uint32_t* live, void* stream);
/* out_i = ReLU(BN(sum_e sum_taps b*Z[src, cell] + Z[i, root]) [+ BN_skip(Zskip[i, skip])]), Cout = 64. */
int eventad_tail_blend(const float* Z, int ldz, int root_col, const float* Zskip, int ldzs, int skip_col,
                       const uint32_t* rowptr, const int32_t* csr_src, const int32_t* csr_dxy, int n_max,
                       const int32_t* n_dev, float div_x, float div_y, const float* bn_scale, const float* bn_shift,
                       const float* skip_scale, const float* skip_shift, float* out, int ldo, void* stream);
/* eventad_tail_blend of a Layer's second convolution followed by eventad_pool_scatter of the next level (max: channels
 * [y | 64 sampled map channels]; mean: bookkeeping only, the caller runs the mean gather through eventad_pool_scatter's
 * path) in one launch: same arithmetic, same integer atomics, same results.  The slot tables (feat u32[S,128], sums, tsum,
 * cnt, maxidx, bitmap) must be zero on entry; node_slot i32[n_max] receives each node's slot. */
int eventad_tail_blend_pool(const float* Z, int ldz, int root_col, const float* Zskip, int ldzs, int skip_col,
                            const uint32_t* rowptr, const int32_t* csr_src, const int32_t* csr_dxy, int n_max,
                            const int32_t* n_dev, float div_x, float div_y, const float* bn_scale, const float* bn_shift,
                            const float* skip_scale, const float* skip_shift, float* out, int ldo, const float* map_nhwc,
                            int map_h, int map_w, const float* npos, const int32_t* npix, const int32_t* nbatch, int W,
                            int H, float vsx, float vsy, int nx, int ny, int B, int mean_aggr, uint32_t* feat,
                            uint64_t* sums, double* tsum, uint32_t* cnt, int32_t* maxidx, int32_t* node_slot,
                            uint32_t* flags, uint32_t* bitmap, void* stream);
/* The mean aggregation of eventad_pool_scatter (mean_aggr = 1) without its bookkeeping pass: feat f32[S, Ca+Cm] = sum of
 * the member rows per slot, members added in ascending node order; node_slot as left by eventad_tail_blend_pool. */
int eventad_pool_mean_gather(const float* xa, int lda, int Ca, const float* map_nhwc, int map_h, int map_w, int Cm,
                             const float* npos, const int32_t* nbatch, const int32_t* node_slot, int n_max,
                             const int32_t* n_dev, int W, int H, int nx, int ny, int B, float* feat, void* stream);
/* NCHW -> NHWC of externally supplied CNN feature maps, up to 8 maps in one launch (LDS-tiled transposes).
 * desc = HOST array, 6 int64 per map: [0] in [1] out [2] B [3] C [4] H [5] W. */
int eventad_nchw_to_nhwc_group(int n_maps, const int64_t* desc, void* stream);

/* ---- (F) EventAD head --------------------------------------------------------------------------- */
/* ROI pooling of out4 node features per box: replaces EventADModel.extract_features' box loop and
 * _process_bbox (reference models/EventAD.py:382-397, 406-499).  feats f32[B,n_frames,31,64] must be
 * zero-filled by the caller; `frame` = 1 for bbox (current), 0 for bbox0 (previous). */
int eventad_roi_pool(const float* node_x, const float* node_pos, const int32_t* node_batch, const int32_t* n_dev,
                     int n_max, const float* bbox, const int64_t* bbox_batch, int M, int frame, int n_frames,
                     int width, int height, int B, float* feats, void* stream);

/* General GEMM form used by the head: trans_a / trans_b select the stored layout ([M,K] vs [K,M],
 * [K,N] vs [N,K] = PyTorch's Linear weight); accumulate != 0 adds into C.  Replaces the M=1 cuDNN /
 * cuBLAS calls behind nn.GRU / nn.Linear in the reference head (models/EventAD.py:16-28,67,86). */
int eventad_gemm_f32_ex(int trans_a, int trans_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        int M, int N, int K, const float* bias, int relu, int accumulate, void* stream);
/* (clip, track) tables + current-frame feature gather: replaces EventAD.py:229-255.  X f32[B*32,64],
 * CO f32[B*32,4], target/valid i32[B*32]; labels f32[M] (nullable); with labels NULL and label_col in 0..5 the label of
 * a box is that column of its bbox row (the reference's callers pass labels = bbox[:,4]); label_col < 0: targets 0. */
int eventad_head_tables(const float* bbox, const int64_t* bbox_batch, int M, const float* labels, int label_col,
                        const float* feats, int B, int width, int height, float* X, float* CO, int32_t* target,
                        int32_t* valid, void* stream);
/* Fused recurrent step of the head (Event_GRUNet / Cor_GRUNet cell, gate order r,z,n, + SpatialAttention over the valid
 * tracks; models/EventAD.py:62-138, 320-323) and its backward: GRU cell + attention
 * of one clip, one 1024-thread workgroup per problem, up to 3 problems per launch.
 * fwd desc = HOST array, 12 int64 per problem: [0] gi [1] gh [2] hprev (= the state the attention keeps for invalid
 * tracks) [3] gates out [4] hn out [5] dropout mask(0=none) [6] xdrop out(0=none) [7] attention w [8] alpha out
 * [9] snext out [10] valid i32[32] [11] H (256|32).
 * bwd desc = 20 int64 per problem: [0] hn [1] attention w [2] alpha [3] dsnext [4] dhn_out(0=none) [5] valid
 * [6] dw accumulator [7] dx_extra(0=none) [8] mask(0=none) [9] gates [10] gh [11] hprev [12] dgi out [13] dgh out
 * [14] dsprev out [15] H [16] dsnext slabs f32[n][32][H] (0=none): dsnext := (sum of slabs, in order) + dsnext [17] n
 * [18] dx_extra slabs f32[n][32][H] (0=none): dx_extra := sum of slabs (the [7] pointer is then ignored) [19] n. */
int eventad_gru_att_fwd_group(int n_prob, const int64_t* desc, void* stream);
int eventad_gru_att_bwd_group(int n_prob, const int64_t* desc, void* stream);
/* One clip stage of the head's forward recurrence in ONE launch (32 workgroups x 1024 threads, 142 KB of LDS each): the
 * SpatialAttention of the previous clip (models/EventAD.py:100-138, 320-323; recomputed by every workgroup), the
 * recurrent products state . Whh^T of up to three GRU layers -- and, for slot 1, the input product -- for the
 * workgroup's hidden columns (fp32 MFMA), and the GRU cell (EventAD.py:62-97; gate order r,z,n).  Replaces the pair
 * eventad_gemm_group + eventad_gru_att_fwd_group per clip; the buffers it fills are the ones eventad_gru_att_bwd_group reads.
 * desc = HOST array, 3 slots x 19 int64: slot 0 = 256-wide layer, input product gi (incl. bias) precomputed; slot 1 =
 * 256-wide layer, input product xin . wih^T + bih computed here; slot 2 = 32-wide layer, gi precomputed.  Per slot:
 * [0] present [1] hn_prev f32[32,H] cell output of the previous clip (0 = none: the state is read from [6] as it is)
 * [2] s_prev f32[32,H] state that entered the previous clip (kept for invalid tracks) [3] valid_prev i32[32]
 * [4] attention w f32[H] [5] alpha_prev out f32[32] [6] s_cur f32[32,H] state entering this clip (out unless [1] = 0)
 * [7] whh f32[3H,H] [8] bhh [9] gi f32[32,3H] [10] xin f32[32,H] [11] wih f32[3H,H] [12] bih [13] gh out f32[32,3H]
 * [14] gates out f32[32,3H] [15] hn out f32[32,H] [16] dropout mask (0 = none) [17] xdrop out = hn * mask (0 = none)
 * [18] ghr out f32[32,3H] = state . whh^T without the bias (0 = none; eventad_head_stage_bwd reads it). */
int eventad_head_stage_fwd(const int64_t* desc, void* stream);
/* Backward of one clip stage in ONE launch (32 workgroups x 1024 threads, 107 KB of LDS): for its 8 hidden columns a
 * workgroup forms the products dgh_next . whh (gradient of the state entering the next clip) and -- GRU layer 0 -- dgi_x . wih_x
 * (gradient of its dropout'd output) with fp32 MFMA, then does attention backward and GRU cell backward (arithmetic of
 * eventad_gru_att_bwd_group).  The attention needs per-track dot products over whole rows of a product no workgroup
 * holds completely: the stage that produced dgh_next leaves per-workgroup partial dots (pd), using the forward's raw
 * recurrent products (ghr).  Replaces eventad_gru_att_bwd_group + eventad_gemm_group + the slab reduce per clip.
 * desc = HOST array, 3 slots x 28 int64: slot 0 = GRU layer 1, slot 1 = GRU layer 0, slot 2 = coordinate GRU.  Per slot:
 * [0] present [1] hn [2] attention w [3] alpha [4] valid [5] dhn_out (0 = none) [6] dropout mask (0 = none) [7] gates
 * [8] gh [9] hprev [10] dgi out [11] dgh out [12] dsprev out f32[32,H] [13] dw accumulator f32[H] [14] has_next
 * [15] dsprev_next [16] dgh_next f32[32,3H] [17] whh [18] pd_in f32[32,32] [19] has_dx (slot 1 only) [20] dgi_x f32[32,768]
 * [21] wih_x [22] has_prev [23] ghr f32[32,3H] [24] hn_prev [25] alpha_prev [26] valid_prev [27] pd_out f32[32,32]. */
int eventad_head_stage_bwd(const int64_t* desc, void* stream);
/* Up to 8 independent small GEMMs per launch (one workgroup per 32x32 tile, its 4 waves split K in 64-deep
 * chunks and are summed in wave order = deterministic).  desc = HOST array, 16 int64 per problem: [0] A [1] B
 * [2] part scratch f32[ceil(K/256)][M][N] (only when K > 256) [3] C [4] bias(0=none) [5] lda [6] ldb [7] ldc [8] M [9] N [10] K
 * [11] trans_a [12] trans_b [13] accumulate [14] relu [15] tile counters i32[ceil(M/32)*ceil(N/32)] (0 = none; zero
 * on first use, every launch leaves them zero): with them the slabs of a K > 256 product are summed, in slab order,
 * by the last workgroup to reach the tile instead of by a second launch; the value 1 = leave the slabs of a K > 256
 * product un-summed in `part` for a consumer that adds them itself (eventad_gru_att_bwd_group does). */
int eventad_gemm_group(int n_prob, const int64_t* desc, void* stream);
/* The same products on 64x64 tiles: a workgroup's 4 waves own the quadrants of its tile over the whole K (one fmaf chain in
 * k order per output: no K split, no slabs, no second launch), operands staged through LDS once per workgroup.  For groups
 * that are bound by MFMA throughput rather than latency -- the head's batched weight gradients dW = dG^T . X over all clips
 * (what autograd does for nn.GRU / nn.Linear weights of models/EventAD.py:16-97).  Same descriptors; [2] and [15] are ignored. */
int eventad_gemm_group64(int n_prob, const int64_t* desc, void* stream);
/* fusion[3] Linear(256,2) + summed cross-entropy (EventAD.py:27,280-281) and its backward.  eventad_fuse_out with dPre
 * non-NULL also writes dPre f32[R,256] = d loss / d (pre-activation of the hidden layer), which eventad_out_bwd then need
 * not compute (pass its dPre = NULL); eventad_out_bwd also clears the n_clear floats at `clear` (nullable): accumulators
 * that the stages after it add into. */
int eventad_fuse_out(const float* F1r, const float* D2, const float* Wo, const float* bo, const int32_t* target,
                     const int32_t* valid, int R, float* logits, float* dlog, float* loss_rows, float* loss, float* dPre,
                     void* stream);
int eventad_out_bwd(const float* dlog, const float* Wo, const float* F1r, const float* D2, int R, float* dPre,
                    float* dWo, float* dbo, float* clear, int n_clear, const float* loss_rows /* nullable */,
                    float* loss /* nullable: adds up loss_rows in row order, for an eventad_fuse_out call made with loss = NULL */,
                    void* stream);
/* up to 12 column sums per launch; desc = HOST array, 5 int64 per problem: [0] A [1] out [2] lda [3] N [4] rows (0 = R) */
int eventad_colsum_group(int n, const int64_t* desc, int R, void* stream);
/* clip_grad_norm_ + AdamW on flat buffers (utils/train.py:27-53,141-152).  state f32[8]:
 * [0] step, [1] grad norm, [2] clip coefficient, [3] 1-beta1^t, [4] 1-beta2^t, [5] non-finite-norm count, [6] 1 if this
 * update was skipped.  A non-finite norm (the reference raises before optimizer.step(), utils/train.py:144-149) or an
 * all-zero gradient (no valid object: the reference's grads are None and AdamW skips them) leaves params, moments and
 * the step count untouched. */
int eventad_clip_adamw(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int n, double lr, double beta1,
                       double beta2, double eps, double weight_decay, double max_norm, float* partial, float* state,
                       void* stream);
/* The same update, which also leaves the step's (loss, gradient norm) in result_out f32[2] -- loss read from loss_in f32[1] --
 * from the update's own launch: what utils/train.py:123-152 reads back with .item() after the step, kept per step without a
 * copy launch behind the update (NULL result_out = eventad_clip_adamw). */
int eventad_clip_adamw_out(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int n, double lr, double beta1,
                           double beta2, double eps, double weight_decay, double max_norm, float* partial, float* state,
                           const float* loss_in, float* result_out, void* stream);

/* Dropout masks of one training step: out f32[n] (16-byte aligned) = 1/keep_prob with probability keep_prob, else 0
 * (the inter-layer dropout of nn.GRU and nn.Dropout of the fusion MLP, models/EventAD.py:26,86; the reference draws them
 * from torch's global generator).  Counter-based: value i depends on (seed, step, i) only; step is read from step_dev (a
 * device word the caller sets before launching or replaying) when that is non-null, so a captured launch serves every step. */
int eventad_dropout_masks(float* out, int64_t n, double keep_prob, int64_t seed, int64_t step, const int64_t* step_dev,
                          void* stream);

/* ---- (G) CNN feature branch: replaces HookModule.forward over torchvision resnet50 ---------------- */
/* (reference src/dagr/model/networks/net_img.py:122-134, net.py:41-49).  NHWC fp32 implicit-GEMM convolution
 * on fp32 MFMA: in f32[B,IH,IW,Cin] (Cin % 4 == 0), w f32[KH*KW*Cin, Cout] with k = (kh*KW+kw)*Cin+ci
 * (Cout % 4 == 0), out f32[B,OH,OW,Cout]; epilogue +bias, *scale+shift (folded eval BatchNorm), +residual, ReLU.
 * workspace f32[workspace_floats] (nullable): lets shapes with too few output tiles for 256 CUs (ResNet layers 3-4)
 * run split-K -- up to 8 slabs of B*OH*OW*Cout floats, summed in slab order by a second launch. */
int eventad_conv2d_nhwc(const float* in, const float* w, float* out, const float* bias, const float* scale,
                        const float* shift, const float* residual, int B, int IH, int IW, int Cin, int OH, int OW,
                        int Cout, int KH, int KW, int stride, int pad, int relu, float* workspace,
                        int64_t workspace_floats, void* stream);
/* Floats of split-K workspace eventad_conv2d_nhwc uses for this shape when given at least that many (0: one pass).
 * No counterpart in the reference (cuDNN manages its own workspaces).  The workspace is written by the convolution launch
 * and read by the slab-sum launch behind it: passes that may overlap in time on one GPU need a workspace each. */
int eventad_conv2d_workspace_floats(int B, int OH, int OW, int Cin, int Cout, int KH, int KW, int64_t* floats);
/* stem: bn1 + ReLU + 3x3/2 max-pool (pad 1), NHWC. */
int eventad_bn_relu_maxpool(const float* in, const float* scale, const float* shift, float* out, int B, int IH, int IW,
                            int C, int OH, int OW, void* stream);
/* image f32[B,3,H,W] -> f32[B,H,W,4] (zero 4th channel) so the 7x7 stem reads float4s. */
int eventad_image_to_nhwc4(const float* in, float* out, int B, int H, int W, void* stream);

/* ---- (H) asynchronous (event-by-event) inference: replaces the CUDA extension `asy_tools` ---------------------- */
/* (reference src/dagr/asynchronous/asy_tools/main.cu:14-188, pybind :239-244).  indices i64[K] list the rows to touch.
 * masked_isdiff(indices, x_old, x_new, atol, rtol) :96-125 (kernel :14-41): indices[i] <- -1 unless row indices[i] of
 * x_old / x_new f32[*, C] differs in some channel by more than atol + rtol * x_new; the caller drops the -1 entries. */
int eventad_masked_isdiff(int64_t* indices, const float* x_old, const float* x_new, int K, int C, float atol, float rtol,
                          void* stream);
/* masked_inplace_BN(indices, x, x_out, running_mean, running_var, weight, bias, eps) :69-94 (kernel :43-67). */
int eventad_masked_inplace_bn(const int64_t* indices, const float* x, float* x_out, const float* running_mean,
                              const float* running_var, const float* weight, const float* bias, int K, int C, float eps,
                              void* stream);
/* masked_lin(indices, x_in, x_out, weight, bias, add) :215-236 and masked_lin_no_bias :190-212 (bias == NULL);
 * kernels :128-188.  weight f32[Cout, Cin]; add != 0 accumulates onto x_out. */
int eventad_masked_lin(const int64_t* indices, const float* x_in, float* x_out, const float* weight, const float* bias,
                       int K, int Cin, int Cout, int add, void* stream);

/* ---- (I) backward of the event-level stage: LUT spline convolution, pool1, feature sampler (fine-tuning the backbone) */
/* The reference obtains these from autograd over MySplineConv's gather + scatter-add (spline_conv.py:39-78).  All of
 * them are free of float atomics: bitwise reproducible.
 * eventad_ell_transpose: out-edges per source from the ELL adjacency.  out_start u32[N+1], out_edge i32[16N] (key =
 * destination * 16 + slot, ascending inside a source's list; E = out_start[N] entries are valid).  Scratch: cnt
 * u32[N+1], scan_tmp u32[N/2048+2], arrival u16[16N], tmp i32[16N]. */
int eventad_ell_transpose(const int32_t* ell_src, const int32_t* deg, int N, uint32_t* out_start, int32_t* out_edge,
                          uint32_t* cnt, uint32_t* scan_tmp, uint16_t* arrival, int32_t* tmp, void* stream);
/* dx_j = sum over out-edges (j -> i) of LUT[cell] g_i.  g f32[N,16] = dL/d(conv output); lut f32[steps+1][cin_pad][16] with
 * the input-channel rows in the forward kernel's [i][j] order, row `steps` = self-edge matrix; dx f32[N, cin_pad];
 * cin_pad = 20 or 16; the table is staged in LDS when its (steps+1) * cin_pad * 64 bytes fit in 150 KB (search radius <= 4 px
 * for cin_pad = 20, <= 5 px for 16), else its rows are read from memory. */
int eventad_l1_conv_bwd_data(const float* g, const uint32_t* out_start, const int32_t* out_edge, const uint16_t* ell_step,
                             const float* lut, float* dx, int N, int steps, int cin_pad, void* stream);
/* dlut[c] = sum over edges with spiral step c of x_src (x) g_dst, f32[steps+1][cin_pad][16] in natural channel order (row
 * `steps` = the self edges: gradient of LUT[0] + W_root).  partial = scratch f32[eventad_l1_conv_bwd_weight_parts(N)]
 * [steps+1][cin_pad][16] (one table per 256 destinations).  (steps+1) * 512 bytes + 24 KB must fit in LDS (search radius
 * <= 7 px), else -1. */
int eventad_l1_conv_bwd_weight_parts(int N);
int eventad_l1_conv_bwd_weight(const float* x, const float* g, const int32_t* ell_src, const uint16_t* ell_step,
                               const int32_t* deg, float* partial, float* dlut, int N, int steps, int cin_pad, void* stream);
/* Backward of pool1's voxel max (Pooling.forward with aggr = max, pooling.py:51-97; the reference's gradient is
 * torch_scatter's arg-based scatter_max backward).  Inputs as eventad_pool_seg0 plus dfeat f32[S,80] = dL/d(slot feature
 * table), S = nx*ny*(B+1).  Out: dy2 f32[N,16] (layer-1 output channels) and dsamp f32[N,64] (sampled map channels), one
 * row per event, pixel-sorted numbering; the gradient of (slot, channel) goes to the member that attains the maximum
 * (smallest sorted position among exactly equal members). */
int eventad_pool_seg0_bwd(const float* y2, const float* map_nhwc, int map_h, int map_w, const float* posn_s,
                          const int32_t* key_s, const int32_t* node_slot, const uint32_t* pix_start,
                          const int32_t* vx_lo, const int32_t* vy_lo, const int32_t* ovf_list, const uint32_t* counters,
                          int N, int B, int W, int H, int nx, int ny, const float* dfeat, float* dy2, float* dsamp,
                          void* stream);
/* Transpose of the bilinear feature sampler (autograd through F.grid_sample, net.py:211-228): dmap f32[B,map_h,map_w,C]
 * (NHWC, C = 16 or 64) from g = rows of `ldg` floats per event (pixel-sorted numbering) whose first C entries are the
 * gradient of the sampled channels.  Gather per map pixel over the pixel CSR, fixed summation order. */
int eventad_sample_bwd(const float* g, int ldg, int C, const float* posn_s, const uint32_t* pix_start, int N, int B,
                       int W, int H, int map_h, int map_w, float* dmap, void* stream);

/* ---- (I') backward of the pooled levels: Layers 2..5 (conv.py:59-72 over MySplineConv, spline_conv.py:39-78), pools
 * 2..4 (pooling.py:51-97) and the samplers at pooled nodes (net.py:200-228).  The reference obtains all of it from
 * autograd.  With Z = X.[W_0..W_24 | root (| skip)] the forward is y = ReLU(scale * (blend(Z) + Z[root]) + shift (+ sscale *
 * Z[skip] + sshift)); its backward is eventad_relu_scale_bwd, eventad_tail_blend_bwd over the transposed graph, and two
 * eventad_gemm_f32_ex products (dX = dZ.W^T, dW = X^T.dZ).  No float atomics: bitwise reproducible. */
/* Out-edges per source of a pooled graph in CSR form (rowptr u32[n+1], csr_src i32[E], rows = destinations).  out_start
 * u32[n+1], out_edge i32[E] (edge ids, ascending inside a source's list), edge_dst i32[E] (destination of every edge).
 * Scratch: cnt u32[n+1], scan_tmp u32[n/2048+2], tmp i32[E]. */
int eventad_csr_transpose(const uint32_t* rowptr, const int32_t* csr_src, int n, int E, uint32_t* out_start,
                          int32_t* out_edge, int32_t* edge_dst, uint32_t* cnt, uint32_t* scan_tmp, int32_t* tmp,
                          void* stream);
/* out_a[i,c] = g[i,c] * [y[i,c] > 0] * scale_a[c] (and out_b with scale_b when out_b != NULL): through the ReLU and the
 * folded eval-mode BatchNorm scale(s).  out_a / out_b f32[n,C] dense. */
int eventad_relu_scale_bwd(const float* g, int ldg, const float* y, int ldy, const float* scale_a, const float* scale_b,
                           int n, int C, float* out_a, float* out_b, void* stream);
/* dZ f32[n, ldz]: columns [0, 25*64) = gradient of the 25 kernel-cell products of every SOURCE node (gather over its
 * out-edges, 4 bilinear taps each, same basis as eventad_tail_blend), block root_col = d (the root term), block skip_col
 * = dskip (f32[n,64], nullable).  d f32[n, ldd] = gradient of the pre-BatchNorm convolution output (64 channels). */
int eventad_tail_blend_bwd(const float* d, int ldd, const uint32_t* out_start, const int32_t* out_edge,
                           const int32_t* edge_dst, const int32_t* csr_dxy, int n, float div_x, float div_y, float* dZ,
                           int ldz, int root_col, const float* dskip, int skip_col, void* stream);
/* Backward of eventad_pool_scatter (+ finalize) on a pooled level: the gradient of (output node, channel) goes to the
 * member that attains the maximum (smallest member index among exactly equal ones: torch_scatter's CPU scatter_max), or
 * to every member divided by the member count (mean != 0).  Inputs as the forward (xa = Ca layer-output channels of the
 * n_in input nodes, Cm channels sampled from map_nhwc at the nodes' positions), node_slot i32[n_in] / slot2node i32[S] /
 * cnt u32[S] as the forward left them, xout f32[n_out, ldo] = the pooled features, dxout f32[n_out, ldd] their gradient.
 * Out: dxa f32[n_in, Ca], dsamp f32[n_in, Cm].  Scratch: argm i32[n_out * (Ca + Cm)]. */
int eventad_pool_max_bwd(const float* xa, int lda, int Ca, const float* map_nhwc, int map_h, int map_w, int Cm,
                         const float* npos, const int32_t* nbatch, const int32_t* node_slot, const int32_t* slot2node,
                         const uint32_t* cnt, int n_in, int n_out, int W, int H, int mean, const float* xout, int ldo,
                         const float* dxout, int ldd, int32_t* argm, float* dxa, float* dsamp, void* stream);
/* Transpose of the bilinear sampler at pooled nodes: dmap f32[B, map_h, map_w, C] (NHWC) from dsamp f32[n, C].  npos /
 * nbatch = the nodes' positions and clips, occ u32[S+1] = the exclusive occupancy scan of the level's slot table (nodes
 * are numbered in slot order), nx, ny = its voxel grid, C <= 64.  Gather per map pixel, fixed summation order. */
int eventad_sample_nodes_bwd(const float* dsamp, int C, const float* npos, const int32_t* nbatch, const uint32_t* occ,
                             int nx, int ny, int B, int W, int H, int map_h, int map_w, float* dmap, void* stream);

/* ---- (K) asynchronous (event-by-event) update of the event-level stage ------------------------------------------------ */
/* Replaces src/dagr/asynchronous/ (make_model_asynchronous __init__.py:41-110, conv.py:96-190, max_pool.py:13-26,245-252)
 * over the incremental graph builder (src/dagr/graph/ev_graph.py:45-103) for the event-level stage: k new events are
 * appended BEHIND the arrays of the last dense pass (eventad_sort_events ... eventad_pool_seg0 on n_dense events; the
 * arrays must hold `capacity` rows), as nodes n_cur .. n_cur+k-1 with sorted position == event id.  Edges point from older
 * to newer events, so the rows of the existing nodes do not change.  Call order per update:
 * eventad_incr_insert, eventad_incr_edges, eventad_l1_conv1_rows, eventad_l1_conv2_rows, eventad_incr_pool0, then
 * eventad_pool_finalize_edges (slot_vd = NULL) and the pooled stage as after a dense pass.
 * pix_new_head i32[B*H*W] (eventad_incr_reset after every dense pass) + new_next i32[capacity - n_dense]: per-pixel lists
 * of the appended events. */
int eventad_incr_reset(int32_t* pix_new_head, int64_t pixels, void* stream);
/* The new events' node rows, exactly as a dense pass writes them: pos_new f32[k,3] normalised (format_data output),
 * polarity_new f32[k], batch_new i64[k]; error flags as eventad_sort_events go to counters[0]. */
int eventad_incr_insert(const float* pos_new, const float* polarity_new, const int64_t* batch_new, int k, int n_cur,
                        int n_dense, int capacity, int B, int W, int H, int T, const float* map0_nhwc, int map_h, int map_w,
                        float vsx, float vsy, int nx, int ny, int32_t* key_s, int32_t* ts_s, int32_t* ev2, int32_t* sid,
                        int32_t* bev, float* posn_s, float* x0s, int32_t* node_slot, int32_t* node_vox, int32_t* pix_new_head,
                        int32_t* new_next, uint32_t* counters, void* stream);
/* ELL rows of nodes n_first .. n_first+k-1: the walk of ev_graph.cu:15-80 over FIFO = appended events of the pixel (newest
 * first) ++ the dense pass's segment (newest first), queue_size entries in all -- the queue as ev_graph.cu:169-212 leaves
 * it after inserting the new events.  More than 16 appended events in one pixel set flag 16 in counters[0]. */
int eventad_incr_edges(const int32_t* ev2, const int32_t* key_s, const uint32_t* pix_start, const int32_t* pix_new_head,
                       const int32_t* new_next, uint32_t* counters, int n_first, int k, int n_dense, int W, int H, int radius_px,
                       int delta_t_us, int max_num_neighbors, int queue_size, int32_t* ell_src, uint16_t* ell_step,
                       int32_t* deg, void* stream);
/* eventad_l1_conv1 / eventad_l1_conv2 on rows [n_first, N) only (work: 256 words each, cleared by the call). */
int eventad_l1_conv1_rows(const float* x0s, const int32_t* ell_src, const uint16_t* ell_step, const int32_t* deg,
                          const float* lut, const float* shift, float* y1, uint32_t* work, int n_first, int N, int steps,
                          int max_num_neighbors, void* stream);
int eventad_l1_conv2_rows(const float* y1, const float* x0s, const int32_t* ell_src, const uint16_t* ell_step,
                          const int32_t* deg, const float* lut, const float* wskip, const float* shift, float* y2,
                          uint32_t* work, int n_first, int N, int steps, int max_num_neighbors, void* stream);
/* Fold rows n_first .. n_first+k-1 into pool1's slot table as eventad_pool_seg0 left it (feat f32[S,80] by maximum, sums /
 * tsum / cnt by addition, maxidx by maximum, bitmap by OR of the new in-edges' marks): integer atomics only, order
 * independent, so the table equals the one a dense pass over all events builds. */
int eventad_incr_pool0(const float* y2, const float* map_nhwc, int map_h, int map_w, const float* posn_s,
                       const int32_t* key_s, const int32_t* node_slot, const int32_t* node_vox, const int32_t* ell_src,
                       const int32_t* deg, int n_first, int k, int W, int H, float* feat, uint64_t* sums, double* tsum,
                       uint32_t* cnt, int32_t* maxidx, uint32_t* bitmap, uint32_t* counters, void* stream);

/* ---- (J) streams restricted to a CU set (no counterpart in the reference) ------------------------------------------ */
/* A default-priority HIP stream whose kernels may only use the CUs whose bit is set in mask (u32[words], bit i = CU i).
 * TrainStep uses a pair of them with disjoint masks for the backbone graph and the head graph. */
int eventad_stream_create_cu_mask(const uint32_t* mask, int words, void** stream_out);
int eventad_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif
