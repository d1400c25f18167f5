// eventad_amd/csrc/incr.hip -- event-by-event (asynchronous) update of the event-level stage.
//
// Reference: src/dagr/asynchronous/ (make_model_asynchronous __init__.py:41-110; conv.py:96-190: new nodes and their
// in-edges are appended to the stored graph and only the affected rows are recomputed; max_pool.py:13-26,245-252: the
// touched voxels are re-pooled) over the incremental graph builder (src/dagr/graph/ev_graph.py:45-103: every forward call
// inserts the new events into the per-pixel FIFO and searches edges for the new events only).
//
// The radius graph points from older to newer events, so appending events never changes the in-edges -- hence the
// layer-1 rows -- of the events already there.  An update of k new events therefore costs O(k) at the event level:
//   1. eventad_incr_insert : the new events become nodes n_cur .. n_cur+k-1 BEHIND the pixel-sorted array of the last
//                            dense pass (sorted position == event id for them); their layer-1 input rows are sampled, and
//                            each is pushed onto a per-pixel list of appended events (the dense pass's pixel CSR stays);
//   2. eventad_incr_edges  : wave per new event, the reference's spiral walk over FIFO = [appended events of the pixel,
//                            newest first] ++ [the pixel's CSR segment, newest first], depth Q in total;
//   3. eventad_l1_conv{1,2}_rows (layer1.hip): the two convolutions on the k new rows;
//   4. eventad_incr_pool0  : the new rows are folded into pool1's voxel-slot table (max of the features, integer position
//                            sums, counts, neighbour-bitmap marks of the new in-edges) -- all order independent, so the
//                            table equals what a dense pass over all events builds;
// after which the host re-runs the (cheap, <= 2240 nodes per clip) finalize + pooled stage on the updated table.
#include "common.h"

int evad_spiral_table(int radius, int* steps, const int16_t** table);    // graph.hip: device table [steps][2] of the radius

#define INCR_NEWCAP 16              // appended events of ONE pixel the edge search collects (more: flag, the caller re-runs densely)

__device__ __forceinline__ int incr_denorm(float p, float scale) { return (int)__fadd_rn(__fmul_rn(scale, p), 1e-3f); }

// Same arithmetic, statement for statement, as the dense pass (graph.hip: k_prep_events + k_gather_sorted): the rows of an
// appended event are bit-identical to the ones a dense pass over all events writes for it.
__global__ void k_incr_insert(const float* __restrict__ pos, const float* __restrict__ pol, const long long* __restrict__ batch, int k,
                              int n_cur, int n_dense, int B, int W, int H, int T, const float* __restrict__ map0, int h_in, int w_in,
                              float vsx, float vsy, int nx, int ny, int* __restrict__ key_s, int* __restrict__ ts_s,
                              int2* __restrict__ ev2, int* __restrict__ sid, int* __restrict__ bev, float* __restrict__ posn_s,
                              float* __restrict__ x0s, int* __restrict__ node_slot, int* __restrict__ node_vox, int* __restrict__ head,
                              int* __restrict__ next, unsigned* __restrict__ flags) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const float px = pos[3 * j], py = pos[3 * j + 1], pt = pos[3 * j + 2];
    int x = incr_denorm(px, (float)W), y = incr_denorm(py, (float)H);
    const int t = incr_denorm(pt, (float)T);
    int b = (int)batch[j];
    unsigned f = 0;
    if (x < 0 || x >= W || y < 0 || y >= H || b < 0 || b >= B) {
        f |= EVAD_FLAG_POS_OUT_OF_FOV;
        x = min(max(x, 0), W - 1); y = min(max(y, 0), H - 1); b = min(max(b, 0), B - 1);
    }
    if (!(pt >= 0.f && pt <= 1.f)) f |= EVAD_FLAG_T_OUT_OF_RANGE;
    if (j > 0 && (int)batch[j - 1] == b && incr_denorm(pos[3 * j - 1], (float)T) > t) f |= EVAD_FLAG_T_NOT_MONOTONE;
    const int key = x + W * (y + H * b), r = n_cur + j;      // sorted position == event id for appended events
    key_s[r] = key; ts_s[r] = t; ev2[r] = make_int2(r, t); sid[r] = r; bev[r] = b;
    posn_s[3 * r] = px; posn_s[3 * r + 1] = py; posn_s[3 * r + 2] = pt;
    {
        // pool1 voxel slot from the float position, as grid_cluster computes it (pooling.py:55-56)
        const int S = nx * ny * (B + 1);
        const int cx = (int)__fdiv_rn(px, vsx), cy = (int)__fdiv_rn(py, vsy), ct = (int)pt;
        long long sl = (long long)cx + (long long)nx * cy + (long long)nx * ny * (b + ct);
        if (sl < 0 || sl >= S) { f |= EVAD_FLAG_POOL_WINDOW; sl = sl < 0 ? 0 : S - 1; }
        node_slot[r] = (int)sl;
        const int pl = (int)(sl / (nx * ny)), rm = (int)(sl % (nx * ny));
        node_vox[r] = (rm % nx) | ((rm / nx) << 8) | (pl << 16);
    }
    if (f) atomicOr(flags, f);
    int x0, y0; float fx, fy;
    evad_bilinear_setup(px, py, (float)W, (float)H, w_in, h_in, &x0, &y0, &fx, &fy);
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    const bool vx0 = x0 >= 0 && x0 < w_in, vx1 = x0 + 1 >= 0 && x0 + 1 < w_in;
    const bool vy0 = y0 >= 0 && y0 < h_in, vy1 = y0 + 1 >= 0 && y0 + 1 < h_in;
    const float4* m = (const float4*)(map0 + (size_t)b * h_in * w_in * 16);
    float4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](bool ok, int yy, int xx, float w) {
        if (!ok) return;
        const float4* p = m + ((size_t)yy * w_in + xx) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 v = p[c];
            acc[c].x += v.x * w; acc[c].y += v.y * w; acc[c].z += v.z * w; acc[c].w += v.w * w;
        }
    };
    add(vx0 && vy0, y0, x0, w00); add(vx1 && vy0, y0, x0 + 1, w01);
    add(vx0 && vy1, y0 + 1, x0, w10); add(vx1 && vy1, y0 + 1, x0 + 1, w11);
    float4* o = (float4*)(x0s + (size_t)r * 20);
    o[0] = make_float4(pol[j], acc[0].x, acc[0].y, acc[0].z);
    o[1] = make_float4(acc[0].w, acc[1].x, acc[1].y, acc[1].z);
    o[2] = make_float4(acc[1].w, acc[2].x, acc[2].y, acc[2].z);
    o[3] = make_float4(acc[2].w, acc[3].x, acc[3].y, acc[3].z);
    o[4] = make_float4(acc[3].w, px, py, 0.f);
    next[r - n_dense] = atomicExch(&head[key], r);           // per-pixel list of appended events (any order; the search sorts)
}

extern "C" int eventad_incr_insert(const float* pos_new, const float* polarity_new, const int64_t* batch_new, int k, int n_cur,
                                   int n_dense, int capacity, int B, int W, int H, int T, const float* map0_nhwc, int map_h,
                                   int map_w, float vsx, float vsy, int nx, int ny, int32_t* key_s, int32_t* ts_s, int32_t* ev2,
                                   int32_t* sid, int32_t* bev, float* posn_s, float* x0s, int32_t* node_slot, int32_t* node_vox,
                                   int32_t* pix_new_head, int32_t* new_next, uint32_t* counters, void* stream) {
    if (!pos_new || !polarity_new || !batch_new || !map0_nhwc || !key_s || !ts_s || !ev2 || !sid || !bev || !posn_s || !x0s ||
        !node_slot || !node_vox || !pix_new_head || !new_next || !counters || k < 0 || n_dense < 0 || n_cur < n_dense ||
        (long long)n_cur + k > capacity || B <= 0 || W <= 0 || H <= 0 || T <= 0 || nx <= 0 || ny <= 0 || nx > 255 || ny > 255 ||
        vsx <= 0 || vsy <= 0)
        return EVAD_ERR_ARG;
    if (k == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_incr_insert, dim3(evad_blocks(k, 256)), dim3(256), 0, (hipStream_t)stream, pos_new, polarity_new,
                       (const long long*)batch_new, k, n_cur, n_dense, B, W, H, T, map0_nhwc, map_h, map_w, vsx, vsy, nx, ny, key_s, ts_s,
                       (int2*)ev2, sid, bev, posn_s, x0s, node_slot, node_vox, pix_new_head, new_next, counters);
    return evad_last_error();
}

// The reference's walk (ev_graph.cu:15-80) for the appended events n_first .. n_first+k-1.  The FIFO of a pixel, as the
// reference's queue holds it after inserting the new events (ev_graph.cu:169-212): the appended events of the pixel,
// newest first, then the dense pass's segment, newest first; Q entries in all.
// One WAVE per new event (an update is a handful of events and what counts is its latency: one thread walking 729 spiral
// steps with two dependent loads each takes 0.27 ms): lane l takes spiral step i0 + l of a pass of 64 steps, walks its
// pixel's FIFO once to count the sources it contributes, the wave scans the counts (lane order = spiral order), and the
// lane walks again to put its sources -- newest first -- at its offset of the row, until 16 slots are taken.
struct IncrFifo { int cand[INCR_NEWCAP]; int nc, n_new, st, en; bool overflow; };

__device__ __forceinline__ void incr_fifo(const int* __restrict__ head, const int* __restrict__ next, const unsigned* __restrict__ start,
                                          int p, int n_dense, IncrFifo& f) {
    f.nc = 0; f.n_new = 0; f.overflow = false;
    for (int e = head[p]; e >= 0; e = next[e - n_dense]) {                // appended events of this pixel, sorted newest first
        ++f.n_new;
        if (f.nc == INCR_NEWCAP) {                                        // keep the newest INCR_NEWCAP; the flag reports the loss
            f.overflow = true;
            if (e < f.cand[f.nc - 1]) continue;
            --f.nc;
        }
        int q = f.nc++;
        while (q > 0 && f.cand[q - 1] < e) { f.cand[q] = f.cand[q - 1]; --q; }
        f.cand[q] = e;
    }
    f.st = (int)start[p]; f.en = (int)start[p + 1];
}

// sources of event (me, ts) in this FIFO, newest first; WRITE: into row[at ..] (slots < 16 only); -> their number (<= 16)
template <bool WRITE>
__device__ __forceinline__ int incr_walk(const int2* __restrict__ ev2, const IncrFifo& f, int me, int ts, int Q, int dt_max, bool monotone,
                                         int* row, unsigned short* srow, int at, int step) {
    const int depth = min(Q, f.n_new + f.en - f.st);
    int seen = 0, n = 0;
    bool stop = false;
    for (int c = 0; c < f.nc && seen < depth && n < 16 && !stop; ++c, ++seen) {
        const int2 o = ev2[f.cand[c]];
        if (me > o.x) {
            if ((float)(ts - o.y) > (float)dt_max) { if (monotone) stop = true; continue; }
            if (WRITE && at + n < 16) { row[at + n] = f.cand[c]; srow[at + n] = (unsigned short)step; }
            ++n;
        }
    }
    seen = max(seen, min(f.n_new, depth));                                // (entries beyond the collected ones still take FIFO room)
    for (int s = f.en - 1; s >= f.st && seen < depth && n < 16 && !stop; --s, ++seen) {
        const int2 o = ev2[s];
        if (me > o.x) {
            if ((float)(ts - o.y) > (float)dt_max) { if (monotone) stop = true; continue; }
            if (WRITE && at + n < 16) { row[at + n] = s; srow[at + n] = (unsigned short)step; }
            ++n;
        }
    }
    return n;
}

__global__ __launch_bounds__(64) void k_incr_edges(const int2* __restrict__ ev2, const int* __restrict__ key_s, const unsigned* __restrict__ start,
                             const int* __restrict__ head, const int* __restrict__ next, const int16_t* __restrict__ spiral,
                             unsigned* __restrict__ flags, int n_first, int k, int n_dense, int W, int H, int steps, int Q, int dt_max,
                             int* __restrict__ ell_src, uint16_t* __restrict__ ell_step, int* __restrict__ deg) {
    constexpr int MAXNB = 16;
    __shared__ int s_row[MAXNB];
    __shared__ unsigned short s_stp[MAXNB];
    const int lane = threadIdx.x, r = n_first + (int)blockIdx.x;
    if ((int)blockIdx.x >= k) return;
    const bool monotone = !(flags[0] & EVAD_FLAG_T_NOT_MONOTONE);
    const int key = key_s[r];
    const int2 mine = ev2[r];
    const int me = mine.x, ts = mine.y;
    const int x = key % W, y = (key / W) % H, plane_base = key - x - W * y;
    if (lane < MAXNB) { s_row[lane] = lane == 0 ? r : 0; s_stp[lane] = 0; }
    __syncthreads();
    int nn = 1;                                              // (uniform: every lane sees the same scan totals)
    bool overflow = false;
    for (int i0 = 0; i0 < steps && nn < MAXNB; i0 += 64) {
        const int i = i0 + lane;
        IncrFifo f;
        f.nc = 0; f.n_new = 0; f.st = 0; f.en = 0; f.overflow = false;
        if (i < steps) {
            const int xn = x + spiral[2 * i], yn = y + spiral[2 * i + 1];
            if (xn >= 0 && yn >= 0 && xn < W && yn < H) incr_fifo(head, next, start, plane_base + xn + W * yn, n_dense, f);
        }
        overflow |= f.overflow;
        const int nq = incr_walk<false>(ev2, f, me, ts, Q, dt_max, monotone, nullptr, nullptr, 0, i);
        int inc = nq;                                        // inclusive sum over the lanes: lane order = spiral order
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        const int at = nn + inc - nq;
        if (nq > 0 && at < MAXNB) incr_walk<true>(ev2, f, me, ts, Q, dt_max, monotone, s_row, s_stp, at, i);
        nn = min(MAXNB, nn + __shfl(inc, 63, 64));
        __syncthreads();
    }
    if (lane < MAXNB) {
        ell_src[(size_t)r * MAXNB + lane] = s_row[lane];
        ell_step[(size_t)r * MAXNB + lane] = s_stp[lane];
    }
    if (lane == 0) deg[r] = nn;
    if (__any(overflow) && lane == 0) atomicOr(flags, EVAD_FLAG_INCR_OVERFLOW);
}

extern "C" int eventad_incr_edges(const int32_t* ev2, const int32_t* key_s, const uint32_t* pix_start, const int32_t* pix_new_head,
                                  const int32_t* new_next, uint32_t* counters, int n_first, int k, int n_dense, int W, int H,
                                  int radius_px, int delta_t_us, int max_num_neighbors, int queue_size, int32_t* ell_src,
                                  uint16_t* ell_step, int32_t* deg, void* stream) {
    if (!ev2 || !key_s || !pix_start || !pix_new_head || !new_next || !counters || !ell_src || !ell_step || !deg || k < 0 ||
        n_dense < 0 || n_first < n_dense || W <= 0 || H <= 0 || radius_px < 0 || max_num_neighbors != 16 || queue_size < 1)
        return EVAD_ERR_ARG;
    if (k == 0) return EVAD_OK;
    int steps; const int16_t* table = nullptr;
    if (evad_spiral_table(radius_px, &steps, &table)) return -2;
    if (steps > 65535) return EVAD_ERR_ARG;
    hipLaunchKernelGGL(k_incr_edges, dim3((unsigned)k), dim3(64), 0, (hipStream_t)stream, (const int2*)ev2, key_s, pix_start,
                       pix_new_head, new_next, table, counters, n_first, k, n_dense, W, H, steps, queue_size, delta_t_us, ell_src,
                       ell_step, deg);
    return evad_last_error();
}

// max(*a, v) for floats with integer atomics (no float atomics anywhere on this path): a non-negative float orders like
// its bits as a signed integer, a negative one in reverse as an unsigned integer.  v + 0 turns -0 into +0.
__device__ __forceinline__ void incr_atomic_max(float* a, float v) {
    v = v + 0.f;
    if (v >= 0.f) atomicMax((int*)a, __float_as_int(v));
    else          atomicMin((unsigned*)a, __float_as_uint(v));
}

// 16 lanes per appended event: lane q samples channels 4q..4q+3 of CNN map 1 at the event (the dense pass's arithmetic,
// pool.hip: k_pool_seg0) and holds channel q of its layer-1 output; lane q >= 1 marks ELL slot q's pooled edge.
__global__ __launch_bounds__(256) void k_incr_pool0(const float* __restrict__ y2, const float* __restrict__ map, int map_h, int map_w,
                                                     const float* __restrict__ posn_s, const int* __restrict__ key_s,
                                                     const int* __restrict__ node_slot, const int* __restrict__ node_vox,
                                                     const int* __restrict__ ell_src, const int* __restrict__ deg, int n_first, int k,
                                                     int W, int H, float* __restrict__ feat, unsigned long long* __restrict__ sums,
                                                     unsigned long long* __restrict__ tsum, unsigned* __restrict__ cnt,
                                                     int* __restrict__ maxidx, unsigned* __restrict__ bitmap, unsigned* __restrict__ flags) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, j = t >> 4, q = t & 15;
    if (j >= k) return;
    const int r = n_first + j;
    const int slot = node_slot[r], key = key_s[r];
    const float px = posn_s[3 * r], py = posn_s[3 * r + 1], pt = posn_s[3 * r + 2];
    const int b = key / (W * H);
    int x0, y0; float fx, fy;
    evad_bilinear_setup(px, py, (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
    const bool vx0 = x0 >= 0 && x0 < map_w, vx1 = x0 + 1 >= 0 && x0 + 1 < map_w;
    const bool vy0 = y0 >= 0 && y0 < map_h, vy1 = y0 + 1 >= 0 && y0 + 1 < map_h;
    const bool m00 = vx0 && vy0, m01 = vx1 && vy0, m10 = vx0 && vy1, m11 = vx1 && vy1;
    const float* p00 = map + (size_t)(((b * map_h + y0) * map_w + x0) * 64) + 4 * q;
    const float4 c00 = *(const float4*)(m00 ? p00 : map);
    const float4 c01 = *(const float4*)(m01 ? p00 + 64 : map);
    const float4 c10 = *(const float4*)(m10 ? p00 + (size_t)map_w * 64 : map);
    const float4 c11 = *(const float4*)(m11 ? p00 + (size_t)map_w * 64 + 64 : map);
    const float w00 = m00 ? (1.f - fx) * (1.f - fy) : 0.f, w01 = m01 ? fx * (1.f - fy) : 0.f;
    const float w10 = m10 ? (1.f - fx) * fy : 0.f, w11 = m11 ? fx * fy : 0.f;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    v.x += c00.x * w00; v.y += c00.y * w00; v.z += c00.z * w00; v.w += c00.w * w00;
    v.x += c01.x * w01; v.y += c01.y * w01; v.z += c01.z * w01; v.w += c01.w * w01;
    v.x += c10.x * w10; v.y += c10.y * w10; v.z += c10.z * w10; v.w += c10.w * w10;
    v.x += c11.x * w11; v.y += c11.y * w11; v.z += c11.z * w11; v.w += c11.w * w11;
    float* f = feat + (size_t)slot * 80;
    incr_atomic_max(f + q, y2[(size_t)r * 16 + q]);
    incr_atomic_max(f + 16 + 4 * q, v.x); incr_atomic_max(f + 17 + 4 * q, v.y);
    incr_atomic_max(f + 18 + 4 * q, v.z); incr_atomic_max(f + 19 + 4 * q, v.w);
    if (q == 0) {
        atomicAdd(cnt + slot, 1u);
        atomicAdd(sums + 2 * (size_t)slot, (unsigned long long)(key % W));
        atomicAdd(sums + 2 * (size_t)slot + 1, (unsigned long long)((key / W) % H));
        atomicAdd(tsum + slot, (unsigned long long)llrintf(pt * 1099511627776.f));
        atomicMax(maxidx + slot, r + 1);                     // (event id + 1; id == sorted position for appended events)
    } else if (q < deg[r]) {
        const int vs = node_vox[ell_src[(size_t)r * 16 + q]], vd = node_vox[r];
        const int dx = (vs & 255) - (vd & 255), dy = ((vs >> 8) & 255) - ((vd >> 8) & 255), dp = (vs >> 16) - (vd >> 16);
        if (dx | dy | dp) {                                  // an edge inside one voxel is a self loop after pooling: dropped
            if (dx < -2 || dx > 2 || dy < -2 || dy > 2 || dp < -1 || dp > 1) atomicOr(flags, EVAD_FLAG_POOL_WINDOW);
            else {
                const int bit = ((dp + 1) * 5 + (dy + 2)) * 5 + (dx + 2);
                atomicOr(bitmap + (size_t)slot * 3 + (bit >> 5), 1u << (bit & 31));
            }
        }
    }
}

extern "C" int eventad_incr_pool0(const float* y2, const float* map_nhwc, int map_h, int map_w, const float* posn_s,
                                  const int32_t* key_s, const int32_t* node_slot, const int32_t* node_vox, const int32_t* ell_src,
                                  const int32_t* deg, int n_first, int k, int W, int H, float* feat, uint64_t* sums, double* tsum,
                                  uint32_t* cnt, int32_t* maxidx, uint32_t* bitmap, uint32_t* counters, void* stream) {
    if (!y2 || !map_nhwc || !posn_s || !key_s || !node_slot || !node_vox || !ell_src || !deg || !feat || !sums || !tsum || !cnt ||
        !maxidx || !bitmap || !counters || n_first < 0 || k < 0 || W <= 0 || H <= 0 || map_h <= 0 || map_w <= 0)
        return EVAD_ERR_ARG;
    if (k == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_incr_pool0, dim3(evad_blocks((long long)k * 16, 256)), dim3(256), 0, (hipStream_t)stream, y2, map_nhwc, map_h,
                       map_w, posn_s, key_s, node_slot, node_vox, ell_src, deg, n_first, k, W, H, feat, (unsigned long long*)sums,
                       (unsigned long long*)tsum, cnt, maxidx, bitmap, counters);
    return evad_last_error();
}

// pixel -> newest appended event (-1: none): filled once per dense pass
__global__ void k_incr_fill(int* __restrict__ p, long long n, int v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int eventad_incr_reset(int32_t* pix_new_head, int64_t pixels, void* stream) {
    if (!pix_new_head || pixels < 0) return EVAD_ERR_ARG;
    if (pixels == 0) return EVAD_OK;
    unsigned blocks = evad_blocks(pixels, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_incr_fill, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pix_new_head, (long long)pixels, -1);
    return evad_last_error();
}
