// eventad_amd/csrc/pool.hip -- voxel-grid pooling (reference src/dagr/model/layers/pooling.py:51-97)
// fused with the image-feature sampling that precedes it (sampling_skip, net.py:15-17,200-228).
//
// Reference: cluster = grid_cluster([x,y,t,b]) -> sorted unique relabel -> scatter_max / mean of
// x, mean of pos (floored to the pixel grid), batch[perm], cluster[edge_index] minus self loops,
// unique(dim=-1).  Natively the cluster id IS an index into a dense voxel-slot table
//      slot = cx + nx*cy + nx*ny*(batch + tcell),   S = nx*ny*(B+1) slots
// (tcell = (int)pos_t is 1 only for events at t == time_window: the reference's "last event lands
// in the next clip's voxel" quirk of pooling.py:51-58, reproduced here), so:
//   pool_scatter   one pass over the nodes: sample the CNN map, atomic max (order-preserving uint
//                  encoding) or float add into the slot's feature row, integer pixel sums, count,
//                  max member index (= the reference's perm on CPU);
//   pool_finalize  occupancy scan -> consecutive node ids in ascending slot order (= sorted unique),
//                  decode features, mean position -> pixel grid, batch, next layer's input row
//                  [features | pos_x | pos_y];
//   pool_edges_*   every input edge sets one bit in its destination slot's 5x5x3 neighbour-window
//                  bitmap (dedup for free); emit walks the bits in ascending slot order, which is
//                  ascending source id = the reference's unique(dim=-1) order within a destination.
#include "common.h"

// grid_cluster (torch_cluster): c_d = (int64)((pos_d - 0) / size_d); slot = cx + nx*cy + nx*ny*(clip + t-cell), clamped into the table
__device__ __forceinline__ int pool_slot_of(float px, float py, float pt, int b, float vsx, float vsy, int nx, int ny, int S, bool* bad) {
    int cx = (int)__fdiv_rn(px, vsx), cy = (int)__fdiv_rn(py, vsy), ct = (int)pt;
    long long sl = (long long)cx + (long long)nx * cy + (long long)nx * ny * (b + ct);
    *bad = sl < 0 || sl >= S;
    return (int)(sl < 0 ? 0 : (sl >= S ? S - 1 : sl));
}

// 16 lanes per node, 16 nodes per 256-thread block.  Max aggregation through order-preserving integer atomicMax (exact and
// order independent); the mean aggregation of pool4 only takes the slot bookkeeping from here (Ca = Cm = 0) and sums in
// k_pool_mean_gather.
__global__ __launch_bounds__(256) void k_pool_scatter(
    const float* __restrict__ xa, int lda, int Ca, const float* __restrict__ map, int map_h, int map_w, int Cm,
    const float* __restrict__ npos, const int* __restrict__ npix, const int* __restrict__ nbatch,
    const int* __restrict__ key_s, const int* __restrict__ nidx, int n_max, const int* __restrict__ n_dev,
    int W, int H, float vsx, float vsy, int nx, int ny, int S,
    unsigned* __restrict__ feat, unsigned long long* __restrict__ sums, double* __restrict__ tsum,
    unsigned* __restrict__ cnt, int* __restrict__ maxidx, int* __restrict__ node_slot, unsigned* __restrict__ flags,
    const unsigned* __restrict__ rowptr, const int* __restrict__ csr_src, unsigned* __restrict__ bitmap) {
    const int n = n_dev ? min(*n_dev, n_max) : n_max;
    const int node = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int lane = threadIdx.x & 15;
    if (node >= n) return;
    const float px = npos[3 * node], py = npos[3 * node + 1], pt = npos[3 * node + 2];
    int b, ipx, ipy;
    if (key_s) { int k = key_s[node]; ipx = k % W; ipy = (k / W) % H; b = k / (W * H); }
    else       { ipx = npix[2 * node]; ipy = npix[2 * node + 1]; b = nbatch[node]; }
    bool bad;
    const int slot = pool_slot_of(px, py, pt, b, vsx, vsy, nx, ny, S, &bad);
    if (bad && lane == 0) atomicOr(flags, EVAD_FLAG_POOL_WINDOW);
    const int C = Ca + Cm;
    unsigned* frow = feat + (size_t)slot * C;
    for (int c = lane; c < Ca; c += 16) {
        float v = xa[(size_t)node * lda + c];
        atomicMax(frow + c, evad_f2ord(v));
    }
    if (Cm > 0) {
        int x0, y0; float fx, fy;
        evad_bilinear_setup(px, py, (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
        float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
        bool vx0 = x0 >= 0 && x0 < map_w, vx1 = x0 + 1 >= 0 && x0 + 1 < map_w;
        bool vy0 = y0 >= 0 && y0 < map_h, vy1 = y0 + 1 >= 0 && y0 + 1 < map_h;
        const float* m = map + (size_t)b * map_h * map_w * Cm;
        const float* p00 = m + ((size_t)y0 * map_w + x0) * Cm;
        const float* p01 = p00 + Cm;
        const float* p10 = p00 + (size_t)map_w * Cm;
        const float* p11 = p10 + Cm;
        for (int c = lane; c < Cm; c += 16) {
            float v = 0.f;
            if (vx0 && vy0) v += p00[c] * w00;
            if (vx1 && vy0) v += p01[c] * w01;
            if (vx0 && vy1) v += p10[c] * w10;
            if (vx1 && vy1) v += p11[c] * w11;
            atomicMax(frow + Ca + c, evad_f2ord(v));
        }
    }
    if (lane == 0) {
        atomicAdd(&sums[2 * (size_t)slot], (unsigned long long)ipx);
        atomicAdd(&sums[2 * (size_t)slot + 1], (unsigned long long)ipy);
        // normalised time summed in 2^-40 fixed point: integer adds are order independent (bitwise reproducible)
        atomicAdd((unsigned long long*)tsum + slot, (unsigned long long)llrintf(pt * 1099511627776.f));
        atomicAdd(&cnt[slot], 1u);
        atomicMax(&maxidx[slot], (nidx ? nidx[node] : node) + 1);       // stored +1: a zero-filled table means 'none'
        node_slot[node] = slot;
    }
    // pooled-edge marks of this node's incoming edges (pooling.py:59-64), fused here: the slot of a source is a pure
    // function of its position, so it is recomputed instead of waiting for that node's own node_slot write
    if (bitmap) {
        for (unsigned e = rowptr[node] + lane; e < rowptr[node + 1]; e += 16) {
            const int src = csr_src[e];
            bool sbad;
            const int ss = pool_slot_of(npos[3 * src], npos[3 * src + 1], npos[3 * src + 2], nbatch[src], vsx, vsy, nx, ny, S, &sbad);
            evad_mark_edge(ss, slot, nx, ny, bitmap, flags);
        }
    }
}

// --------------------------------------------------------------------------- second convolution of a Layer + the next pooling
// The Layer's output row y (evad_tail_blend_row: blend, BatchNorm, skip, ReLU) goes straight into the next level's voxel
// table: one wave per node, lane = channel -- y[c] and the sample of map channel c are max-ed into the slot row while the
// row is in registers, lane 0 keeps the slot statistics, the lanes share the node's in-edges for the pooled-edge marks.
// Same arithmetic and the same integer atomics as k_tail_blend followed by k_pool_scatter: same results, one launch and
// one round trip of y less per level.  (mean aggregation: bookkeeping only, k_pool_mean_gather follows.)
__global__ __launch_bounds__(256) void k_tail_blend_pool(
    const float* __restrict__ Z, int ldz, int root_col, const float* __restrict__ Zs, int ldzs, int skip_col,
    const unsigned* __restrict__ rowptr, const int* __restrict__ csr_src, const int* __restrict__ csr_dxy, int n_max,
    const int* __restrict__ n_dev, float div_x, float div_y, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ sscale, const float* __restrict__ sshift, float* __restrict__ out, int ldo,
    const float* __restrict__ map, int map_h, int map_w, const float* __restrict__ npos, const int* __restrict__ npix,
    const int* __restrict__ nbatch, int W, int H, float vsx, float vsy, int nx, int ny, int S, int mean,
    unsigned* __restrict__ feat, unsigned long long* __restrict__ sums, double* __restrict__ tsum, unsigned* __restrict__ cnt,
    int* __restrict__ maxidx, int* __restrict__ node_slot, unsigned* __restrict__ flags, unsigned* __restrict__ bitmap) {
    const int n = min(*n_dev, n_max);
    const int node = blockIdx.x * 4 + (threadIdx.x >> 6), c = threadIdx.x & 63;
    if (node >= n) return;
    const float y = evad_tail_blend_row(Z, ldz, root_col, Zs, ldzs, skip_col, rowptr, csr_src, csr_dxy, node, c, div_x, div_y,
                                        scale, shift, sscale, sshift);
    out[(size_t)node * ldo + c] = y;
    const float px = npos[3 * node], py = npos[3 * node + 1], pt = npos[3 * node + 2];
    const int b = nbatch[node];
    bool bad;
    const int slot = pool_slot_of(px, py, pt, b, vsx, vsy, nx, ny, S, &bad);
    if (bad && c == 0) atomicOr(flags, EVAD_FLAG_POOL_WINDOW);
    if (!mean) {
        unsigned* frow = feat + (size_t)slot * 128;
        atomicMax(frow + c, evad_f2ord(y));
        int x0, y0; float fx, fy;
        evad_bilinear_setup(px, py, (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
        const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
        const bool vx0 = x0 >= 0 && x0 < map_w, vx1 = x0 + 1 >= 0 && x0 + 1 < map_w;
        const bool vy0 = y0 >= 0 && y0 < map_h, vy1 = y0 + 1 >= 0 && y0 + 1 < map_h;
        const float* p00 = map + ((size_t)b * map_h * map_w + (size_t)y0 * map_w + x0) * 64 + c;
        float v = 0.f;                                       // the four terms in k_pool_scatter's order
        if (vx0 && vy0) v += p00[0] * w00;
        if (vx1 && vy0) v += p00[64] * w01;
        if (vx0 && vy1) v += p00[(size_t)map_w * 64] * w10;
        if (vx1 && vy1) v += p00[(size_t)map_w * 64 + 64] * w11;
        atomicMax(frow + 64 + c, evad_f2ord(v));
    }
    if (c == 0) {
        atomicAdd(&sums[2 * (size_t)slot], (unsigned long long)npix[2 * node]);
        atomicAdd(&sums[2 * (size_t)slot + 1], (unsigned long long)npix[2 * node + 1]);
        atomicAdd((unsigned long long*)tsum + slot, (unsigned long long)llrintf(pt * 1099511627776.f));
        atomicAdd(&cnt[slot], 1u);
        atomicMax(&maxidx[slot], node + 1);
        node_slot[node] = slot;
    }
    // Pooled-edge marks of the node's in-edges: all of them land in the three bitmap words of ONE slot, and atomics on one
    // word are served one after the other -- so the wave ORs its lanes' bits first and sends each word once (the bits are
    // evad_mark_edge's).
    unsigned bm0 = 0, bm1 = 0, bm2 = 0;
    bool far = false;
    const int plane = nx * ny;
    for (unsigned e = rowptr[node] + c; e < rowptr[node + 1]; e += 64) {
        const int src = csr_src[e];
        bool sbad;
        const int ss = pool_slot_of(npos[3 * src], npos[3 * src + 1], npos[3 * src + 2], nbatch[src], vsx, vsy, nx, ny, S, &sbad);
        if (ss == slot) continue;                            // self loop after clustering: dropped (pooling.py:62-64)
        const int dx = ss % nx - slot % nx, dy = (ss / nx) % ny - (slot / nx) % ny, dp = ss / plane - slot / plane;
        if (dx < -2 || dx > 2 || dy < -2 || dy > 2 || dp < -1 || dp > 1) { far = true; continue; }
        const int bit = ((dp + 1) * 5 + (dy + 2)) * 5 + (dx + 2);
        const unsigned mk = 1u << (bit & 31);
        if (bit < 32) bm0 |= mk; else if (bit < 64) bm1 |= mk; else bm2 |= mk;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { bm0 |= __shfl_xor(bm0, d, 64); bm1 |= __shfl_xor(bm1, d, 64); bm2 |= __shfl_xor(bm2, d, 64); }
    if (c < 3) {
        const unsigned w = c == 0 ? bm0 : (c == 1 ? bm1 : bm2);
        unsigned* dst = bitmap + (size_t)slot * 3 + c;
        if (w) atomicOr(dst, w);                             // (no read-back first: the wave has nothing left to wait for)
    }
    if (__any(far) && c == 0) atomicOr(flags, EVAD_FLAG_POOL_WINDOW);
}


extern "C" int eventad_tail_blend_pool(const float* Z, int ldz, int root_col, const float* Zskip, int ldzs, int skip_col,
                                       const uint32_t* rowptr, const int32_t* csr_src, const int32_t* csr_dxy, int n_max,
                                       const int32_t* n_dev, float div_x, float div_y, const float* bn_scale,
                                       const float* bn_shift, const float* skip_scale, const float* skip_shift, float* out,
                                       int ldo, const float* map_nhwc, int map_h, int map_w, const float* npos,
                                       const int32_t* npix, const int32_t* nbatch, int W, int H, float vsx, float vsy, int nx,
                                       int ny, int B, int mean_aggr, uint32_t* feat, uint64_t* sums, double* tsum,
                                       uint32_t* cnt, int32_t* maxidx, int32_t* node_slot, uint32_t* flags, uint32_t* bitmap,
                                       void* stream) {
    if (!Z || !rowptr || !csr_src || !csr_dxy || !n_dev || !bn_scale || !bn_shift || !out || n_max < 0 || ldo < 64 ||
        div_x <= 0 || div_y <= 0 || (Zskip && (!skip_scale || !skip_shift)) || !npos || !npix || !nbatch || !feat || !sums ||
        !tsum || !cnt || !maxidx || !node_slot || !flags || !bitmap || (!mean_aggr && !map_nhwc) || nx <= 0 || ny <= 0 ||
        B <= 0 || vsx <= 0 || vsy <= 0)
        return EVAD_ERR_ARG;
    if (n_max == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_tail_blend_pool, dim3(evad_blocks(n_max, 4)), dim3(256), 0, (hipStream_t)stream, Z, ldz, root_col, Zskip,
                       ldzs, skip_col, rowptr, csr_src, csr_dxy, n_max, n_dev, div_x, div_y, bn_scale, bn_shift, skip_scale,
                       skip_shift, out, ldo, map_nhwc, map_h, map_w, npos, npix, nbatch, W, H, vsx, vsy, nx, ny,
                       nx * ny * (B + 1), mean_aggr, feat, (unsigned long long*)sums, tsum, cnt, maxidx, node_slot, flags, bitmap);
    return evad_last_error();
}

// --------------------------------------------------------------------------- mean pooling without float atomics
// pool4 (mean aggregation, <= 140 input nodes per clip): one wave per output slot scans the input nodes
// in ascending order and adds the members' rows -- the summation order of the reference's scatter-mean
// on CPU (sequential index_add_), and bitwise reproducible run to run.  Lane c owns channels c and c+64.
__global__ __launch_bounds__(256) void k_pool_mean_gather(const float* __restrict__ xa, int lda, int Ca,
                                                          const float* __restrict__ map, int map_h, int map_w, int Cm,
                                                          const float* __restrict__ npos, const int* __restrict__ nbatch,
                                                          const int* __restrict__ node_slot, int n_max,
                                                          const int* __restrict__ n_dev, int W, int H, int S,
                                                          float* __restrict__ feat) {
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (slot >= S) return;
    const int n = min(*n_dev, n_max), C = Ca + Cm;
    float acc0 = 0.f, acc1 = 0.f;
    for (int base = 0; base < n; base += 64) {              // 64 candidates per step, members found by ballot
      unsigned long long members = __ballot(base + lane < n && node_slot[base + lane] == slot);
      while (members) {
        const int i = base + __ffsll((long long)members) - 1;   // ascending node order = the reference's summation order
        members &= members - 1;
        for (int c = lane, q = 0; c < C; c += 64, ++q) {
            float v;
            if (c < Ca) v = xa[(size_t)i * lda + c];
            else {
                int x0, y0; float fx, fy;
                evad_bilinear_setup(npos[3 * i], npos[3 * i + 1], (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
                float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
                bool vx0 = x0 >= 0 && x0 < map_w, vx1 = x0 + 1 >= 0 && x0 + 1 < map_w;
                bool vy0 = y0 >= 0 && y0 < map_h, vy1 = y0 + 1 >= 0 && y0 + 1 < map_h;
                const float* p00 = map + (((size_t)nbatch[i] * map_h + y0) * map_w + x0) * Cm + (c - Ca);
                v = 0.f;
                if (vx0 && vy0) v += p00[0] * w00;
                if (vx1 && vy0) v += p00[Cm] * w01;
                if (vx0 && vy1) v += p00[(size_t)map_w * Cm] * w10;
                if (vx1 && vy1) v += p00[(size_t)map_w * Cm + Cm] * w11;
            }
            if (q == 0) acc0 += v; else acc1 += v;
        }
      }
    }
    if (lane < C) feat[(size_t)slot * C + lane] = acc0;
    if (lane + 64 < C) feat[(size_t)slot * C + lane + 64] = acc1;
}

extern "C" int eventad_pool_scatter(const float* xa, int lda, int Ca, const float* map_nhwc, int map_h, int map_w, int Cm,
                                    const float* npos, const int32_t* npix, const int32_t* nbatch, const int32_t* key_s,
                                    const int32_t* nidx, int n_max, const int32_t* n_dev, int W, int H, float vsx,
                                    float vsy, int nx, int ny, int B, int mean_aggr, uint32_t* feat, uint64_t* sums,
                                    double* tsum, uint32_t* cnt, int32_t* maxidx, int32_t* node_slot, uint32_t* flags,
                                    int prezeroed, const uint32_t* rowptr, const int32_t* csr_src, uint32_t* bitmap,
                                    void* stream) {
    if (!xa || !npos || !feat || !sums || !tsum || !cnt || !maxidx || !node_slot || !flags || (!key_s && (!npix || !nbatch)) ||
        (Cm > 0 && !map_nhwc) || Ca < 0 || Cm < 0 || n_max < 0 || nx <= 0 || ny <= 0 || B <= 0 || vsx <= 0 || vsy <= 0 ||
        (bitmap && (!rowptr || !csr_src || key_s || !nbatch)))
        return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int S = nx * ny * (B + 1), C = Ca + Cm;
    // one memset when the caller laid the tables out back to back (feat | sums | tsum | cnt | maxidx), else five
    const char* f0 = (const char*)feat; const size_t nf = sizeof(uint32_t) * (size_t)S * C;
    if (prezeroed) {}                                        // the caller zeroed the tables (one memset for all levels)
    else if ((const char*)sums == f0 + nf && (const char*)tsum == (const char*)sums + 16 * (size_t)S &&
        (const char*)cnt == (const char*)tsum + 8 * (size_t)S && (const char*)maxidx == (const char*)cnt + 4 * (size_t)S)
        evad_zero_async(feat, nf + 32 * (size_t)S, st);
    else {
        evad_zero_async(feat, nf, st);
        evad_zero_async(sums, sizeof(uint64_t) * 2 * (size_t)S, st);
        evad_zero_async(tsum, sizeof(double) * (size_t)S, st);
        evad_zero_async(cnt, sizeof(uint32_t) * (size_t)S, st);
        evad_zero_async(maxidx, sizeof(int32_t) * (size_t)S, st);
    }
    if (n_max == 0) return evad_last_error();
    dim3 grid(evad_blocks(n_max, 16)), block(256);
    if (mean_aggr) {
        if (!n_dev || key_s || C > 128) return EVAD_ERR_ARG;                     // pooled levels only
        hipLaunchKernelGGL(k_pool_scatter, grid, block, 0, st, xa, lda, 0, map_nhwc, map_h, map_w, 0, npos, npix,
                           nbatch, key_s, nidx, n_max, n_dev, W, H, vsx, vsy, nx, ny, S, feat, (unsigned long long*)sums,
                           tsum, cnt, maxidx, node_slot, flags, rowptr, csr_src, bitmap);
        hipLaunchKernelGGL(k_pool_mean_gather, dim3(evad_blocks(S, 4)), dim3(256), 0, st, xa, lda, Ca, map_nhwc, map_h, map_w,
                           Cm, npos, nbatch, node_slot, n_max, n_dev, W, H, S, (float*)feat);
    }
    else
        hipLaunchKernelGGL(k_pool_scatter, grid, block, 0, st, xa, lda, Ca, map_nhwc, map_h, map_w, Cm, npos, npix,
                           nbatch, key_s, nidx, n_max, n_dev, W, H, vsx, vsy, nx, ny, S, feat, (unsigned long long*)sums,
                           tsum, cnt, maxidx, node_slot, flags, rowptr, csr_src, bitmap);
    return evad_last_error();
}



// The in-order mean of eventad_pool_scatter's mean path on its own, for a caller that has already filled node_slot and the
// slot bookkeeping (eventad_tail_blend_pool with mean_aggr = 1).
extern "C" int eventad_pool_mean_gather(const float* xa, int lda, int Ca, const float* map_nhwc, int map_h, int map_w, int Cm,
                                        const float* npos, const int32_t* nbatch, const int32_t* node_slot, int n_max,
                                        const int32_t* n_dev, int W, int H, int nx, int ny, int B, float* feat, void* stream) {
    if (!xa || !npos || !nbatch || !node_slot || !n_dev || !feat || Ca < 0 || Cm < 0 || Ca + Cm > 128 || Ca + Cm < 1 ||
        (Cm > 0 && !map_nhwc) || n_max < 0 || nx <= 0 || ny <= 0 || B <= 0)
        return EVAD_ERR_ARG;
    const int S = nx * ny * (B + 1);
    hipLaunchKernelGGL(k_pool_mean_gather, dim3(evad_blocks(S, 4)), dim3(256), 0, (hipStream_t)stream, xa, lda, Ca, map_nhwc, map_h,
                       map_w, Cm, npos, nbatch, node_slot, n_max, n_dev, W, H, S, feat);
    return evad_last_error();
}

// --------------------------------------------------------------------------- level 0 without atomics
// One wave per voxel slot, two phases per chunk of up to 64 candidate events:
//  (1) lane l takes candidate l of the slot's pixel rows (a voxel row is one contiguous range of the pixel-sorted
//      array; the lanes enumerate up to 8 row ranges at once through their prefix sums): it loads the candidate's
//      slot, position, pixel key and id, keeps the integer statistics, and computes the bilinear set-up ONCE per event;
//  (2) 4 groups of 16 lanes then walk the chunk 4 members at a time, taking a member's set-up from its lane by
//      ds_bpermute: lane q of a group reads sampled channels 4q..4q+3 (one float4 per bilinear corner) and channel q
//      of the layer-1 output.  All loads of an iteration are unconditional (a corner outside the map reads a valid
//      address with weight 0; adding c * 0 leaves the sum unchanged), so iterations overlap.
// A fixed shuffle tree merges the groups.  Max is exact and order independent; integer pixel sums too.
__global__ __launch_bounds__(256) void k_pool_seg0(
    const float* __restrict__ y2, const float* __restrict__ map, int map_h, int map_w, const float* __restrict__ posn_s,
    const int* __restrict__ key_s, const int* __restrict__ sid, const int* __restrict__ node_slot,
    const unsigned* __restrict__ pix_start, const int* __restrict__ vx_lo, const int* __restrict__ vy_lo,
    const int* __restrict__ ovf_list, unsigned* __restrict__ counters, int B, int W, int H, int nx, int ny, int S,
    float* __restrict__ feat, unsigned long long* __restrict__ sums, unsigned long long* __restrict__ tsum, unsigned* __restrict__ cnt,
    int* __restrict__ maxidx, const int* __restrict__ ell_src, const int* __restrict__ deg, const int* __restrict__ node_vox,
    unsigned* __restrict__ bitmap, unsigned* __restrict__ slot_vd) {
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= S) return;
    const int lane = threadIdx.x & 63, grp = lane >> 4, q = lane & 15;
    const int plane = slot / (nx * ny), rem = slot % (nx * ny), cy = rem / nx, cx = rem % nx;
    const float NEG = -__builtin_inff();
    float4 mx = make_float4(NEG, NEG, NEG, NEG);
    float my = NEG;
    unsigned n = 0, sx = 0, sy = 0; unsigned long long ts = 0; int mi = 0;         // per-lane partial statistics; mi = max member id + 1
    // pool1's neighbour bitmap (which voxels of the 5x5x3 window around this one send it an edge): this wave owns the
    // slot's three words, so the marks of all its members' in-edges are OR-ed in registers and stored once -- no atomics,
    // no pre-zeroed table.  Lane q of a group looks at ELL slot q of the group's member.
    unsigned bm0 = 0, bm1 = 0, bm2 = 0; bool bad = false;
    // one chunk: candidate r of this lane (have = the lane holds a candidate), count = candidates in the chunk
    auto chunk = [&](int r, bool have, int count) {
        int off = 0, cmask = 0; float fx = 0.f, fy = 0.f, px = 0.f, py = 0.f;
        if (have) {
            const int ns = node_slot[r];
            px = posn_s[3 * r]; py = posn_s[3 * r + 1];
            const float pt = posn_s[3 * r + 2];
            const int k = key_s[r], id = sid[r];
            if (ns == slot) {
                ++n; sx += (unsigned)(k % W); sy += (unsigned)((k / W) % H);
                ts += (unsigned long long)llrintf(pt * 1099511627776.f); mi = max(mi, id + 1);
                const int b = k / (W * H);
                int x0, y0;
                evad_bilinear_setup(px, py, (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
                const bool vx0 = x0 >= 0 && x0 < map_w, vx1 = x0 + 1 >= 0 && x0 + 1 < map_w;
                const bool vy0 = y0 >= 0 && y0 < map_h, vy1 = y0 + 1 >= 0 && y0 + 1 < map_h;
                cmask = 16 | (vx0 && vy0 ? 1 : 0) | (vx1 && vy0 ? 2 : 0) | (vx0 && vy1 ? 4 : 0) | (vx1 && vy1 ? 8 : 0);
                off = ((b * map_h + y0) * map_w + x0) * 64;   // float offset of the top-left corner (only used where valid)
            }
        }
        // the map sample of member m (a lane of this chunk; m < 0: none): channels 4q .. 4q+3 into the running maximum
        auto sample = [&](int m) {
            const int mm = max(m, 0);
            const int m_mask_l = __shfl(cmask, mm, 64), m_off = __shfl(off, mm, 64);     // (every lane takes part in the shuffles)
            const int m_mask = m >= 0 ? m_mask_l : 0;
            const float m_fx = __shfl(fx, mm, 64), m_fy = __shfl(fy, mm, 64);
            const bool ok = (m_mask & 16) != 0;
            const float* p00 = map + (ok ? m_off : 0) + 4 * q;
            const float4 c00 = *(const float4*)((m_mask & 1) ? p00 : map);
            const float4 c01 = *(const float4*)((m_mask & 2) ? p00 + 64 : map);
            const float4 c10 = *(const float4*)((m_mask & 4) ? p00 + (size_t)map_w * 64 : map);
            const float4 c11 = *(const float4*)((m_mask & 8) ? p00 + (size_t)map_w * 64 + 64 : map);
            const float w00 = (m_mask & 1) ? (1.f - m_fx) * (1.f - m_fy) : 0.f, w01 = (m_mask & 2) ? m_fx * (1.f - m_fy) : 0.f;
            const float w10 = (m_mask & 4) ? (1.f - m_fx) * m_fy : 0.f, w11 = (m_mask & 8) ? m_fx * m_fy : 0.f;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            v.x += c00.x * w00; v.y += c00.y * w00; v.z += c00.z * w00; v.w += c00.w * w00;
            v.x += c01.x * w01; v.y += c01.y * w01; v.z += c01.z * w01; v.w += c01.w * w01;
            v.x += c10.x * w10; v.y += c10.y * w10; v.z += c10.z * w10; v.w += c10.w * w10;
            v.x += c11.x * w11; v.y += c11.y * w11; v.z += c11.z * w11; v.w += c11.w * w11;
            if (ok) { mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w); }
        };
        // Members at the same position as the member in the lane before them (several events of one pixel: they are
        // neighbours in the pixel-sorted array) have the same sample, and a maximum does not care how often it sees a value:
        // at two or more events per pixel most samples are such repeats.  A chunk in which a third or more of the
        // members are repeats samples its distinct positions in a loop of its own; the others (the sparse case) keep
        // sample and row in one loop.
        const bool member = (cmask & 16) != 0;
        const float ppx = __shfl_up(px, 1, 64), ppy = __shfl_up(py, 1, 64);
        const int pmask = __shfl_up(cmask, 1, 64);
        const bool repeat = member && lane > 0 && (pmask & 16) && ppx == px && ppy == py;
        unsigned long long distinct = __ballot(member && !repeat);
        const bool split = 3 * __popcll(__ballot(repeat)) >= __popcll(__ballot(member)) && __ballot(repeat) != 0ull;   // a third or more
        if (split) {
            while (distinct) {
                int m = -1;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {             // group g4 takes the g4-th lowest lane of the set (scalar unit)
                    const int mk = distinct ? __ffsll((long long)distinct) - 1 : -1;
                    distinct &= distinct - 1ull;             // (clearing a bit of 0 leaves 0)
                    m = grp == g4 ? mk : m;
                }
                sample(m);
            }
        }
        const int iters = (count + 3) >> 2;
#pragma unroll 2
        for (int it = 0; it < iters; ++it) {
            const int m = it * 4 + grp;                      // < 64 whenever it < iters
            const int m_r = __shfl(r, m, 64), m_mask = __shfl(cmask, m, 64);
            const bool ok = (m_mask & 16) != 0;
            const int rr = ok ? m_r : 0;
            const float yv = y2[(size_t)rr * 16 + q];
            int e_src = 0, e_deg = 0;
            if (bitmap) { e_src = ell_src[(size_t)rr * 16 + q]; e_deg = deg[rr]; }
            if (!split) sample(m);
            if (ok) my = fmaxf(my, yv);
            if (bitmap && ok && q >= 1 && q < e_deg) {
                const int vs = node_vox[e_src];
                const int dx = (vs & 255) - cx, dy = ((vs >> 8) & 255) - cy, dp = (vs >> 16) - plane;
                if (dx | dy | dp) {                          // an edge inside one voxel is a self loop after pooling: dropped
                    if (dx < -2 || dx > 2 || dy < -2 || dy > 2 || dp < -1 || dp > 1) bad = true;
                    else {
                        const int bit = ((dp + 1) * 5 + (dy + 2)) * 5 + (dx + 2);
                        const unsigned mk = 1u << (bit & 31);
                        if (bit < 32) bm0 |= mk; else if (bit < 64) bm1 |= mk; else bm2 |= mk;
                    }
                }
            }
        }
    };
    if (plane < B) {
        const int x0 = vx_lo[cx], x1 = vx_lo[cx + 1], y0 = vy_lo[cy], y1 = vy_lo[cy + 1];
        for (int yb = y0; yb < y1; yb += 8) {               // the CSR bounds of 8 pixel rows load together
            int ra[8], pre[9];
            pre[0] = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int a = 0, b = 0;
                if (yb + i < y1) {
                    const int base = W * (yb + i + H * plane);
                    a = (int)pix_start[base + x0]; b = (int)pix_start[base + x1];   // a voxel row is contiguous in sorted order
                }
                ra[i] = a; pre[i + 1] = b - a;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) pre[i + 1] += pre[i];
            const int total = pre[8];
            for (int c0 = 0; c0 < total; c0 += 64) {
                const int idx = c0 + lane;
                int o = ra[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) o = idx >= pre[i] ? ra[i] - pre[i] : o;
                chunk(o + idx, idx < total, min(64, total - c0));
            }
        }
    }
    const int n_ovf = (int)counters[2];
    for (int c0 = 0; c0 < n_ovf; c0 += 64) {
        const bool have = c0 + lane < n_ovf;
        chunk(have ? ovf_list[c0 + lane] : 0, have, min(64, n_ovf - c0));
    }
#pragma unroll
    for (int d = 16; d <= 32; d <<= 1) {
        mx.x = fmaxf(mx.x, __shfl_xor(mx.x, d, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, d, 64));
        mx.z = fmaxf(mx.z, __shfl_xor(mx.z, d, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, d, 64));
        my = fmaxf(my, __shfl_xor(my, d, 64));
    }
    unsigned long long sx64 = sx, sy64 = sy;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        n += __shfl_xor(n, d, 64); sx64 += __shfl_xor(sx64, d, 64); sy64 += __shfl_xor(sy64, d, 64);
        ts += __shfl_xor(ts, d, 64); mi = max(mi, __shfl_xor(mi, d, 64));
    }
    if (bitmap) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { bm0 |= __shfl_xor(bm0, d, 64); bm1 |= __shfl_xor(bm1, d, 64); bm2 |= __shfl_xor(bm2, d, 64); }
        if (lane == 0) {
            unsigned* w = bitmap + (size_t)slot * 3; w[0] = bm0; w[1] = bm1; w[2] = bm2;
            // occupied << 31 | pooled in-edges of the slot: what the scans of the finalize stage add up
            if (slot_vd) slot_vd[slot] = n > 0 ? (0x80000000u | (unsigned)(__popc(bm0) + __popc(bm1) + __popc(bm2))) : 0u;
        }
        if (bad) atomicOr(counters, EVAD_FLAG_POOL_WINDOW);
    }
    if (grp == 0) {
        float* f = feat + (size_t)slot * 80;
        f[q] = my;
        *(float4*)(f + 16 + 4 * q) = mx;
        if (q == 0) { cnt[slot] = n; sums[2 * (size_t)slot] = sx64; sums[2 * (size_t)slot + 1] = sy64; tsum[slot] = ts; maxidx[slot] = mi; }
    }
}

extern "C" int eventad_pool_seg0(const float* y2, const float* map_nhwc, int map_h, int map_w, const float* posn_s,
                                 const int32_t* key_s, const int32_t* sid, const int32_t* node_slot,
                                 const uint32_t* pix_start, const int32_t* vx_lo, const int32_t* vy_lo,
                                 const int32_t* ovf_list, uint32_t* counters, int N, int B, int W, int H, int nx,
                                 int ny, float* feat, uint64_t* sums, double* tsum, uint32_t* cnt, int32_t* maxidx,
                                 const int32_t* ell_src, const int32_t* deg, const int32_t* node_vox, uint32_t* bitmap,
                                 uint32_t* slot_vd, void* stream) {
    if (!y2 || !map_nhwc || !posn_s || !key_s || !sid || !node_slot || !pix_start || !vx_lo || !vy_lo || !ovf_list ||
        !counters || !feat || !sums || !tsum || !cnt || !maxidx || N < 0 || B <= 0 || nx <= 0 || ny <= 0 ||
        (bitmap && (!ell_src || !deg || !node_vox || nx > 255 || ny > 255)) || (slot_vd && !bitmap))
        return EVAD_ERR_ARG;
    int S = nx * ny * (B + 1);
    hipLaunchKernelGGL(k_pool_seg0, dim3(evad_blocks(S, 4)), dim3(256), 0, (hipStream_t)stream, y2, map_nhwc, map_h, map_w,
                       posn_s, key_s, sid, node_slot, pix_start, vx_lo, vy_lo, ovf_list, counters, B, W, H, nx, ny, S, feat,
                       (unsigned long long*)sums, (unsigned long long*)tsum, cnt, maxidx, ell_src, deg, node_vox, bitmap, slot_vd);
    return evad_last_error();
}

// --------------------------------------------------------------------------- finalize
int evad_exclusive_scan(const unsigned* in, unsigned* out, unsigned* tile_sums, long long n, hipStream_t st);

// occupancy flags + exclusive scan in ONE single-workgroup launch (S <= ~100k slots: 4 launches -> 1)
__global__ __launch_bounds__(1024) void k_occ_scan_small(const unsigned* __restrict__ cnt, unsigned* __restrict__ out, int S) {
    constexpr int IT = 16;                                   // items per thread per pass: 16K slots per pass, loads in flight together
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < S; base += 1024 * IT) {
        unsigned v[IT], s = 0;
        int i0 = base + tid * IT;
#pragma unroll
        for (int k = 0; k < IT; ++k) { v[k] = (i0 + k < S && cnt[i0 + k] > 0) ? 1u : 0u; s += v[k]; }
        unsigned inc = s;
        for (int d = 1; d < 64; d <<= 1) { unsigned t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        unsigned wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) wbase += wsum[w]; tot += wsum[w]; }
        unsigned ex = carry_s + wbase + inc - s;
#pragma unroll
        for (int k = 0; k < IT; ++k) { if (i0 + k < S) out[i0 + k] = ex; ex += v[k]; }
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0) out[S] = carry_s;
}

__global__ void k_occupancy(const unsigned* __restrict__ cnt, unsigned* __restrict__ occ, int S) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) occ[s] = cnt[s] > 0 ? 1u : 0u;
}

// One launch for node attributes + features: thread (slot s, channel c) writes channel c of the slot's node; the
// c == 0 thread also writes the node's position / pixel / clip and the pos channels of x (two dependent small
// launches cost ~4.5 us each inside a replayed graph -- more than the work of a pooled level).
template <int MODE>      // 0: ordered-uint max, 1: float sum -> mean, 2: float max
__global__ void k_finalize(const unsigned* __restrict__ occ_scan, const unsigned* __restrict__ cnt,
                           const unsigned long long* __restrict__ sums, const unsigned long long* __restrict__ tsum,
                           const int* __restrict__ maxidx, const int* __restrict__ batch_of_index,
                           const unsigned* __restrict__ feat, int S, int level0, float Wf, float Hf, float whx, float why, int C,
                           int ldx, float* __restrict__ xout, float* __restrict__ npos, int* __restrict__ npix,
                           int* __restrict__ nbatch, int* __restrict__ node_slot_out, int* __restrict__ slot2node,
                           int* __restrict__ n_out) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) *n_out = (int)occ_scan[S];
    if (t >= (long long)S * C) return;
    int s = (int)(t / C), ch = (int)(t % C);
    unsigned c = cnt[s];
    if (c == 0) { if (ch == 0) slot2node[s] = -1; return; }
    int j = (int)occ_scan[s];
    float v = MODE == 1 ? __fdiv_rn(__uint_as_float(feat[t]), (float)c) : (MODE == 2 ? __uint_as_float(feat[t]) : evad_ord2f(feat[t]));
    xout[(size_t)j * ldx + ch] = v;
    if (ch != 0) return;
    slot2node[s] = j; node_slot_out[j] = s;
    // pool_pos: mean of the members' normalised positions.  Level 0: pos = px/W; pooled levels:
    // pos = q * (1/W) (round_to_pixel).  Integer pixel sums make the mean order independent.
    float sx = (float)sums[2 * (size_t)s], sy = (float)sums[2 * (size_t)s + 1];
    float fx = level0 ? __fdiv_rn(sx, Wf) : __fmul_rn(sx, whx);
    float fy = level0 ? __fdiv_rn(sy, Hf) : __fmul_rn(sy, why);
    float cf = (float)c;
    float mx = __fdiv_rn(fx, cf), my = __fdiv_rn(fy, cf);
    float mt = (float)(((double)tsum[s] / 1099511627776.0) / (double)c);
    // round_to_pixel (pooling.py:47-49): floor((p + 1e-5) / wh_inv) * wh_inv
    float qx = floorf(__fdiv_rn(__fadd_rn(mx, 1e-5f), whx)), qy = floorf(__fdiv_rn(__fadd_rn(my, 1e-5f), why));
    float ox = __fmul_rn(qx, whx), oy = __fmul_rn(qy, why);
    npos[3 * j] = ox; npos[3 * j + 1] = oy; npos[3 * j + 2] = mt;
    npix[2 * j] = (int)qx; npix[2 * j + 1] = (int)qy;
    nbatch[j] = batch_of_index[maxidx[s] - 1];
    xout[(size_t)j * ldx + C] = ox; xout[(size_t)j * ldx + C + 1] = oy;
    for (int k = C + 2; k < ldx; ++k) xout[(size_t)j * ldx + k] = 0.f;
}

extern "C" int eventad_pool_finalize(const uint32_t* feat, const uint64_t* sums, const double* tsum, const uint32_t* cnt,
                                     const int32_t* maxidx, const int32_t* batch_of_index, int nx, int ny, int B, int C,
                                     int feat_mode, int level0, int W, int H, float wh_inv_x, float wh_inv_y,
                                     uint32_t* occ_scan, uint32_t* scan_tmp, float* xout, int ldx, float* npos_out,
                                     int32_t* npix_out, int32_t* nbatch_out, int32_t* node_slot_out, int32_t* slot2node,
                                     int32_t* n_out_dev, void* stream) {
    if (!feat || !sums || !tsum || !cnt || !maxidx || !batch_of_index || !occ_scan || !scan_tmp || !xout || !npos_out ||
        !npix_out || !nbatch_out || !node_slot_out || !slot2node || !n_out_dev || C < 1 || ldx < C + 2 || nx <= 0 || ny <= 0 || B <= 0)
        return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int S = nx * ny * (B + 1);
    if (S <= 131072) {
        hipLaunchKernelGGL(k_occ_scan_small, dim3(1), dim3(1024), 0, st, cnt, occ_scan, S);
    } else {
        hipLaunchKernelGGL(k_occupancy, dim3(evad_blocks(S, 256)), dim3(256), 0, st, cnt, occ_scan, S);
        int rc = evad_exclusive_scan(occ_scan, occ_scan, scan_tmp, S, st);
        if (rc) return rc;
    }
    dim3 grid(evad_blocks((long long)S * C, 256));
#define EVAD_FIN(MODE) hipLaunchKernelGGL(k_finalize<MODE>, grid, dim3(256), 0, st, occ_scan, cnt, (const unsigned long long*)sums, \
                       (const unsigned long long*)tsum, maxidx, batch_of_index, feat, S, level0, (float)W, (float)H, wh_inv_x, wh_inv_y, \
                       C, ldx, xout, npos_out, npix_out, nbatch_out, node_slot_out, slot2node, n_out_dev)
    if (feat_mode == 1) EVAD_FIN(1); else if (feat_mode == 2) EVAD_FIN(2); else EVAD_FIN(0);
#undef EVAD_FIN
    return evad_last_error();
}

// --------------------------------------------------------------------------- pooled edges
__global__ void k_mark_ell(const int* __restrict__ ell_src, const int* __restrict__ deg, const int* __restrict__ node_slot,
                           int N, int max_nb, int nx, int ny, unsigned* __restrict__ bitmap, unsigned* __restrict__ flags) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * max_nb) return;
    int r = (int)(t / max_nb), k = (int)(t % max_nb);
    if (k == 0 || k >= deg[r]) return;
    evad_mark_edge(node_slot[ell_src[t]], node_slot[r], nx, ny, bitmap, flags);
}

__global__ void k_mark_csr(const unsigned* __restrict__ rowptr, const int* __restrict__ csr_src,
                           const int* __restrict__ node_slot, int n_max, const int* __restrict__ n_dev, int nx, int ny,
                           unsigned* __restrict__ bitmap, unsigned* __restrict__ flags) {
    // one 16-lane group per destination node
    int n = min(*n_dev, n_max);
    int node = blockIdx.x * 16 + (threadIdx.x >> 4), lane = threadIdx.x & 15;
    if (node >= n) return;
    int sd = node_slot[node];
    for (unsigned e = rowptr[node] + lane; e < rowptr[node + 1]; e += 16) evad_mark_edge(node_slot[csr_src[e]], sd, nx, ny, bitmap, flags);
}

__global__ void k_edge_count(const unsigned* __restrict__ bitmap, const int* __restrict__ node_slot_out, int n_max,
                             const int* __restrict__ n_dev, unsigned* __restrict__ degree) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_max) return;
    int n = min(*n_dev, n_max);
    unsigned d = 0;
    if (j < n) { const unsigned* w = bitmap + (size_t)node_slot_out[j] * 3; d = __popc(w[0]) + __popc(w[1]) + __popc(w[2]); }
    degree[j] = d;
}

// per-destination edge counts + exclusive scan in one single-workgroup launch (small graphs)
__global__ __launch_bounds__(1024) void k_edge_count_scan_small(const unsigned* __restrict__ bitmap, const int* __restrict__ node_slot_out,
                                                                int n_max, const int* __restrict__ n_dev, unsigned* __restrict__ rowptr) {
    constexpr int IT = 8;                                    // nodes per thread per pass (their slot + 3 bitmap words load together)
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(*n_dev, n_max), total = n_max + 1;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < total; base += 1024 * IT) {
        const int j0 = base + tid * IT;
        unsigned d[IT], s = 0;
        int sl[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) sl[k] = j0 + k < n ? node_slot_out[j0 + k] : -1;
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            d[k] = 0;
            if (sl[k] >= 0) { const unsigned* w = bitmap + (size_t)sl[k] * 3; d[k] = __popc(w[0]) + __popc(w[1]) + __popc(w[2]); }
            s += d[k];
        }
        unsigned inc = s;
        for (int dd = 1; dd < 64; dd <<= 1) { unsigned t = __shfl_up(inc, dd, 64); if (lane >= dd) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        unsigned wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) wbase += wsum[w]; tot += wsum[w]; }
        unsigned ex = carry_s + wbase + inc - s;
#pragma unroll
        for (int k = 0; k < IT; ++k) { if (j0 + k < total) rowptr[j0 + k] = ex; ex += d[k]; }
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0) rowptr[total] = carry_s;
}

__global__ void k_edge_emit(const unsigned* __restrict__ bitmap, const int* __restrict__ node_slot_out,
                            const int* __restrict__ slot2node, const int* __restrict__ npix, int n_max,
                            const int* __restrict__ n_dev, int nx, int ny, const unsigned* __restrict__ rowptr,
                            int* __restrict__ csr_src, int* __restrict__ csr_dxy, int* __restrict__ e_out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int n = min(*n_dev, n_max);
    if (j == 0) *e_out = (int)rowptr[n];
    if (j >= n) return;
    int sd = node_slot_out[j];
    const unsigned* w = bitmap + (size_t)sd * 3;
    unsigned o = rowptr[j];
    int pxd = npix[2 * j], pyd = npix[2 * j + 1];
    for (int word = 0; word < 3; ++word) {
        unsigned bits = w[word];
        while (bits) {
            int bit = __ffs(bits) - 1 + word * 32;
            bits &= bits - 1;
            int dx = bit % 5 - 2, dy = (bit / 5) % 5 - 2, dp = bit / 25 - 1;
            int src = slot2node[sd + dx + nx * dy + nx * ny * dp];
            csr_src[o] = src;
            int ddx = npix[2 * src] - pxd, ddy = npix[2 * src + 1] - pyd;       // pos[src] - pos[dst] in pixels
            csr_dxy[o] = (ddx & 0xffff) | (ddy << 16);
            ++o;
        }
    }
}

// --------------------------------------------------------------------------- finalize + edges in two launches
// (the pooled levels hold 20160 / 5040 / 1260 / 315 slots at B = 8 -- every dependent launch costs more
// than its work.)  Launch 1, one workgroup: BOTH exclusive scans of a level in one pass over the slots --
// occupancy (= node numbering, slot order) and out-degree popcount(bitmap) of the occupied slots (= CSR row
// pointers; a set bit always names an occupied neighbour slot).  Launch 2: node attributes, features and the CSR
// columns together; the pixel of a neighbour node is recomputed from its slot sums by the same function that
// produces npix, so csr_dxy is bit-identical to the three-launch path.
__global__ __launch_bounds__(1024) void k_slot_scans_small(const unsigned* __restrict__ cnt, const unsigned* __restrict__ bitmap, int S,
                                                           int n_max, unsigned* __restrict__ occ_scan, unsigned* __restrict__ rowptr,
                                                           int* __restrict__ n_out, int* __restrict__ e_out) {
    // A pass covers IT*1024 consecutive slots.  Loads are coalesced (thread t takes slots t, t+1024, ...) and all in
    // flight together; the (occupied, degree) pairs go through LDS so that each thread then scans IT CONSECUTIVE
    // slots, and the occupancy prefixes return through LDS for a coalesced store.
    constexpr int IT = 20, CH = 1024 * IT;                   // level 0 at B = 8 (20160 slots) is one pass
    __shared__ unsigned char s_vd[CH];                       // occupied << 7 | degree (<= 75)
    __shared__ unsigned short s_pre[CH];                     // occupancy prefix inside the pass (< 20480)
    __shared__ unsigned wsum[16], wsum2[16];
    __shared__ unsigned carry_n, carry_e;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { carry_n = 0; carry_e = 0; }
    __syncthreads();
    for (int base = 0; base < S; base += CH) {
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int s = base + k * 1024 + tid;
            unsigned char vd = 0;
            if (s < S) {
                const unsigned c = cnt[s];
                const unsigned* w = bitmap + (size_t)s * 3;
                const unsigned pc = __popc(w[0]) + __popc(w[1]) + __popc(w[2]);
                vd = c > 0 ? (unsigned char)(0x80u | pc) : 0;
            }
            s_vd[k * 1024 + tid] = vd;
        }
        __syncthreads();
        unsigned v[IT], d[IT], sn = 0, se = 0;
        const int l0 = tid * IT;
#pragma unroll
        for (int k = 0; k < IT; ++k) { const unsigned x = s_vd[l0 + k]; v[k] = x >> 7; d[k] = x & 0x7fu; sn += v[k]; se += d[k]; }
        unsigned in_ = sn, ie = se;
        for (int dd = 1; dd < 64; dd <<= 1) {
            unsigned t = __shfl_up(in_, dd, 64), u = __shfl_up(ie, dd, 64);
            if (lane >= dd) { in_ += t; ie += u; }
        }
        if (lane == 63) { wsum[wv] = in_; wsum2[wv] = ie; }
        __syncthreads();
        unsigned bn = 0, be = 0, tn = 0, te = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) { bn += wsum[w]; be += wsum2[w]; } tn += wsum[w]; te += wsum2[w]; }
        unsigned en = bn + in_ - sn, ee = carry_e + be + ie - se;   // en: prefix inside the pass
        const unsigned cn = carry_n;
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            s_pre[l0 + k] = (unsigned short)en;
            if (v[k]) rowptr[cn + en] = ee;
            en += v[k]; ee += d[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int s = base + k * 1024 + tid;
            if (s < S) occ_scan[s] = cn + s_pre[k * 1024 + tid];
        }
        if (tid == 0) { carry_n += tn; carry_e += te; }
        __syncthreads();
    }
    const unsigned n = carry_n, E = carry_e;
    if (tid == 0) { occ_scan[S] = n; *n_out = (int)n; *e_out = (int)E; }
    for (int j = (int)n + tid; j <= n_max + 1; j += 1024) rowptr[j] = E;     // rows past the last node are empty
}

// The same two scans with one workgroup per tile of EVAD_SCAN_TILE slots, for a producer that leaves one word per slot
// (occupied << 31 | in-edges: eventad_pool_seg0's slot_vd): a workgroup adds up the words of all slots (those before its
// tile give its base), scans its own tile and writes its part of occ_scan / rowptr.  The single workgroup above takes
// 19 us for level 0 at C2 (20160 slots).
__global__ __launch_bounds__(EVAD_SCAN_TILE) void k_slot_scans_tiled(const unsigned* __restrict__ slot_vd, int S, int n_max,
                                                                     unsigned* __restrict__ occ_scan, unsigned* __restrict__ rowptr,
                                                                     int* __restrict__ n_out, int* __restrict__ e_out) {
    constexpr int NW = EVAD_SCAN_TILE / 64;
    __shared__ unsigned s_bn[NW], s_be[NW], s_an[NW], s_ae[NW], s_wn[NW], s_we[NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, tile = blockIdx.x, ntiles = gridDim.x;
    const int first = tile * EVAD_SCAN_TILE;
    unsigned bn = 0, be = 0, an = 0, ae = 0;                 // slots / edges before this tile, and of all tiles
    for (int u0 = 0; u0 < S; u0 += 8 * EVAD_SCAN_TILE) {    // 8 independent loads in flight per thread
        unsigned x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int u = u0 + k * EVAD_SCAN_TILE + tid; x[k] = u < S ? slot_vd[u] : 0u; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = u0 + k * EVAD_SCAN_TILE + tid;
            const unsigned o = x[k] >> 31, e = x[k] & 0x7fffffffu;
            an += o; ae += e;
            if (u < first) { bn += o; be += e; }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        bn += __shfl_xor(bn, d, 64); be += __shfl_xor(be, d, 64); an += __shfl_xor(an, d, 64); ae += __shfl_xor(ae, d, 64);
    }
    const int s = first + tid;
    const unsigned x = s < S ? slot_vd[s] : 0u, v = x >> 31, dg = x & 0x7fffffffu;
    unsigned in_ = v, ie = dg;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned a = __shfl_up(in_, d, 64), b = __shfl_up(ie, d, 64); if (lane >= d) { in_ += a; ie += b; } }
    if (lane == 0) { s_bn[wv] = bn; s_be[wv] = be; s_an[wv] = an; s_ae[wv] = ae; }
    if (lane == 63) { s_wn[wv] = in_; s_we[wv] = ie; }
    __syncthreads();
    unsigned base_n = 0, base_e = 0, n = 0, E = 0, wn = 0, we = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        base_n += s_bn[w]; base_e += s_be[w]; n += s_an[w]; E += s_ae[w];
        if (w < wv) { wn += s_wn[w]; we += s_we[w]; }
    }
    const unsigned en = base_n + wn + in_ - v, ee = base_e + we + ie - dg;
    if (s < S) { occ_scan[s] = en; if (v) rowptr[en] = ee; }
    if (tile == 0 && tid == 0) { occ_scan[S] = n; *n_out = (int)n; *e_out = (int)E; }
    for (int j = (int)n + first + tid; j <= n_max + 1; j += ntiles * EVAD_SCAN_TILE) rowptr[j] = E;   // rows past the last node are empty
}

struct SlotPix { int qx, qy; float ox, oy, mt; };
__device__ __forceinline__ SlotPix slot_pixel(const unsigned long long* __restrict__ sums, const unsigned long long* __restrict__ tsum,
                                              unsigned c, int s, int level0, float Wf, float Hf, float whx, float why, bool want_t) {
    // pool_pos: mean of the members' normalised positions (level 0: pos = px/W; pooled levels: pos = q * (1/W)), then
    // round_to_pixel (pooling.py:47-49): floor((p + 1e-5) / wh_inv) * wh_inv.  Same expressions as k_finalize.
    float sx = (float)sums[2 * (size_t)s], sy = (float)sums[2 * (size_t)s + 1];
    float fx = level0 ? __fdiv_rn(sx, Wf) : __fmul_rn(sx, whx);
    float fy = level0 ? __fdiv_rn(sy, Hf) : __fmul_rn(sy, why);
    float cf = (float)c;
    float mx = __fdiv_rn(fx, cf), my = __fdiv_rn(fy, cf);
    SlotPix r;
    r.mt = want_t ? (float)(((double)tsum[s] / 1099511627776.0) / (double)c) : 0.f;
    float qx = floorf(__fdiv_rn(__fadd_rn(mx, 1e-5f), whx)), qy = floorf(__fdiv_rn(__fadd_rn(my, 1e-5f), why));
    r.qx = (int)qx; r.qy = (int)qy; r.ox = __fmul_rn(qx, whx); r.oy = __fmul_rn(qy, why);
    return r;
}

template <int MODE>      // 0: ordered-uint max, 1: float sum -> mean, 2: float max
__global__ void k_finalize_emit(const unsigned* __restrict__ occ_scan, const unsigned* __restrict__ cnt,
                                const unsigned long long* __restrict__ sums, const unsigned long long* __restrict__ tsum,
                                const int* __restrict__ maxidx, const int* __restrict__ batch_of_index,
                                const unsigned* __restrict__ feat, const unsigned* __restrict__ bitmap,
                                const unsigned* __restrict__ rowptr, int S, int nx, int ny, int level0, float Wf, float Hf, float whx,
                                float why, int C, int ldx, float* __restrict__ xout, float* __restrict__ npos, int* __restrict__ npix,
                                int* __restrict__ nbatch, int* __restrict__ node_slot_out, int* __restrict__ slot2node,
                                int* __restrict__ csr_src, int* __restrict__ csr_dxy, int blocks_fin) {
    if ((int)blockIdx.x >= blocks_fin) {                     // CSR rows: one thread per slot, in their own blocks of the launch
        const int s = ((int)blockIdx.x - blocks_fin) * blockDim.x + threadIdx.x;   // (one emit lane per feature wave would
        if (s >= S) return;                                                      //  stall that whole wave behind it)
        const unsigned c = cnt[s];
        if (c == 0) return;
        const unsigned* w = bitmap + (size_t)s * 3;
        unsigned o = rowptr[occ_scan[s]];
        const SlotPix pd = slot_pixel(sums, tsum, c, s, level0, Wf, Hf, whx, why, false);
        for (int word = 0; word < 3; ++word) {
            unsigned bits = w[word];
            while (bits) {
                const int bit = __ffs(bits) - 1 + word * 32;
                bits &= bits - 1;
                const int dx = bit % 5 - 2, dy = (bit / 5) % 5 - 2, dp = bit / 25 - 1;
                const int ss = s + dx + nx * dy + nx * ny * dp;
                const SlotPix ps = slot_pixel(sums, tsum, cnt[ss], ss, level0, Wf, Hf, whx, why, false);
                csr_src[o] = (int)occ_scan[ss];                              // sources in bit order = slot order
                csr_dxy[o] = ((ps.qx - pd.qx) & 0xffff) | ((ps.qy - pd.qy) << 16);   // pos[src] - pos[dst] in pixels
                ++o;
            }
        }
        return;
    }
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)S * C) return;
    const int s = (int)(t / C), ch = (int)(t % C);
    const unsigned c = cnt[s];
    if (c == 0) { if (ch == 0) slot2node[s] = -1; return; }
    const int j = (int)occ_scan[s];
    float v = MODE == 1 ? __fdiv_rn(__uint_as_float(feat[t]), (float)c) : (MODE == 2 ? __uint_as_float(feat[t]) : evad_ord2f(feat[t]));
    xout[(size_t)j * ldx + ch] = v;
    if (ch == 0) {
        slot2node[s] = j; node_slot_out[j] = s;
        const SlotPix p = slot_pixel(sums, tsum, c, s, level0, Wf, Hf, whx, why, true);
        npos[3 * j] = p.ox; npos[3 * j + 1] = p.oy; npos[3 * j + 2] = p.mt;
        npix[2 * j] = p.qx; npix[2 * j + 1] = p.qy;
        nbatch[j] = batch_of_index[maxidx[s] - 1];
        xout[(size_t)j * ldx + C] = p.ox; xout[(size_t)j * ldx + C + 1] = p.oy;
        for (int k = C + 2; k < ldx; ++k) xout[(size_t)j * ldx + k] = 0.f;
    }
}

extern "C" int eventad_pool_finalize_edges(const uint32_t* feat, const uint64_t* sums, const double* tsum, const uint32_t* cnt,
                                           const int32_t* maxidx, const int32_t* batch_of_index, const uint32_t* bitmap, int nx,
                                           int ny, int B, int C, int feat_mode, int level0, int W, int H, float wh_inv_x,
                                           float wh_inv_y, uint32_t* occ_scan, float* xout, int ldx, float* npos_out,
                                           int32_t* npix_out, int32_t* nbatch_out, int32_t* node_slot_out, int32_t* slot2node,
                                           int32_t* n_out_dev, uint32_t* rowptr, int32_t* csr_src, int32_t* csr_dxy,
                                           int32_t* e_out_dev, const uint32_t* slot_vd, void* stream) {
    if (!feat || !sums || !tsum || !cnt || !maxidx || !batch_of_index || !bitmap || !occ_scan || !xout || !npos_out ||
        !npix_out || !nbatch_out || !node_slot_out || !slot2node || !n_out_dev || !rowptr || !csr_src || !csr_dxy || !e_out_dev ||
        C < 1 || ldx < C + 2 || nx <= 0 || ny <= 0 || B <= 0)
        return EVAD_ERR_ARG;
    const int S = nx * ny * (B + 1);
    if (S > 131072) return EVAD_ERR_ARG;                     // larger tables: eventad_pool_finalize + eventad_pool_edges_emit
    hipStream_t st = (hipStream_t)stream;
    if (slot_vd)
        hipLaunchKernelGGL(k_slot_scans_tiled, dim3(evad_blocks(S, EVAD_SCAN_TILE)), dim3(EVAD_SCAN_TILE), 0, st, slot_vd, S, S,
                           occ_scan, rowptr, n_out_dev, e_out_dev);
    else
        hipLaunchKernelGGL(k_slot_scans_small, dim3(1), dim3(1024), 0, st, cnt, bitmap, S, S, occ_scan, rowptr, n_out_dev, e_out_dev);
    const unsigned blocks_fin = evad_blocks((long long)S * C, 256);
    dim3 grid(blocks_fin + evad_blocks(S, 256));
#define EVAD_FE(MODE) hipLaunchKernelGGL(k_finalize_emit<MODE>, grid, dim3(256), 0, st, occ_scan, cnt, (const unsigned long long*)sums, \
                       (const unsigned long long*)tsum, maxidx, batch_of_index, feat, bitmap, rowptr, S, nx, ny, level0, (float)W, (float)H, \
                       wh_inv_x, wh_inv_y, C, ldx, xout, npos_out, npix_out, nbatch_out, node_slot_out, slot2node, csr_src, csr_dxy, (int)blocks_fin)
    if (feat_mode == 1) EVAD_FE(1); else if (feat_mode == 2) EVAD_FE(2); else EVAD_FE(0);
#undef EVAD_FE
    return evad_last_error();
}

extern "C" int eventad_pool_edges_mark_ell(const int32_t* ell_src, const int32_t* deg, const int32_t* node_slot, int N,
                                           int max_num_neighbors, int nx, int ny, int B, uint32_t* bitmap, uint32_t* flags,
                                           void* stream) {
    if (!ell_src || !deg || !node_slot || !bitmap || !flags || N < 0 || nx <= 0 || ny <= 0 || B <= 0) return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    evad_zero_async(bitmap, sizeof(uint32_t) * 3 * (size_t)nx * ny * (B + 1), st);
    if (N > 0)
        hipLaunchKernelGGL(k_mark_ell, dim3(evad_blocks((long long)N * max_num_neighbors, 256)), dim3(256), 0, st, ell_src,
                           deg, node_slot, N, max_num_neighbors, nx, ny, bitmap, flags);
    return evad_last_error();
}

extern "C" int eventad_pool_edges_mark_csr(const uint32_t* rowptr, const int32_t* csr_src, const int32_t* node_slot,
                                           int n_max, const int32_t* n_dev, int nx, int ny, int B, uint32_t* bitmap,
                                           uint32_t* flags, int prezeroed, void* stream) {
    if (!rowptr || !csr_src || !node_slot || !n_dev || !bitmap || !flags || n_max < 0 || nx <= 0 || ny <= 0 || B <= 0)
        return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!prezeroed) evad_zero_async(bitmap, sizeof(uint32_t) * 3 * (size_t)nx * ny * (B + 1), st);
    if (n_max > 0)
        hipLaunchKernelGGL(k_mark_csr, dim3(evad_blocks(n_max, 16)), dim3(256), 0, st, rowptr, csr_src, node_slot, n_max,
                           n_dev, nx, ny, bitmap, flags);
    return evad_last_error();
}

extern "C" int eventad_pool_edges_emit(const uint32_t* bitmap, const int32_t* node_slot_out, const int32_t* slot2node,
                                       const int32_t* npix_out, int n_max_out, const int32_t* n_out_dev, int nx, int ny,
                                       uint32_t* rowptr /* [n_max_out+2] */, uint32_t* scan_tmp, int32_t* csr_src,
                                       int32_t* csr_dxy, int32_t* e_out_dev, void* stream) {
    if (!bitmap || !node_slot_out || !slot2node || !npix_out || !n_out_dev || !rowptr || !scan_tmp || !csr_src ||
        !csr_dxy || !e_out_dev || n_max_out < 0 || nx <= 0 || ny <= 0)
        return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n_max_out <= 131072) {
        hipLaunchKernelGGL(k_edge_count_scan_small, dim3(1), dim3(1024), 0, st, bitmap, node_slot_out, n_max_out, n_out_dev, rowptr);
    } else {
        hipLaunchKernelGGL(k_edge_count, dim3(evad_blocks(n_max_out + 1, 256)), dim3(256), 0, st, bitmap, node_slot_out,
                           n_max_out, n_out_dev, rowptr);
        int rc = evad_exclusive_scan(rowptr, rowptr, scan_tmp, n_max_out + 1, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_edge_emit, dim3(evad_blocks(n_max_out > 0 ? n_max_out : 1, 256)), dim3(256), 0, st, bitmap,
                       node_slot_out, slot2node, npix_out, n_max_out, n_out_dev, nx, ny, rowptr, csr_src, csr_dxy, e_out_dev);
    return evad_last_error();
}
