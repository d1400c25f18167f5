// eventad_amd/csrc/graph.hip -- spatio-temporal radius-graph construction for gfx950.
//
// Two families of entry points (declared in include/eventad_hip.h):
//
//  (A) reference-compatible: eventad_insert_in_queue / _single / eventad_fill_edges operate on the
//      reference's own dense FIFO tensor `queue` i32[B,Q,H,W] and int64 edge buffer, so they are a
//      drop-in for the pybind module `ev_graph_cuda` (reference src/dagr/graph/ev_graph.cu:82-128,
//      215-276).  Same results bit for bit; launched on the caller's stream.
//
//  (B) native: eventad_sort_events + eventad_build_ell.  No dense FIFO (268 MB..3.8 GB of -1 fill
//      per batch in the reference): events are counting-sorted by pixel (CSR over pixels), which
//      *is* the FIFO content -- the FIFO of a pixel after one batch insert holds its newest Q
//      events newest-first (ev_graph.cu:201-211), i.e. the tail of the pixel's id-sorted segment.
//      The edge search (ev_graph.cu:15-80) then walks the same spiral over CSR segments and writes
//      an ELL adjacency (<=16 sources per destination, self edge first) in PIXEL-SORTED node
//      numbering, so the layer-1 neighbour gathers that follow hit nearby rows.
#include "common.h"
#include <mutex>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include <stdio.h>

// ------------------------------------------------------------------ spiral (reference spiral.h)
// Offsets are generated on the host once per radius (eventad_spiral_offsets) and passed as a table.
extern "C" int eventad_spiral_offsets(int radius, int16_t* dxdy /* [(2r+1)^2][2] host memory */) {
    if (radius < 0 || !dxdy) return EVAD_ERR_ARG;
    int steps = (2 * radius + 1) * (2 * radius + 1);
    int x = 0, y = 0, layer = 1, leg = 0;
    for (int i = 0; i < steps; ++i) {
        dxdy[2 * i] = (int16_t)x; dxdy[2 * i + 1] = (int16_t)y;
        if (leg == 0)      { ++x; if (x == layer) ++leg; }
        else if (leg == 1) { ++y; if (y == layer) ++leg; }
        else if (leg == 2) { --x; if (-x == layer) ++leg; }
        else               { --y; if (-y == layer) { leg = 0; ++layer; } }
    }
    return steps;
}

// =============================================================================================
// (A) reference-compatible kernels
// =============================================================================================
__global__ void k_insert_in_queue(const int32_t* __restrict__ indices, const int32_t* __restrict__ unique_coords,
                                  const int32_t* __restrict__ cumsum, int32_t* __restrict__ queue,
                                  int Q, int H, int W, int K) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    int offset = k > 0 ? cumsum[k - 1] : 0;
    int counts = cumsum[k] - offset;
    int c = unique_coords[k];
    int x = c % W, y = ((c - x) / W) % H, b = c / (W * H);
    int64_t plane = (int64_t)H * W;
    int32_t* col = queue + (int64_t)b * plane * Q + (int64_t)y * W + x;   // FIFO column of this pixel
    // shift the old content down by `counts`, newest new ids go to slots 0..counts-1
    for (int q = Q - 1; q >= 0; --q)
        col[q * plane] = (q >= counts) ? col[(q - counts) * plane] : indices[offset + counts - 1 - q];
}

__global__ void k_insert_in_queue_single(const int32_t* __restrict__ indices, const int32_t* __restrict__ events,
                                         int32_t* __restrict__ queue, int Q, int H, int W) {
    if (blockIdx.x * blockDim.x + threadIdx.x != 0) return;
    int x = events[0], y = events[1];
    int64_t plane = (int64_t)H * W;
    int32_t* col = queue + (int64_t)y * W + x;                              // clip index forced to 0
    for (int q = Q - 1; q >= 1; --q) col[q * plane] = col[(q - 1) * plane];
    col[0] = indices[0];
}

__global__ void k_fill_edges(const int32_t* __restrict__ batch, const int32_t* __restrict__ pos,
                             const int32_t* __restrict__ all_ts, const int32_t* __restrict__ indices,
                             const int32_t* __restrict__ queue, const int16_t* __restrict__ spiral,
                             int64_t* __restrict__ edges, int Q, int H, int W, int N, int64_t K,
                             int steps, float delta_t_us, int max_nb, int min_index) {
    int ev = blockIdx.x * blockDim.x + threadIdx.x;
    if (ev >= N) return;
    int b = batch[ev], x = pos[3 * ev], y = pos[3 * ev + 1], ts = pos[3 * ev + 2];
    int me = indices[ev];
    int64_t off = (int64_t)ev * max_nb;
    int nn = 0;
    edges[off] = me - min_index;  edges[off + K] = me - min_index;  nn = 1;       // self edge
    int64_t plane = (int64_t)H * W;
    for (int i = 0; i < steps && nn < max_nb; ++i) {
        int xn = x + spiral[2 * i], yn = y + spiral[2 * i + 1];
        if (xn < 0 || yn < 0 || xn >= W || yn >= H) continue;
        const int32_t* col = queue + (int64_t)b * plane * Q + (int64_t)yn * W + xn;
        for (int q = 0; q < Q; ++q) {
            int idx = col[q * plane];
            if (idx < min_index) break;                                   // FIFO exhausted
            if (me > idx) {
                int dt = ts - all_ts[idx - min_index];
                if ((float)dt > delta_t_us) continue;                     // too old: keeps scanning
                edges[off + nn] = idx - min_index;  edges[off + nn + K] = me - min_index;
                if (++nn >= max_nb) break;
            }
        }
    }
}

// Spiral tables on the device.  Radii 0..SPIRAL_MAX_R live back to back in constant memory (table of radius r starts at
// entry spiral_base(r)): written once per GPU, never changed afterwards, so models with different radii -- and graphs
// captured for them -- can coexist in one process.  The spiral step index is wave-uniform in the per-event walks, so reads
// through these symbols are scalar loads.  Larger radii (the generic kernels only) get a device table per (GPU, radius),
// kept until the process ends.
#define SPIRAL_MAX_R 13
#define SPIRAL_TOTAL 4495                    /* sum over r = 0..13 of (2r+1)^2 */
__constant__ int16_t c_spiral[2 * SPIRAL_TOTAL];
__constant__ int16_t c_stepof[SPIRAL_TOTAL];   // inverse: (dy+R)*D + (dx+R) -> spiral step
static inline int spiral_base(int r) { return r * (2 * r - 1) * (2 * r + 1) / 3; }     // sum over k < r of (2k+1)^2

#define EVAD_MAX_DEVICES 64
static std::mutex g_spiral_mutex;
static bool g_spiral_const_ready[EVAD_MAX_DEVICES];
// Ring bands of a spiral for the row-major wave-per-pixel search (k_build_ell_pixr): consecutive whole rings with at most
// BAND_CAP pixels per band; inside a band the pixels come in row-major order (dy, then dx), each as
// (dx + 128) | (dy + 128) << 8 | spiral step << 16.
#define BAND_CAP 256
#define BAND_MAX 8
struct BandPasses { int n; int off[BAND_MAX], cnt[BAND_MAX], step0[BAND_MAX]; };
struct SpiralDev { int dev, radius; int16_t* table; uint32_t* band; BandPasses passes; };
static std::vector<SpiralDev*> g_spiral_tables;

// Tables for `radius` on the current device; *entry_out (nullable) = the radius' own device tables (spiral [steps][2], ring
// bands).  The first call for a GPU (and for a new radius) allocates and copies synchronously: make it outside any stream
// capture (eventad_spiral_prepare); later calls only look the tables up.
static int ensure_spiral(int radius, int* steps_out, const SpiralDev** entry_out) {
    const int steps = (2 * radius + 1) * (2 * radius + 1);
    *steps_out = steps;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EVAD_MAX_DEVICES) return -2;
    std::lock_guard<std::mutex> lock(g_spiral_mutex);
    if (!g_spiral_const_ready[dev]) {
        std::vector<int16_t> all(2 * SPIRAL_TOTAL), inv(SPIRAL_TOTAL);
        for (int r = 0; r <= SPIRAL_MAX_R; ++r) {
            const int b0 = spiral_base(r), D = 2 * r + 1, n = D * D;
            int16_t* t = all.data() + 2 * b0;
            eventad_spiral_offsets(r, t);
            for (int i = 0; i < n; ++i) inv[b0 + (t[2 * i + 1] + r) * D + t[2 * i] + r] = (int16_t)i;
        }
        if (hipMemcpyToSymbol(HIP_SYMBOL(c_spiral), all.data(), sizeof(int16_t) * all.size()) != hipSuccess ||
            hipMemcpyToSymbol(HIP_SYMBOL(c_stepof), inv.data(), sizeof(int16_t) * inv.size()) != hipSuccess) {
            (void)hipGetLastError();
            return -2;
        }
        g_spiral_const_ready[dev] = true;
    }
    if (!entry_out) return 0;
    for (const SpiralDev* e : g_spiral_tables)
        if (e->dev == dev && e->radius == radius) { *entry_out = e; return 0; }
    std::vector<int16_t> host(2 * (size_t)steps);
    eventad_spiral_offsets(radius, host.data());
    SpiralDev* e = new SpiralDev{dev, radius, nullptr, nullptr, {}};
    if (hipMalloc(&e->table, sizeof(int16_t) * host.size()) != hipSuccess ||
        hipMemcpy(e->table, host.data(), sizeof(int16_t) * host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError(); delete e;
        return -2;
    }
    if (radius <= SPIRAL_MAX_R) {                            // ring bands (radius <= 13: at most 4)
        const int D = 2 * radius + 1;
        std::vector<int> stepof((size_t)D * D);
        for (int i = 0; i < steps; ++i) stepof[(size_t)(host[2 * i + 1] + radius) * D + host[2 * i] + radius] = i;
        std::vector<uint32_t> band;
        BandPasses& P = e->passes;
        P.n = 0;
        for (int r0 = 0; r0 <= radius && P.n < BAND_MAX;) {
            int r1 = r0;
            const int inner = r0 == 0 ? 0 : (2 * r0 - 1) * (2 * r0 - 1);
            while (r1 + 1 <= radius && (2 * r1 + 3) * (2 * r1 + 3) - inner <= BAND_CAP) ++r1;
            P.off[P.n] = (int)band.size(); P.step0[P.n] = inner;
            for (int dy = -r1; dy <= r1; ++dy)
                for (int dx = -r1; dx <= r1; ++dx) {
                    const int ring = std::max(std::abs(dx), std::abs(dy));
                    if (ring < r0) continue;
                    band.push_back((uint32_t)(dx + 128) | (uint32_t)(dy + 128) << 8 | (uint32_t)stepof[(size_t)(dy + radius) * D + dx + radius] << 16);
                }
            P.cnt[P.n] = (int)band.size() - P.off[P.n];
            if (P.cnt[P.n] > BAND_CAP) { P.n = 0; break; }     // (a single ring larger than a band: radius > 32, not reached)
            ++P.n;
            r0 = r1 + 1;
        }
        if (P.n > 0 && (hipMalloc(&e->band, sizeof(uint32_t) * band.size()) != hipSuccess ||
                        hipMemcpy(e->band, band.data(), sizeof(uint32_t) * band.size(), hipMemcpyHostToDevice) != hipSuccess)) {
            (void)hipGetLastError(); hipFree(e->table); delete e;
            return -2;
        }
    }
    g_spiral_tables.push_back(e);
    *entry_out = e;
    return 0;
}

// (for the other translation units of the library) device table [steps][2] of `radius` on the current GPU
int evad_spiral_table(int radius, int* steps, const int16_t** table) {
    const SpiralDev* e = nullptr;
    if (ensure_spiral(radius, steps, &e)) return -2;
    *table = e->table;
    return 0;
}

extern "C" int eventad_spiral_prepare(int radius) {
    if (radius < 0 || radius > 127) return EVAD_ERR_ARG;
    int steps; const SpiralDev* e = nullptr;
    return ensure_spiral(radius, &steps, &e);
}

extern "C" int eventad_insert_in_queue(const int32_t* indices, const int32_t* unique_coords,
                                       const int32_t* cumsum_counts, int32_t* queue,
                                       int B, int Q, int H, int W, int K, void* stream) {
    if (!indices || !unique_coords || !cumsum_counts || !queue || B <= 0 || Q <= 0 || H <= 0 || W <= 0 || K < 0)
        return EVAD_ERR_ARG;
    if (K == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_insert_in_queue, dim3(evad_blocks(K, 256)), dim3(256), 0, (hipStream_t)stream,
                       indices, unique_coords, cumsum_counts, queue, Q, H, W, K);
    return evad_last_error();
}

extern "C" int eventad_insert_in_queue_single(const int32_t* indices, const int32_t* events, int32_t* queue,
                                              int B, int Q, int H, int W, void* stream) {
    if (!indices || !events || !queue || B <= 0 || Q <= 0 || H <= 0 || W <= 0) return EVAD_ERR_ARG;
    hipLaunchKernelGGL(k_insert_in_queue_single, dim3(1), dim3(64), 0, (hipStream_t)stream, indices, events, queue, Q, H, W);
    return evad_last_error();
}

extern "C" int eventad_fill_edges(const int32_t* batch, const int32_t* pos, const int32_t* all_timestamps,
                                  const int32_t* event_queue, const int32_t* indices, int max_num_neighbors,
                                  float radius, float delta_t_us, int64_t* edges, int min_index,
                                  int B, int Q, int H, int W, int N, int64_t K, void* stream) {
    if (!batch || !pos || !all_timestamps || !event_queue || !indices || !edges || max_num_neighbors < 1 ||
        radius < 0 || B <= 0 || Q <= 0 || H <= 0 || W <= 0 || N < 0 || K < (int64_t)N * max_num_neighbors)
        return EVAD_ERR_ARG;
    if (N == 0) return EVAD_OK;
    int steps; const SpiralDev* sp = nullptr;
    if (ensure_spiral((int)radius, &steps, &sp)) return -2;
    hipLaunchKernelGGL(k_fill_edges, dim3(evad_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       batch, pos, all_timestamps, indices, event_queue, sp->table, edges, Q, H, W, N, K,
                       steps, delta_t_us, max_num_neighbors, min_index);
    return evad_last_error();
}

// =============================================================================================
// generic exclusive scan (u32), used for pixel CSR, voxel compaction and pooled-edge row pointers
// =============================================================================================
#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* total) {
    __shared__ unsigned wave_sums[SCAN_THREADS / 64];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        unsigned t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wave_sums[wv] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) { if (w < wv) base += wave_sums[w]; tot += wave_sums[w]; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// tile_max (nullable): also the largest input of each tile (reduced to one number by k_scan_sums)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(const unsigned* __restrict__ in, unsigned* __restrict__ out,
                                                             unsigned* __restrict__ tile_sums, long long n,
                                                             unsigned* __restrict__ tile_max) {
    __shared__ unsigned s_max;
    long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned v[SCAN_ITEMS], s = 0, m = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < n) ? in[base + k] : 0u; s += v[k]; m = max(m, v[k]); }
    if (tile_max) {
        if (threadIdx.x == 0) s_max = 0;
        __syncthreads();
        for (int d = 32; d > 0; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&s_max, m);
    }
    unsigned tot, ex = block_exclusive_scan(s, &tot);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
    if (threadIdx.x == 0) { tile_sums[blockIdx.x] = tot; if (tile_max) tile_max[blockIdx.x] = s_max; }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(unsigned* __restrict__ tile_sums, int n_tiles,
                                                            unsigned* __restrict__ grand_total,
                                                            const unsigned* __restrict__ tile_max, unsigned* __restrict__ max_out) {
    if (tile_max) {
        __shared__ unsigned s_max;
        if (threadIdx.x == 0) s_max = 0;
        __syncthreads();
        unsigned m = 0;
        for (int i = threadIdx.x; i < n_tiles; i += SCAN_THREADS) m = max(m, tile_max[i]);
        for (int d = 32; d > 0; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&s_max, m);
        __syncthreads();
        if (threadIdx.x == 0) *max_out = s_max;
    }
    unsigned carry = 0;
    for (int base = 0; base < n_tiles; base += SCAN_THREADS) {
        int i = base + threadIdx.x;
        unsigned v = i < n_tiles ? tile_sums[i] : 0u, tot;
        unsigned ex = block_exclusive_scan(v, &tot);
        if (i < n_tiles) tile_sums[i] = ex + carry;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}

__global__ void k_scan_add(unsigned* __restrict__ out, const unsigned* __restrict__ tile_sums, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += tile_sums[i / SCAN_TILE];
}

// Second and last launch of the scan when there are few tiles: the workgroup of tile t adds up the totals of the tiles
// before it itself (a few loads per thread) instead of waiting for a single-workgroup scan of the totals; the last
// workgroup also writes the grand total and the maximum.
#define SCAN_FUSED_MAX_TILES 8192
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_add_fused(unsigned* __restrict__ out, const unsigned* __restrict__ tile_sums,
                                                                 int n_tiles, long long n, const unsigned* __restrict__ tile_max,
                                                                 unsigned* __restrict__ max_out) {
    __shared__ unsigned s_sum[SCAN_THREADS / 64], s_max[SCAN_THREADS / 64];
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool last = tile == n_tiles - 1;
    const int upto = last ? n_tiles : tile;                  // the last workgroup needs the grand total as well
    unsigned acc = 0, own = 0, m = 0;
    for (int u = tid; u < upto; u += SCAN_THREADS) {
        const unsigned v = tile_sums[u];
        if (u < tile) acc += v; else own += v;
        if (last && tile_max) m = max(m, tile_max[u]);
    }
    unsigned both = acc + own;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        acc += __shfl_xor(acc, d, 64); both += __shfl_xor(both, d, 64); m = max(m, (unsigned)__shfl_xor((int)m, d, 64));
    }
    if (lane == 0) { s_sum[wv] = acc; s_max[wv] = m; }
    __syncthreads();
    unsigned base = 0, mm = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) { base += s_sum[w]; mm = max(mm, s_max[w]); }
    if (last) {
        __shared__ unsigned s_all[SCAN_THREADS / 64];
        if (lane == 0) s_all[wv] = both;
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int w = 0; w < SCAN_THREADS / 64; ++w) tot += s_all[w];
            out[n] = tot;
            if (max_out) *max_out = mm;
        }
    }
    if (tile == 0) return;                                   // base 0
    const long long i0 = (long long)tile * SCAN_TILE + (long long)tid * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (i0 + k < n) out[i0 + k] += base;
}

// out has n+1 entries (out[n] = grand total); tile_sums needs ceil(n/2048) entries.  in may alias out.
// max_out (nullable): also max(in); tile_sums then needs 2*ceil(n/2048) entries.
int evad_exclusive_scan_max(const unsigned* in, unsigned* out, unsigned* tile_sums, long long n, unsigned* max_out, hipStream_t st) {
    if (n <= 0) { evad_zero_async(out, sizeof(unsigned), st); return evad_last_error(); }
    int tiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    unsigned* tile_max = max_out ? tile_sums + tiles : nullptr;
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(SCAN_THREADS), 0, st, in, out, tile_sums, n, tile_max);
    if (tiles <= SCAN_FUSED_MAX_TILES) {
        hipLaunchKernelGGL(k_scan_add_fused, dim3(tiles), dim3(SCAN_THREADS), 0, st, out, tile_sums, tiles, n, tile_max, max_out);
        return evad_last_error();
    }
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, tile_sums, tiles, out + n, tile_max, max_out);
    if (tiles > 1) hipLaunchKernelGGL(k_scan_add, dim3(evad_blocks(n, 256)), dim3(256), 0, st, out, tile_sums, n);
    return evad_last_error();
}

int evad_exclusive_scan(const unsigned* in, unsigned* out, unsigned* tile_sums, long long n, hipStream_t st) {
    return evad_exclusive_scan_max(in, out, tile_sums, n, nullptr, st);
}


// =============================================================================================
// format_data (reference src/dagr/utils/buffers.py:33-44) for the events of a loader batch in ONE launch:
//   pos f32[N,3] = cat(pos i16[N,2], t i32[N]) / (W, H, T)   (torch: integer cat, then true division = both operands
//   converted to fp32 and divided, IEEE round-to-nearest),   x f32[N] = float(x i8[N]).
__global__ void k_format_events(const short* __restrict__ pos, const int* __restrict__ t, const signed char* __restrict__ x, int N,
                                const long long* __restrict__ width, const long long* __restrict__ height,
                                const long long* __restrict__ time_window, float* __restrict__ pos_out, float* __restrict__ x_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float Wf = (float)width[0], Hf = (float)height[0], Tf = (float)time_window[0];     // the batch's own (device) scalars: no host sync
    pos_out[3 * i] = __fdiv_rn((float)pos[2 * i], Wf);
    pos_out[3 * i + 1] = __fdiv_rn((float)pos[2 * i + 1], Hf);
    pos_out[3 * i + 2] = __fdiv_rn((float)t[i], Tf);
    x_out[i] = (float)x[i];
}

extern "C" int eventad_format_events(const int16_t* pos, const int32_t* t, const int8_t* x, int N, const int64_t* width,
                                     const int64_t* height, const int64_t* time_window, float* pos_out, float* x_out, void* stream) {
    if (!pos || !t || !x || !pos_out || !x_out || N < 0 || !width || !height || !time_window) return EVAD_ERR_ARG;
    if (N == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_format_events, dim3(evad_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream, (const short*)pos, t,
                       (const signed char*)x, N, (const long long*)width, (const long long*)height, (const long long*)time_window, pos_out, x_out);
    return evad_last_error();
}

// Up to EVAD_COPY_MAX device-to-device copies in ONE launch: the fields of the next batch into the static input buffers a
// captured step reads (a batch is ~10 tensors; ten copy launches in front of the backbone graph cost more stream time
// than the 10 MB they move).  blockIdx.y = segment; 16-byte vectors where both ends are 16-byte aligned, bytes else.
#define EVAD_COPY_MAX 16
struct CopyGroup { const char* src[EVAD_COPY_MAX]; char* dst[EVAD_COPY_MAX]; unsigned long long bytes[EVAD_COPY_MAX]; };
__global__ __launch_bounds__(256) void k_copy_group(CopyGroup g) {
    const char* __restrict__ src = g.src[blockIdx.y];
    char* __restrict__ dst = g.dst[blockIdx.y];
    const unsigned long long bytes = g.bytes[blockIdx.y];
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long done = 0;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const unsigned long long nv = bytes >> 4;
        for (unsigned long long i = tid; i < nv; i += nthr) ((uint4*)dst)[i] = ((const uint4*)src)[i];
        done = nv << 4;
    }
    for (unsigned long long i = done + tid; i < bytes; i += nthr) dst[i] = src[i];
}

extern "C" int eventad_copy_group(int n, const void* const* src, void* const* dst, const uint64_t* bytes, void* stream) {
    if (n < 0 || n > EVAD_COPY_MAX || (n > 0 && (!src || !dst || !bytes))) return EVAD_ERR_ARG;
    CopyGroup g;
    unsigned long long most = 0;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] == 0) continue;
        if (!src[i] || !dst[i]) return EVAD_ERR_ARG;
        g.src[m] = (const char*)src[i]; g.dst[m] = (char*)dst[i]; g.bytes[m] = bytes[i];
        if (bytes[i] > most) most = bytes[i];
        ++m;
    }
    if (m == 0) return EVAD_OK;
    for (int i = m; i < EVAD_COPY_MAX; ++i) { g.src[i] = nullptr; g.dst[i] = nullptr; g.bytes[i] = 0; }
    unsigned long long blocks = (most / 16 + 255) / 256;     // one 16-byte vector per thread of the largest segment ...
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;                        // ... up to a grid-stride loop
    hipLaunchKernelGGL(k_copy_group, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, (hipStream_t)stream, g);
    return evad_last_error();
}

// The same hand-over for a batch that sits in PINNED host memory: n asynchronous host-to-device copies on the given stream.
// (Plain hipMemcpyAsync; what it is for is the stream it is given: the CU-masked streams of a pipelined step are created
// outside torch, and a torch copy from pinned memory makes torch's host allocator record events on the copying stream that
// it touches again when the buffer is freed -- possibly after that stream is gone.)  The caller keeps the host buffers alive
// and unchanged until the stream has passed the copies.
extern "C" int eventad_copy_to_device(int n, const void* const* src_host, void* const* dst, const uint64_t* bytes, void* stream) {
    if (n < 0 || (n > 0 && (!src_host || !dst || !bytes))) return EVAD_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] == 0) continue;
        if (!src_host[i] || !dst[i]) return EVAD_ERR_ARG;
        if (hipMemcpyAsync(dst[i], src_host[i], bytes[i], hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return evad_last_error() ? evad_last_error() : -2;
    }
    return EVAD_OK;
}

// =============================================================================================
// (B) native path: counting sort by pixel + ELL edge search
// =============================================================================================
// denormalize_pos (reference ev_tgn.py:11-16): (int)(W*pos + 1e-3) with separate fp32 mul and add.
__device__ __forceinline__ int denorm(float p, float scale) { return (int)__fadd_rn(__fmul_rn(scale, p), 1e-3f); }

__global__ void k_prep_events(const float* __restrict__ pos, const long long* __restrict__ batch, int N, int B, int W,
                              int H, int T, int* __restrict__ key, int* __restrict__ tq, int* __restrict__ bev,
                              unsigned* __restrict__ count, unsigned* __restrict__ flags, int* __restrict__ arrival) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float pt = pos[3 * i + 2];
    int x = denorm(pos[3 * i], (float)W), y = denorm(pos[3 * i + 1], (float)H), t = denorm(pt, (float)T);
    int b = (int)batch[i];
    unsigned f = 0;
    if (x < 0 || x >= W || y < 0 || y >= H || b < 0 || b >= B) {
        f |= EVAD_FLAG_POS_OUT_OF_FOV;
        x = min(max(x, 0), W - 1); y = min(max(y, 0), H - 1); b = min(max(b, 0), B - 1);
    }
    if (!(pt >= 0.f && pt <= 1.f)) f |= EVAD_FLAG_T_OUT_OF_RANGE;
    if (i > 0 && (int)batch[i - 1] == b && denorm(pos[3 * i - 1], (float)T) > t) f |= EVAD_FLAG_T_NOT_MONOTONE;
    if (f) atomicOr(flags, f);
    int k = x + W * (y + H * b);
    key[i] = k; tq[i] = t; bev[i] = b;
    arrival[i] = (int)atomicAdd(&count[k], 1u);          // arrival order inside the pixel: the scatter needs no second atomic
}

__global__ void k_scatter_events(const int* __restrict__ key, const unsigned* __restrict__ start,
                                 const int* __restrict__ arrival, int* __restrict__ sid_tmp, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    sid_tmp[start[key[i]] + arrival[i]] = i;             // unordered inside the segment; k_gather_sorted ranks by id
}

// The same placement, but the event travels with its id: a 32-byte record {id, pixel key, t, polarity | x, y, t, 0} at the
// (unordered) slot of its pixel segment.  k_gather_sorted then finds everything it needs about an event in ONE coalesced
// 32-byte read instead of five scattered 4..12-byte reads by event id (each of which costs a whole 32..64-byte sector).
__global__ void k_scatter_records(const int* __restrict__ key, const unsigned* __restrict__ start, const int* __restrict__ arrival,
                                  const int* __restrict__ tq, const float* __restrict__ pos, const float* __restrict__ pol,
                                  int4* __restrict__ rec, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = key[i];
    int4* r = rec + 2 * (size_t)(start[k] + arrival[i]);
    r[0] = make_int4(i, k, tq[i], __float_as_int(pol[i]));
    r[1] = make_int4(__float_as_int(pos[3 * i]), __float_as_int(pos[3 * i + 1]), __float_as_int(pos[3 * i + 2]), 0);
}

// Sorted-order event records + the 19(+1)-channel layer-1 input row
//   x0s[r] = [polarity, 16 bilinear samples of CNN map 0, pos_x, pos_y, 0]   (reference net.py:116,122-123)
// The sampler restates F.grid_sample(align_corners=True, zeros padding) as called by
// net.py:211-228; the batch coordinate is integral so the trilinear blend reduces to bilinear.
template <bool REC>      // REC: sid_tmp holds k_scatter_records' 32-byte records instead of bare event ids
__global__ void k_gather_sorted(const int* __restrict__ sid_tmp, const unsigned* __restrict__ start, int* __restrict__ sid,
                                const int* __restrict__ key, const int* __restrict__ tq,
                                const float* __restrict__ pos, const float* __restrict__ pol,
                                const float* __restrict__ map0 /* NHWC [B,h,w,16] */, int h_in, int w_in,
                                int N, int W, int H, int* __restrict__ key_s, int* __restrict__ ts_s, int2* __restrict__ ev2,
                                float* __restrict__ posn_s, float* __restrict__ x0s,
                                float vsx, float vsy, int nx, int ny, int S, const int* __restrict__ px_to_cx,
                                const int* __restrict__ py_to_cy, int* __restrict__ node_slot, int* __restrict__ node_vox,
                                int* __restrict__ ovf_list, unsigned* __restrict__ counters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    // Thread t owns the event the counting sort dropped at (unordered) slot t of its pixel segment; its final
    // position is segment start + its id rank inside the segment (FIFO order = ascending id).  Ranking here
    // replaces two ordering launches; a segment of length L costs L loads per thread (L = 1..3 typically).
    int i, k, tqi; float px, py, pt, poli;
    if (REC) {
        const int4 a = ((const int4*)sid_tmp)[2 * (size_t)t], b4 = ((const int4*)sid_tmp)[2 * (size_t)t + 1];
        i = a.x; k = a.y; tqi = a.z; poli = __int_as_float(a.w);
        px = __int_as_float(b4.x); py = __int_as_float(b4.y); pt = __int_as_float(b4.z);
    } else {
        i = sid_tmp[t]; k = key[i]; tqi = tq[i]; poli = pol[i];
        px = pos[3 * i]; py = pos[3 * i + 1]; pt = pos[3 * i + 2];
    }
    int r = t;
    {
        unsigned s0 = start[k], s1 = start[k + 1];
        if (s1 - s0 > 1) {
            unsigned rank = 0;
            for (unsigned c = s0; c < s1; ++c) rank += ((REC ? sid_tmp[8 * (size_t)c] : sid_tmp[c]) < i);
            r = (int)(s0 + rank);
        }
    }
    sid[r] = i;
    key_s[r] = k; ts_s[r] = tqi;
    if (ev2) ev2[r] = make_int2(i, tqi);
    posn_s[3 * r] = px; posn_s[3 * r + 1] = py; posn_s[3 * r + 2] = pt;
    int b = k / (W * H);
    if (node_slot) {
        // pool1 voxel slot of this event, exactly as grid_cluster computes it from the float position
        // (pooling.py:55-56); events whose slot is not the one of their pixel (t == time_window, or a
        // position off the pixel grid) go to a side list the segmented pool kernel scans.
        int cx = (int)__fdiv_rn(px, vsx), cy = (int)__fdiv_rn(py, vsy), ct = (int)pt;
        long long sl = (long long)cx + (long long)nx * cy + (long long)nx * ny * (b + ct);
        if (sl < 0 || sl >= S) { atomicOr(counters, EVAD_FLAG_POOL_WINDOW); sl = sl < 0 ? 0 : S - 1; }
        int xi = k % W, yi = (k / W) % H;
        int tab = px_to_cx[xi] + nx * py_to_cy[yi] + nx * ny * b;
        node_slot[r] = (int)sl;
        { int pl = (int)(sl / (nx * ny)), rm = (int)(sl % (nx * ny)); node_vox[r] = (rm % nx) | ((rm / nx) << 8) | (pl << 16); }
        if ((int)sl != tab) ovf_list[atomicAdd(counters + 2, 1u)] = r;
    }
    int x0, y0; float fx, fy;
    evad_bilinear_setup(px, py, (float)W, (float)H, w_in, h_in, &x0, &y0, &fx, &fy);
    float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    bool vx0 = x0 >= 0 && x0 < w_in, vx1 = x0 + 1 >= 0 && x0 + 1 < w_in;
    bool vy0 = y0 >= 0 && y0 < h_in, vy1 = y0 + 1 >= 0 && y0 + 1 < h_in;
    const float4* m = (const float4*)(map0 + (size_t)b * h_in * w_in * 16);
    float4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](bool ok, int yy, int xx, float w) {
        if (!ok) return;
        const float4* p = m + ((size_t)yy * w_in + xx) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = p[c];
            acc[c].x += v.x * w; acc[c].y += v.y * w; acc[c].z += v.z * w; acc[c].w += v.w * w;
        }
    };
    add(vx0 && vy0, y0, x0, w00); add(vx1 && vy0, y0, x0 + 1, w01);
    add(vx0 && vy1, y0 + 1, x0, w10); add(vx1 && vy1, y0 + 1, x0 + 1, w11);
    float4* o = (float4*)(x0s + (size_t)r * 20);
    o[0] = make_float4(poli, acc[0].x, acc[0].y, acc[0].z);
    o[1] = make_float4(acc[0].w, acc[1].x, acc[1].y, acc[1].z);
    o[2] = make_float4(acc[1].w, acc[2].x, acc[2].y, acc[2].z);
    o[3] = make_float4(acc[2].w, acc[3].x, acc[3].y, acc[3].z);
    o[4] = make_float4(acc[3].w, px, py, 0.f);
}

static int sort_events_impl(const float* pos, const int64_t* batch, const float* polarity,
                            const float* map0_nhwc, int map_h, int map_w,
                            int N, int B, int W, int H, int T,
                            int32_t* key, int32_t* tq, int32_t* bev, uint32_t* pix_count, uint32_t* pix_start,
                            uint32_t* scan_tmp, int32_t* sid_tmp, int32_t* rec /* nullable: i32[N,8] */, int32_t* sid,
                            uint32_t* counters /* [0]=flags [2]=ovf count [3]=longest segment */,
                            int32_t* key_s, int32_t* ts_s, int32_t* ev2, float* posn_s, float* x0s,
                            float vsx, float vsy, int nx, int ny, const int32_t* px_to_cx, const int32_t* py_to_cy,
                            int32_t* node_slot, int32_t* node_vox, int32_t* ovf_list, int zero_tail_words, void* stream) {
    if (zero_tail_words < 0 || !pos || !batch || !polarity || !map0_nhwc || !key || !tq || !bev || !pix_count || !pix_start || !scan_tmp ||
        !sid_tmp || !sid || !counters || !key_s || !ts_s || !posn_s || !x0s ||
        N < 0 || B <= 0 || W <= 0 || H <= 0 || T <= 0 || map_h <= 0 || map_w <= 0)
        return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    long long P = (long long)B * H * W;
    if (P >= (1ll << 31)) return EVAD_ERR_ARG;
    // counters (all four words: this is the first call of a step), the pixel histogram and -- when the caller laid
    // them out back to back -- `zero_tail_words` more words behind it (the step's other zero-initialised scratch)
    if (counters + 4 == pix_count) evad_zero_async(counters, sizeof(uint32_t) * (4 + P + 1 + (size_t)zero_tail_words), st);
    else {
        if (zero_tail_words) return EVAD_ERR_ARG;
        evad_zero_async(pix_count, sizeof(uint32_t) * (P + 1), st);
        evad_zero_async(counters, 4 * sizeof(uint32_t), st);
    }
    if (node_slot && (!px_to_cx || !py_to_cy || !ovf_list || !node_vox || nx <= 0 || ny <= 0 || vsx <= 0 || vsy <= 0)) return EVAD_ERR_ARG;
    if (N > 0)
        hipLaunchKernelGGL(k_prep_events, dim3(evad_blocks(N, 256)), dim3(256), 0, st, pos, (const long long*)batch, N, B,
                           W, H, T, key, tq, bev, pix_count, counters, sid);      // sid doubles as the arrival-rank scratch
    int rc = evad_exclusive_scan_max(pix_count, pix_start, scan_tmp, P, counters + 3, st);   // [3] = longest pixel segment
    if (rc) return rc;
    if (N == 0) return evad_last_error();
    if (rec) {
        if ((uintptr_t)rec & 15) return EVAD_ERR_ARG;
        hipLaunchKernelGGL(k_scatter_records, dim3(evad_blocks(N, 256)), dim3(256), 0, st, key, pix_start, sid, tq, pos, polarity,
                           (int4*)rec, N);
        hipLaunchKernelGGL(k_gather_sorted<true>, dim3(evad_blocks(N, 256)), dim3(256), 0, st, rec, pix_start, sid, key, tq, pos,
                           polarity, map0_nhwc, map_h, map_w, N, W, H, key_s, ts_s, (int2*)ev2, posn_s, x0s, vsx, vsy, nx, ny,
                           nx * ny * (B + 1), px_to_cx, py_to_cy, node_slot, node_vox, ovf_list, counters);
        return evad_last_error();
    }
    hipLaunchKernelGGL(k_scatter_events, dim3(evad_blocks(N, 256)), dim3(256), 0, st, key, pix_start, sid, sid_tmp, N);
    hipLaunchKernelGGL(k_gather_sorted<false>, dim3(evad_blocks(N, 256)), dim3(256), 0, st, sid_tmp, pix_start, sid, key, tq, pos, polarity,
                       map0_nhwc, map_h, map_w, N, W, H, key_s, ts_s, (int2*)ev2, posn_s, x0s, vsx, vsy, nx, ny, nx * ny * (B + 1),
                       px_to_cx, py_to_cy, node_slot, node_vox, ovf_list, counters);
    return evad_last_error();
}

extern "C" int eventad_sort_events(const float* pos, const int64_t* batch, const float* polarity,
                                   const float* map0_nhwc, int map_h, int map_w,
                                   int N, int B, int W, int H, int T,
                                   int32_t* key, int32_t* tq, int32_t* bev, uint32_t* pix_count, uint32_t* pix_start,
                                   uint32_t* scan_tmp, int32_t* sid_tmp, int32_t* rec, int32_t* sid, uint32_t* counters,
                                   int32_t* key_s, int32_t* ts_s, int32_t* ev2, float* posn_s, float* x0s,
                                   float vsx, float vsy, int nx, int ny, const int32_t* px_to_cx, const int32_t* py_to_cy,
                                   int32_t* node_slot, int32_t* node_vox, int32_t* ovf_list, int zero_tail_words, void* stream) {
    return sort_events_impl(pos, batch, polarity, map0_nhwc, map_h, map_w, N, B, W, H, T, key, tq, bev, pix_count, pix_start, scan_tmp,
                            sid_tmp, rec, sid, counters, key_s, ts_s, ev2, posn_s, x0s, vsx, vsy, nx, ny, px_to_cx, py_to_cy, node_slot,
                            node_vox, ovf_list, zero_tail_words, stream);
}

// ---------------------------------------------------------------------------------------------
// The same sort for callers that cannot allocate device memory (code under hipGraph capture, tight serving loops):
// the pixel histogram, the scan's tile sums and the unordered scatter target live in ONE file-scope device array
// instead of caller-provided workspace.  Every GPU the library runs on holds its own copy of the array (8.1 MiB) until
// the process ends; its address on a device is looked up once per device with that device current.  All calls on
// one GPU share the array, so calls that overlap on one GPU (different streams) race: serialise them.
#define EVAD_STATIC_MAX_PIXELS (1 << 20)                 /* B*H*W  (C2: 8 x 256 x 256 = 524288) */
#define EVAD_STATIC_MAX_EVENTS (1 << 20)                 /* N      (C2: 400000) */
#define EVAD_STATIC_SCAN_WORDS (2 * (EVAD_STATIC_MAX_PIXELS / SCAN_TILE + 2))
#define EVAD_STATIC_OFF_SCAN   (EVAD_STATIC_MAX_PIXELS + 64)                                   /* histogram: P + 1 words */
#define EVAD_STATIC_OFF_SID    (EVAD_STATIC_OFF_SCAN + ((EVAD_STATIC_SCAN_WORDS + 63) / 64) * 64)
#define EVAD_STATIC_WORDS      (EVAD_STATIC_OFF_SID + EVAD_STATIC_MAX_EVENTS)
static __device__ __align__(256) uint32_t scratch[EVAD_STATIC_WORDS];
static uint32_t* g_scratch_ptr[EVAD_MAX_DEVICES];        // device address of `scratch` per GPU (the symbol itself is no address)
static std::mutex g_scratch_mutex;

static int scratch_for_current_device(uint32_t** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EVAD_MAX_DEVICES) return -2;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return e != hipSuccess ? (int)e : -2;
        g_scratch_ptr[dev] = (uint32_t*)p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int eventad_sort_events_static(const float* pos, const int64_t* batch, const float* polarity,
                                          const float* map0_nhwc, int map_h, int map_w, int N, int B, int W, int H, int T,
                                          int32_t* key, int32_t* tq, int32_t* bev, uint32_t* pix_start, int32_t* sid,
                                          uint32_t* counters, int32_t* key_s, int32_t* ts_s, int32_t* ev2, float* posn_s,
                                          float* x0s, float vsx, float vsy, int nx, int ny, const int32_t* px_to_cx,
                                          const int32_t* py_to_cy, int32_t* node_slot, int32_t* node_vox, int32_t* ovf_list,
                                          void* stream) {
    if (N < 0 || B <= 0 || W <= 0 || H <= 0) return EVAD_ERR_ARG;
    if ((long long)B * H * W > EVAD_STATIC_MAX_PIXELS || N > EVAD_STATIC_MAX_EVENTS) {
        fprintf(stderr, "eventad_sort_events_static: B*H*W = %lld pixels / N = %d events exceed the static workspace "
                        "(at most %d pixels and %d events); use eventad_sort_events with caller-owned workspace\n",
                (long long)B * H * W, N, EVAD_STATIC_MAX_PIXELS, EVAD_STATIC_MAX_EVENTS);
        return EVENTAD_ERR_TOO_LARGE;
    }
    uint32_t* ws = nullptr;
    int rc = scratch_for_current_device(&ws);
    if (rc) return rc;
    // the histogram is cleared by sort_events_impl (fill kernel) on every call: the array keeps the last call's contents
    return sort_events_impl(pos, batch, polarity, map0_nhwc, map_h, map_w, N, B, W, H, T, key, tq, bev, ws /* pix_count */, pix_start,
                            ws + EVAD_STATIC_OFF_SCAN, (int32_t*)(ws + EVAD_STATIC_OFF_SID), nullptr /* no record scratch: ids only */, sid, counters,
                            key_s, ts_s, ev2, posn_s,
                            x0s, vsx, vsy, nx, ny, px_to_cx, py_to_cy, node_slot, node_vox, ovf_list, 0, stream);
}

// Edge search over pixel CSR segments (semantics of ev_graph.cu:15-80, see header comment).
// One thread per destination event, in pixel-sorted order r.  Writes ELL row r:
//   ell_src[r][k] = sorted position of the k-th source (k=0 is the self edge),
//   ell_step[r][k] = spiral step (= LUT cell) of that source's pixel,  deg[r] = number of slots used.
__global__ __launch_bounds__(256) void k_build_ell(const int* __restrict__ sid, const int* __restrict__ ts_s,
                                                   const int* __restrict__ key_s, const unsigned* __restrict__ start,
                                                   const int16_t* __restrict__ spiral, const unsigned* __restrict__ flags,
                                                   int N, int W, int H, int steps, int Q, int dt_max, int max_nb,
                                                   int* __restrict__ ell_src, uint16_t* __restrict__ ell_step,
                                                   int* __restrict__ deg) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const bool monotone = !(*flags & EVAD_FLAG_T_NOT_MONOTONE);
    int k = key_s[r], me = sid[r], ts = ts_s[r];
    int x = k % W, y = (k / W) % H;
    int plane_base = k - x - W * y;                     // W*H*b
    int* row = ell_src + (size_t)r * max_nb;
    uint16_t* srow = ell_step + (size_t)r * max_nb;
    row[0] = r; srow[0] = 0;
    int nn = 1;
    for (int i = 0; i < steps && nn < max_nb; ++i) {
        int xn = x + spiral[2 * i], yn = y + spiral[2 * i + 1];
        if (xn < 0 || yn < 0 || xn >= W || yn >= H) continue;
        int p = plane_base + xn + W * yn;
        int st = (int)start[p], en = (int)start[p + 1];
        int lo = max(st, en - Q);                        // the FIFO keeps only the newest Q ids
        for (int s = en - 1; s >= lo; --s) {
            if (me > sid[s]) {
                // dt > delta: the reference keeps scanning (`continue`); with timestamps that are
                // non-decreasing in id every older entry fails too, so the scan can stop.
                if ((float)(ts - ts_s[s]) > (float)dt_max) { if (monotone) break; else continue; }
                row[nn] = s; srow[nn] = (uint16_t)i;
                if (++nn >= max_nb) break;
            }
        }
    }
    deg[r] = nn;
}

// Thread-per-event walk with (a) the spiral table read through scalar loads and (b) the ELL row kept in
// registers and written once as 4 x int4 + 2 x uint4.  Measured with PMC counters on C2, the plain walk
// issues ~620 vector-memory instructions per wave, a third of them 2- and 4-byte scattered stores; on
// CDNA a load's s_waitcnt also waits for every older store (vmcnt counts both), so stores inside the
// dependent chain lengthen every later round trip.
// (c) Lane assignment: a wave pays for the longest walk of its 64 lanes, and the length of a walk depends mostly on
// how EARLY the event is in its clip (an early event finds few older events per pixel and walks far; with dense
// input every 64 consecutive pixel-sorted events hold one such event).  A workgroup therefore takes a tile of
// 1024 consecutive pixel-sorted events and deals them to its 16 waves by id rank (64 LDS buckets over the tile's
// id range): the long walks of a tile share a few waves and the other waves finish early.  Which lane computes
// which row changes nothing in the rows themselves.  The price is that a wave's loads touch more cache lines, so
// the host asks for it (BY_RANK) only for dense input (C5: 23.2 vs 28.5 ms; C4, 0.5 events per pixel: 7.3 vs 6.5 ms).
#define ELL_TILE 1024
template <bool MARK, bool BY_RANK>
__global__ __launch_bounds__(ELL_TILE) void k_build_ell_reg(const int2* __restrict__ ev2,
                                                       const int* __restrict__ key_s, const unsigned* __restrict__ start,
                                                       const unsigned* flags_in, int N, int W, int H, int steps, int sp0, int Q,
                                                       int dt_max, int* __restrict__ ell_src, uint16_t* __restrict__ ell_step,
                                                       int* __restrict__ deg, const int* __restrict__ node_slot,
                                                       const int* __restrict__ node_vox, unsigned* __restrict__ bitmap,
                                                       unsigned* flags, int only_not_monotone) {
    constexpr int MAXNB = 16;
    // only_not_monotone: this launch stands behind a kernel that leaves batches with non-monotone timestamps alone
    if (only_not_monotone && !(*flags_in & EVAD_FLAG_T_NOT_MONOTONE)) return;
    __shared__ int s_perm[ELL_TILE];
    __shared__ int s_cnt[64];
    __shared__ int s_mm[2];
    const int tid = threadIdx.x, r0 = blockIdx.x * blockDim.x + tid;     // BY_RANK: blockDim.x == ELL_TILE
    int r = r0;
    if (BY_RANK) {
    if (tid < 64) s_cnt[tid] = 0;
    if (tid == 0) { s_mm[0] = 0x7fffffff; s_mm[1] = 0; }
    __syncthreads();
    const int id0 = r0 < N ? ev2[r0].x : -1;
    {
        int mn = id0 < 0 ? 0x7fffffff : id0, mx = id0 < 0 ? 0 : id0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { mn = min(mn, __shfl_xor(mn, d, 64)); mx = max(mx, __shfl_xor(mx, d, 64)); }
        if ((tid & 63) == 0) { atomicMin(&s_mm[0], mn); atomicMax(&s_mm[1], mx); }
    }
    __syncthreads();
    const int lo_id = s_mm[0];
    const float scale = 64.f / (float)(s_mm[1] - lo_id + 1);
    const int bucket = id0 < 0 ? 63 : min(63, (int)((float)(id0 - lo_id) * scale));
    const int rank_in_bucket = atomicAdd(&s_cnt[bucket], 1);
    __syncthreads();
    if (tid < 64) {                                              // exclusive scan of the 64 bucket counts (wave 0)
        const int c = s_cnt[tid];
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (tid >= d) incl += o; }
        s_cnt[tid] = incl - c;
    }
    __syncthreads();
    s_perm[s_cnt[bucket] + rank_in_bucket] = r0;
    __syncthreads();
    r = s_perm[tid];
    }
    if (r >= N) return;
    const bool monotone = !(*flags_in & EVAD_FLAG_T_NOT_MONOTONE);
    const int k = key_s[r];
    const int2 mine = ev2[r];
    const int me = mine.x, ts = mine.y;
    const int x = k % W, y = (k / W) % H, plane_base = k - x - W * y;
    int nb[MAXNB], stp[MAXNB];
#pragma unroll
    for (int e = 0; e < MAXNB; ++e) { nb[e] = 0; stp[e] = 0; }
    nb[0] = r;
    int nn = 1;
    for (int i = 0; i < steps && nn < MAXNB; ++i) {
        int xn = x + c_spiral[2 * (sp0 + i)], yn = y + c_spiral[2 * (sp0 + i) + 1];
        if (xn < 0 || yn < 0 || xn >= W || yn >= H) continue;
        int p = plane_base + xn + W * yn;
        int st = (int)start[p], en = (int)start[p + 1];
        int lo = max(st, en - Q);
        for (int s = en - 1; s >= lo; --s) {
            const int2 c = ev2[s];                               // {id, t} in one 8-byte load
            if (me > c.x) {
                if ((float)(ts - c.y) > (float)dt_max) { if (monotone) break; else continue; }
#pragma unroll
                for (int e = 1; e < MAXNB; ++e) if (e == nn) { nb[e] = s; stp[e] = i; }
                if (++nn >= MAXNB) break;
            }
        }
    }
    int4* rp = (int4*)(ell_src + (size_t)r * MAXNB);
    rp[0] = make_int4(nb[0], nb[1], nb[2], nb[3]); rp[1] = make_int4(nb[4], nb[5], nb[6], nb[7]);
    rp[2] = make_int4(nb[8], nb[9], nb[10], nb[11]); rp[3] = make_int4(nb[12], nb[13], nb[14], nb[15]);
    uint4* sp4 = (uint4*)(ell_step + (size_t)r * MAXNB);
    sp4[0] = make_uint4(stp[0] | stp[1] << 16, stp[2] | stp[3] << 16, stp[4] | stp[5] << 16, stp[6] | stp[7] << 16);
    sp4[1] = make_uint4(stp[8] | stp[9] << 16, stp[10] | stp[11] << 16, stp[12] | stp[13] << 16, stp[14] | stp[15] << 16);
    deg[r] = nn;
    if (MARK) {     // pool1 neighbour bitmap while the row is still in registers (replaces k_mark_ell's second pass)
        const int my_slot = node_slot[r], my_vox = node_vox[r];
        int vs[MAXNB];
#pragma unroll
        for (int e = 1; e < MAXNB; ++e) vs[e] = e < nn ? node_vox[nb[e]] : my_vox;      // independent gathers
#pragma unroll
        for (int e = 1; e < MAXNB; ++e) if (e < nn) evad_mark_edge_packed(vs[e], my_vox, my_slot, bitmap, flags);
    }
}

// ---------------------------------------------------------------------------------------------
// Row-scan edge search for sparse windows (mean events per (2R+1)^2 window well below 64; R = 1 or 3).
// The 49 probes of a 7x7 window cost ~245 lane-instructions each in the kernels above, mostly address
// arithmetic and divergence on (half-)empty pixels.  But the pixels x-R..x+R of one sensor row are ONE
// contiguous range of the pixel-sorted event array, so a destination needs 2R+1 range look-ups, not (2R+1)^2:
// every lane scans the events of its 2R+1 row ranges (all lanes converged; trip count = the longest range of the
// wave) and tests each candidate exactly as the reference does (older id, dt <= delta_t).  The survivors of one pixel
// are consecutive candidates (a pixel is a contiguous piece of the row range, ids and -- in a batch with monotone
// timestamps -- times ascend inside it), so a lane keeps them as RUNS: per spiral step (offset of the run's first
// survivor in the row range, length) in a per-lane LDS table, plus a 64-bit mask of the steps that have a run.  The
// reference's row is then read off in its own order -- steps ascending = the mask's set bits from the lowest, newest
// first inside a run -- until 15 sources are taken: no sorting.  (Selecting the 15 smallest of per-candidate keys, the
// earlier form, was two thirds of the kernel's instructions.)  Batches whose timestamps are not monotone, with a pixel
// deeper than the FIFO (the depth cap then matters) or than 31 events, or row ranges longer than 2047 take the exact
// per-lane spiral walk instead.
#define RS_UNR 8
template <int MAXNB>
__device__ __forceinline__ int ell_exact_walk(const int2* __restrict__ ev2, const unsigned* __restrict__ start, int me, int ts,
                                              int x, int y, int plane_base, int W, int H, int steps, int sp0, int Q, int dt_max,
                                              bool monotone, int (&nb)[MAXNB], int (&stp)[MAXNB]) {
    int nn = 1;
    for (int i = 0; i < steps && nn < MAXNB; ++i) {
        int xn = x + c_spiral[2 * (sp0 + i)], yn = y + c_spiral[2 * (sp0 + i) + 1];
        if (xn < 0 || yn < 0 || xn >= W || yn >= H) continue;
        int p = plane_base + xn + W * yn;
        int st = (int)start[p], en = (int)start[p + 1];
        int lo = max(st, en - Q);
        for (int s = en - 1; s >= lo; --s) {
            const int2 c = ev2[s];
            if (me > c.x) {
                if ((float)(ts - c.y) > (float)dt_max) { if (monotone) break; else continue; }
#pragma unroll
                for (int e = 1; e < MAXNB; ++e) if (e == nn) { nb[e] = s; stp[e] = i; }
                if (++nn >= MAXNB) break;
            }
        }
    }
    return nn;
}

// waves_per_eu(5,5) is a register cap (<= 96 VGPRs; the kernel needs 86 without spilling, 112 uncapped): its LDS allows 4
// waves per SIMD anyway, and 4 x 86 leaves room for a wave of the head's kernels on the same SIMD when the streams overlap.
template <int R, bool MARK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_build_ell_rowscan(const int2* __restrict__ ev2, const int* __restrict__ key_s,
                                                           const unsigned* __restrict__ start, const unsigned* flags_in, int N,
                                                           int W, int H, int steps, int sp0, int Q, int dt_max, int* __restrict__ ell_src,
                                                           uint16_t* __restrict__ ell_step, int* __restrict__ deg,
                                                           const int* __restrict__ node_slot, const int* __restrict__ node_vox,
                                                           unsigned* __restrict__ bitmap, unsigned* flags) {
    constexpr int MAXNB = 16, D = 2 * R + 1;
    __shared__ unsigned short s_run[256][D * D + 1 + (D & 1 ? 0 : 1)];     // [lane][step] = first offset << 5 | length (odd word stride)
    __shared__ unsigned short s_stepof[D * D];
    __shared__ signed char s_dyi[D * D];
    const int tid = threadIdx.x;
    if (tid < D * D) { s_stepof[tid] = (unsigned short)c_stepof[sp0 + tid]; s_dyi[tid] = (signed char)(c_spiral[2 * (sp0 + tid) + 1] + R); }
    __syncthreads();
    const int r = blockIdx.x * 256 + tid;
    const bool active = r < N;
    const bool monotone = !(flags_in[0] & EVAD_FLAG_T_NOT_MONOTONE);
    const int lmax = (int)flags_in[3];
    int me = 0, ts = 0, x = 0, y = 0, plane_base = 0;
    if (active) {
        const int k = key_s[r];
        const int2 mine = ev2[r];
        me = mine.x; ts = mine.y; x = k % W; y = (k / W) % H; plane_base = k - x - W * y;
    }
    int nb[MAXNB], stp[MAXNB];
#pragma unroll
    for (int e = 0; e < MAXNB; ++e) { nb[e] = 0; stp[e] = 0; }
    nb[0] = r;
    int nn = 1;
    // uniform: survivors of a pixel not contiguous / depth cap binding / run length or offsets beyond their bit fields
    const bool exact = !monotone || lmax > Q || lmax > 31 || lmax * D > 2047;
    if (!exact) {
        int ra[D], rbv[D], n = 0;
        unsigned long long mask = 0ull;                      // bit s: spiral step s has a run
        const int xlo = max(x - R, 0), xhi = min(x + R, W - 1);
#pragma unroll
        for (int dyi = 0; dyi < D; ++dyi) {                  // all 2(2R+1) range bounds in flight at once
            const int yn = y + dyi - R;
            ra[dyi] = 0; rbv[dyi] = 0;
            if (active && yn >= 0 && yn < H) {
                const int rowbase = plane_base + W * yn;
                ra[dyi] = (int)start[rowbase + xlo]; rbv[dyi] = (int)start[rowbase + xhi + 1];
            }
        }
#pragma unroll
        for (int dyi = 0; dyi < D; ++dyi) {
            const int len = rbv[dyi] - ra[dyi];
            int maxlen = len;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, d, 64));
            const int xbase = plane_base + W * (y + dyi - R) + x - R;   // key of the window's left-most pixel in this row
            int cur_s = -1, cur_first = 0, cur_cnt = 0;      // the run being collected (a pixel's survivors are consecutive)
            for (int j0 = 0; j0 < maxlen; j0 += RS_UNR) {    // RS_UNR candidates (2 RS_UNR independent loads) per round trip
                int2 ce[RS_UNR]; int kc[RS_UNR];
#pragma unroll
                for (int u = 0; u < RS_UNR; ++u) {
                    ce[u] = make_int2(0x7fffffff, 0); kc[u] = 0;
                    if (j0 + u < len) { ce[u] = ev2[ra[dyi] + j0 + u]; kc[u] = key_s[ra[dyi] + j0 + u]; }
                }
#pragma unroll
                for (int u = 0; u < RS_UNR; ++u) {
                    if (me > ce[u].x && !((float)(ts - ce[u].y) > (float)dt_max)) {
                        const int st = (int)s_stepof[dyi * D + kc[u] - xbase];        // (kc - xbase = 0..2R)
                        if (st != cur_s) {
                            if (cur_cnt) s_run[tid][cur_s] = (unsigned short)(cur_first << 5 | cur_cnt);
                            mask |= 1ull << st;
                            cur_s = st; cur_first = j0 + u; cur_cnt = 0;
                        }
                        ++cur_cnt; ++n;
                    }
                }
            }
            if (cur_cnt) s_run[tid][cur_s] = (unsigned short)(cur_first << 5 | cur_cnt);
        }
        // (s_run[tid] is written and read by this lane only: program order is all it needs)
        asm volatile("" ::: "memory");
        const int ntake = min(n, MAXNB - 1);
        int left = 0, top = 0, cur = 0, taken = 0;           // sources left in the current run, row position of its newest + 1
#pragma unroll
        for (int e = 1; e < MAXNB; ++e) {
            if (e <= ntake && (left > 0 || mask != 0ull)) {  // (the runs hold n sources; never read past the last run)
                ++taken;
                if (left == 0) {
                    cur = __ffsll((long long)mask) - 1;      // the next spiral step that has survivors
                    mask &= mask - 1ull;
                    const unsigned ent = s_run[tid][cur];
                    left = (int)(ent & 31u);
                    const int dyi = s_dyi[cur];
                    int base = ra[0];
#pragma unroll
                    for (int t = 1; t < D; ++t) if (t == dyi) base = ra[t];
                    top = base + (int)(ent >> 5) + left;
                }
                nb[e] = top - 1; stp[e] = cur;               // newest first inside a pixel
                --top; --left;
            }
        }
        nn = 1 + taken;
    }
    if (exact && active) nn = ell_exact_walk<MAXNB>(ev2, start, me, ts, x, y, plane_base, W, H, steps, sp0, Q, dt_max, monotone, nb, stp);
    if (active) {
        int4* rp = (int4*)(ell_src + (size_t)r * MAXNB);
        rp[0] = make_int4(nb[0], nb[1], nb[2], nb[3]); rp[1] = make_int4(nb[4], nb[5], nb[6], nb[7]);
        rp[2] = make_int4(nb[8], nb[9], nb[10], nb[11]); rp[3] = make_int4(nb[12], nb[13], nb[14], nb[15]);
        uint4* sp4 = (uint4*)(ell_step + (size_t)r * MAXNB);
        sp4[0] = make_uint4(stp[0] | stp[1] << 16, stp[2] | stp[3] << 16, stp[4] | stp[5] << 16, stp[6] | stp[7] << 16);
        sp4[1] = make_uint4(stp[8] | stp[9] << 16, stp[10] | stp[11] << 16, stp[12] | stp[13] << 16, stp[14] | stp[15] << 16);
        deg[r] = nn;
    }
    if (MARK) {                                              // every lane takes part in the wave-level combine
        const int my_slot = active ? node_slot[r] : -1, my_vox = active ? node_vox[r] : 0;
        if (!active) nn = 0;
        int vs[MAXNB];
#pragma unroll
        for (int e = 1; e < MAXNB; ++e) vs[e] = e < nn ? node_vox[nb[e]] : my_vox;
        evad_mark_row_packed<MAXNB>(vs, nn, my_vox, my_slot, bitmap, flags);
    }
}

// ---------------------------------------------------------------------------------------------
// Pixel-centric edge search for dense input (several events per pixel: C3, C5).
// All events of one pixel walk the SAME neighbour pixels in the same order; they differ only in their own id and time.
// The per-event walks above load every neighbour pixel's bounds and FIFO entries once per destination EVENT, one
// dependent round trip at a time, and a wave waits for the longest walk of its 64 lanes.  Here one wave owns one
// destination PIXEL at a time.  A pass covers GS consecutive spiral steps: lane l of each of the 64/GS lane
// groups holds, in registers, the bounds and the PIX_DEPTH newest FIFO entries of step i0 + l (two round trips per
// pass), and the groups then go through the pixel's unfinished events, 64/GS at a time, one event per group.  For its
// event a lane counts, among its entries, those not older than the event (ids ascend inside a segment, so they are the
// newest `nnew`) and those that pass the reference's tests (older id, dt <= delta_t; with timestamps that do not
// decrease with the id they are the next `nq`); the exclusive sum of nq over the lower lanes of the group -- spiral
// order -- is where the lane's sources go in the row, newest first, until 16 slots are taken: the reference's row.
// A lane whose registers do not decide a deep pixel (all PIX_DEPTH entries newer or taken) reads on from memory.
// The first pass takes the whole spiral when it has at most 64 steps (radius <= 3; one event at a time), else it runs
// with 64/GS1 groups of GS1 steps (16 at five or more events per pixel, else 32: most events are done after that many
// pixels) and the events that still have fewer than 16 sources -- the early ones of the clip -- go on alone with 64
// steps per pass.
// Rows are assembled in LDS and written as whole lines.  Batches whose timestamps are not monotone take an exact
// per-event walk (the `continue` of the reference matters then).
#define PIX_DEPTH 8            // (default; k_build_ell_pix takes the depth as a template argument)
// LDS written by one lane and read by another lane of the same wave: the LDS unit executes one wave's instructions in
// order, so only the compiler has to keep them in order.  (A wavefront-scope fence also waits for every outstanding
// global load and store -- the rows just written, the bounds requested ahead for the next pixel.)
#define PIX_LDS_ORDER() asm volatile("" ::: "memory")
#define PIX_CHUNK 16            // destination events of a pixel per round (a pixel rarely holds more)
// segment bounds of the neighbour pixel at offset (sdx, sdy) from (x, y); live = the lane's spiral step exists.  A pixel
// outside the sensor (or a lane past the last step) gets an empty segment.
struct __attribute__((aligned(4))) PixBounds { unsigned first, end; };
__device__ __forceinline__ void pix_bounds(const unsigned* __restrict__ start, int x, int y, int plane_base, int W, int H, int sdx, int sdy,
                                           bool live, int& st, int& en) {
    const int xn = x + sdx, yn = y + sdy;
    const bool ok = live && xn >= 0 && yn >= 0 && xn < W && yn < H;
    const PixBounds se = *(const PixBounds*)(start + (ok ? plane_base + xn + W * yn : 0));      // one 8-byte load (4-byte aligned)
    en = (int)se.end;
    st = ok ? (int)se.first : en;
}

template <int GS, bool INT_CMP, bool PRE, int DEPTH>
__device__ __forceinline__ void ell_pix_pass(const int2* __restrict__ ev2, const unsigned* __restrict__ start, int N, int W, int H,
                                             int steps, int sp0, int Q, int dt_max, int x, int y, int plane_base, int i0, int md,
                                             int* my_src, unsigned short* my_stp, const int2* my_dest, int* my_cnt, int st, int en) {
    constexpr int MAXNB = 16, NG = 64 / GS;
    const int lane = threadIdx.x & 63, g = lane / GS, l = lane % GS;
    const float fdt = (float)dt_max;
    const int i = i0 + l;
    if (!PRE) {                                              // (the first pass gets its bounds from the caller, loaded ahead)
        const int ii = min(i, steps - 1);
        pix_bounds(start, x, y, plane_base, W, H, c_spiral[2 * (sp0 + ii)], c_spiral[2 * (sp0 + ii) + 1], i < steps, st, en);
    }
    const int lo = max(st, en - Q);                          // the FIFO keeps only the newest Q ids
    const int len = en - lo, lenr = min(len, DEPTH), ninv = DEPTH - lenr;
    // oldest entry of a deep pixel: if even that one is not older than the event, nothing below the registers is
    const int eold = len > DEPTH ? ev2[min(lo, N - 1)].x : 0x7fffffff;
    int eid[DEPTH], et[DEPTH];
    // Padding (entries past the segment's length) counts as "not older" -- subtracted again through ninv -- and as out of
    // the time window.  The asm statement keeps the padded values (two compares per entry and event below, instead of a
    // mask per entry carried into the loop) and the two load sequences apart.
#define PIX_ENTRY(V, CX, CY) { eid[V] = (V) < lenr ? (CX) : 0x7fffffff; et[V] = (V) < lenr ? (CY) : -(1 << 30); \
                               asm volatile("" : "+v"(eid[V]), "+v"(et[V])); }
    // The DEPTH newest entries end at en - 1: one base address, 16-byte loads of two entries each (entries are 8-byte
    // aligned, which is all a global load needs; an odd length reads one entry of the neighbouring segment, harmless).  A
    // lane only asks for the pairs its segment has.  Only a wave that touches the first DEPTH events of the whole array
    // clamps every index instead.
    if (__all(N >= DEPTH && (en >= DEPTH || len == 0))) {
        struct __attribute__((aligned(8))) Piece { int4 v; };
        const Piece* top = (const Piece*)(ev2 + (len > 0 ? en - DEPTH : 0));
#pragma unroll
        for (int c = DEPTH / 2 - 1; c >= 0; --c) {
            int4 r = make_int4(0, 0, 0, 0);
            if (lenr > 2 * (DEPTH / 2 - 1 - c)) r = top[c].v;    // entries en - DEPTH + 2c (older) and en - DEPTH + 1 + 2c
            PIX_ENTRY(DEPTH - 1 - 2 * c, r.x, r.y) PIX_ENTRY(DEPTH - 2 - 2 * c, r.z, r.w)
        }
    } else {
#pragma unroll
        for (int v = 0; v < DEPTH; ++v) { const int2 c = ev2[min(max(en - 1 - v, lo), N - 1)]; PIX_ENTRY(v, c.x, c.y) }
    }
#undef PIX_ENTRY
    unsigned long long todo = __ballot(lane < PIX_CHUNK && my_cnt[lane & (PIX_CHUNK - 1)] < MAXNB);   // bit d: event d still lacks sources
    while (todo) {
        int d = __ffsll((long long)todo) - 1;                // this group's event: the g-th lowest set bit, found on the scalar
        todo &= todo - 1;                                    // unit for every group and then picked by the lane's group
#pragma unroll
        for (int k = 1; k < NG; ++k) {
            const int dk = __ffsll((long long)todo) - 1;     // -1 once no event is left (clearing a bit of 0 leaves 0)
            todo &= todo - 1;
            d = g == k ? dk : d;
        }
        const bool act = d >= 0;
        d = max(d, 0);
        const int base = my_cnt[d];
        const int2 mine = my_dest[d];
        const int me = mine.x, ts = mine.y;
        // Entries run from the newest to older ones, so "not older than the event" holds for a prefix of them and "inside the
        // time window" for a prefix too (timestamps ascend with the ids here): two independent counts, and the sources are
        // the entries between the two prefixes.
        int cge = 0, nin = 0;
        // (float)(ts - t) > (float)delta_t  <=>  ts - t > delta_t  <=>  t < ts - delta_t  whenever |delta_t| + 1 is exactly
        // representable (timestamps are microseconds inside one clip: no overflow)
        const int t_min = ts - dt_max;
#pragma unroll
        for (int v = 0; v < DEPTH; ++v) {
            cge += eid[v] >= me ? 1 : 0;
            if (INT_CMP) nin += et[v] >= t_min ? 1 : 0;
            else         nin += (float)(ts - et[v]) > fdt ? 0 : 1;
        }
        int nnew = cge - ninv;
        int nq = max(0, nin - nnew);
        if (!act) nq = 0;
        if (act && eold < me && nnew + nq == DEPTH) {                // deep pixel, registers undecided: read on,
            bool stop = false;                                           // four entries per round trip
            for (int s = en - 1 - DEPTH; s >= lo && nq < MAXNB && !stop; s -= 4) {
                int2 c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = ev2[max(s - k, lo)];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (stop || s - k < lo || nq >= MAXNB) continue;
                    if (c[k].x >= me) { ++nnew; continue; }
                    const int dt = ts - c[k].y;
                    if (INT_CMP ? dt > dt_max : (float)dt > fdt) stop = true; else ++nq;
                }
            }
        }
        // inclusive sum of nq over the lanes of the group, lane order = spiral order: a shift-and-add scan inside each row of
        // 16 lanes (DPP row_shr 1, 2, 4, 8), then the last lane of a row added to the whole next row where a group spans rows
        int inc = nq;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
        if (GS >= 32) inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);    // row_bcast15 -> rows 1 and 3
        if (GS == 64) inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);    // row_bcast31 -> rows 2 and 3
        const int off = inc - nq;
        const int take = min(nq, MAXNB - base - off);        // slots left for this lane's sources
        for (int k = 0; k < take; ++k) {
            my_src[d * MAXNB + base + off + k] = en - 1 - nnew - k;
            my_stp[d * MAXNB + base + off + k] = (unsigned short)i;
        }
        if (l == GS - 1 && act) my_cnt[d] = min(MAXNB, base + inc);      // the group's last lane holds the total
    }
    PIX_LDS_ORDER();
}

// First pass for DEEP pixels (tens of events per pixel: the 32 M-event configuration).  With the newest DEPTH entries of a
// neighbour pixel in registers, an event in the middle of the clip finds all of them newer than itself and reads on from
// memory, four entries per dependent round trip, past every newer entry and then through its sources -- 6 to 10 round trips
// per event, which is where such a pixel's time went.  But a group's events come in ascending id, ids and timestamps ascend
// inside a segment, and both "older than the event" and "inside its time window" are prefixes / suffixes of the segment that
// only move UP from one event to the next.  So a lane keeps two positions in its neighbour's segment across the events of
// the pixel: p = newest entry older than the event, q = oldest entry inside the event's time window.  The sources are the
// entries q..p, newest first -- counted by subtraction, not by walking -- and per event each position advances by about one
// entry (DEEP_WIN entries are looked at per round trip).  The first event of a group in a pixel finds both positions by
// binary search.  Same rows as ell_pix_pass: the sums over the lower lanes, the 16-slot cut and the slot order are its code.
#define DEEP_WIN 4
template <int GS, bool INT_CMP>
__device__ __forceinline__ void ell_pix_pass_deep(const int2* __restrict__ ev2, int N, int Q, int dt_max, int* my_src,
                                                  unsigned short* my_stp, const int2* my_dest, int* my_cnt, int st, int en,
                                                  int& p, int& q, bool& fresh) {
    constexpr int MAXNB = 16, NG = 64 / GS;
    struct __attribute__((aligned(8))) Piece { int4 v; };
    const int lane = threadIdx.x & 63, g = lane / GS, l = lane % GS;
    const float fdt = (float)dt_max;
    const int lo = max(st, en - Q);                          // the FIFO keeps only the newest Q ids
    unsigned long long todo = __ballot(lane < PIX_CHUNK && my_cnt[lane & (PIX_CHUNK - 1)] < MAXNB);
    while (todo) {
        int d = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
#pragma unroll
        for (int k = 1; k < NG; ++k) {
            const int dk = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            d = g == k ? dk : d;
        }
        const bool act = d >= 0;
        d = max(d, 0);
        const int base = my_cnt[d];
        const int2 mine = my_dest[d];
        const int me = mine.x, ts = mine.y, t_min = ts - dt_max;
        bool more = act;
        if (__any(fresh && act)) {
            // a = first entry that is not older than the event, a2 = first entry inside its time window
            const bool go = fresh && act;
            int a = go ? lo : 0, b = go ? en : 0, a2 = a, b2 = b;
            while (__any(a < b || a2 < b2)) {
                if (a < b) { const int m = (a + b) >> 1; if (ev2[m].x < me) a = m + 1; else b = m; }
                if (a2 < b2) {
                    const int m = (a2 + b2) >> 1, t = ev2[m].y;
                    if (INT_CMP ? t < t_min : (float)(ts - t) > fdt) a2 = m + 1; else b2 = m;
                }
            }
            if (go) { p = a - 1; q = a2; fresh = false; more = false; }
        }
        while (__any(more)) {                                // both positions move up past the entries this event leaves behind
            if (more) {
                bool okp = true, okq = true;
                int advp = 0, advq = 0;
                int4 rp[DEEP_WIN / 2], rq[DEEP_WIN / 2];
#pragma unroll
                for (int k = 0; k < DEEP_WIN / 2; ++k) {     // 16-byte loads of two entries each, every address inside the array
                    rp[k] = ((const Piece*)(ev2 + min(p + 1 + 2 * k, N - 2)))->v;
                    rq[k] = ((const Piece*)(ev2 + min(q + 2 * k, N - 2)))->v;
                }
#pragma unroll
                for (int k = 0; k < DEEP_WIN / 2; ++k) {
                    const int ip = p + 1 + 2 * k, iq = q + 2 * k;
                    const int id0 = ip > N - 2 ? rp[k].z : rp[k].x, id1 = rp[k].z;       // entries ip and ip + 1
                    const int t0 = iq > N - 2 ? rq[k].w : rq[k].y, t1 = rq[k].w;          // entries iq and iq + 1
                    okp = okp && ip < en && id0 < me;          advp += okp ? 1 : 0;
                    okp = okp && ip + 1 < en && id1 < me;      advp += okp ? 1 : 0;
                    okq = okq && iq < en && (INT_CMP ? t0 < t_min : (float)(ts - t0) > fdt);       advq += okq ? 1 : 0;
                    okq = okq && iq + 1 < en && (INT_CMP ? t1 < t_min : (float)(ts - t1) > fdt);   advq += okq ? 1 : 0;
                }
                p += advp; q += advq;
                more = okp || okq;                           // every entry looked at was left behind: look further
            }
        }
        const int nq = act ? min(MAXNB, max(0, p - q + 1)) : 0;
        int inc = nq;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
        if (GS >= 32) inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);    // row_bcast15 -> rows 1 and 3
        if (GS == 64) inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);    // row_bcast31 -> rows 2 and 3
        const int off = inc - nq;
        const int take = min(nq, MAXNB - base - off);        // slots left for this lane's sources
        for (int k = 0; k < take; ++k) {
            my_src[d * MAXNB + base + off + k] = p - k;
            my_stp[d * MAXNB + base + off + k] = (unsigned short)l;
        }
        if (l == GS - 1 && act) my_cnt[d] = min(MAXNB, base + inc);      // the group's last lane holds the total
    }
    PIX_LDS_ORDER();
}

// Batches whose timestamps are not monotone in the id (rare): plain per-event walk over the events [ds, de) of the pixel
// (x, y), one event per lane, rows written slot by slot; the reference's `continue` on a too-old entry matters here.
__device__ __forceinline__ void pix_walk_events(const int2* __restrict__ ev2, const unsigned* __restrict__ start, int ds, int de, int x,
                                                int y, int plane_base, int W, int H, int steps, int sp0, int Q, int dt_max,
                                                int* __restrict__ ell_src, uint16_t* __restrict__ ell_step, int* __restrict__ deg) {
    constexpr int MAXNB = 16;
    const int lane = threadIdx.x & 63;
    for (int r = ds + lane; r < de; r += 64) {
        const int2 mine = ev2[r];
        int* orow = ell_src + (size_t)r * MAXNB;
        uint16_t* srow = ell_step + (size_t)r * MAXNB;
        for (int e = 1; e < MAXNB; ++e) { orow[e] = 0; srow[e] = 0; }
        orow[0] = r; srow[0] = 0;
        int nn = 1;
        for (int i = 0; i < steps && nn < MAXNB; ++i) {
            const int xn = x + c_spiral[2 * (sp0 + i)], yn = y + c_spiral[2 * (sp0 + i) + 1];
            if (xn < 0 || yn < 0 || xn >= W || yn >= H) continue;
            const int q = plane_base + xn + W * yn;
            const int st = (int)start[q], en = (int)start[q + 1];
            const int lo = max(st, en - Q);
            for (int s = en - 1; s >= lo; --s) {
                const int2 c = ev2[s];
                if (mine.x > c.x) {
                    if ((float)(mine.y - c.y) > (float)dt_max) continue;
                    orow[nn] = s; srow[nn] = (uint16_t)i;
                    if (++nn >= MAXNB) break;
                }
            }
        }
        deg[r] = nn;
    }
}

// Workgroup = 4 waves on a strip of PIX_STRIP pixels of one sensor row: at any moment the four waves work on four adjacent
// pixels, whose 7x7 (...) neighbourhoods are nearly the same cache lines, so most of a wave's gathers hit the CU's L1 (one
// wave per workgroup, dealt pixel by pixel over all CUs, missed it on nearly every line and waited for L2).  Strips are
// numbered so that each XCD (workgroup id mod 8) owns one contiguous eighth of them: its L2 then holds one band of rows.
#define PIX_STRIP 16
template <int GS1, bool INT_CMP, int DEPTH, bool DEEP = false>
__global__ __launch_bounds__(256, DEPTH <= 2 ? 8 : 5) void k_build_ell_pix(const int2* __restrict__ ev2, const unsigned* __restrict__ start,
                                                      const unsigned* __restrict__ flags_in, int N, int n_strips, int W, int H, int steps,
                                                      int sp0, int Q, int dt_max, int* __restrict__ ell_src,
                                                      uint16_t* __restrict__ ell_step, int* __restrict__ deg) {
    constexpr int MAXNB = 16;
    static_assert(PIX_CHUNK == 16, "one int4 per lane covers the chunk's rows");
    __shared__ int s_src[4][PIX_CHUNK * MAXNB];
    __shared__ unsigned short s_stp[4][PIX_CHUNK * MAXNB];
    __shared__ int2 s_dest[4][PIX_CHUNK];
    __shared__ int s_cnt[4][PIX_CHUNK];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (uniform: a pixel's own bounds are a scalar load)
    int* my_src = s_src[wv]; unsigned short* my_stp = s_stp[wv]; int2* my_dest = s_dest[wv]; int* my_cnt = s_cnt[wv];
    const int per_xcd = (n_strips + 7) >> 3;
    const int strip = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (strip >= n_strips) return;                           // (no workgroup barrier below: the waves are independent)
    const int strips_x = (W + PIX_STRIP - 1) / PIX_STRIP;
    const int row = strip / strips_x, y = row % H, plane_base = (row / H) * W * H, x0 = (strip % strips_x) * PIX_STRIP;
    const bool walk = (flags_in[0] & EVAD_FLAG_T_NOT_MONOTONE) != 0;
    // A wave's time goes into waiting for dependent loads (own segment -> neighbour bounds -> entries), not into arithmetic:
    // the first pass's spiral offsets are the same for every pixel and are read once, and the bounds a pixel needs first
    // (its own segment, its neighbours') are requested one pixel ahead.
    const int first = min(GS1, steps), l1 = lane % GS1, ii1 = min(l1, steps - 1);
    const int sdx = c_spiral[2 * (sp0 + ii1)], sdy = c_spiral[2 * (sp0 + ii1) + 1];
    const int xend = min(x0 + PIX_STRIP, W);
    int x = x0 + wv;
    if (x >= xend) return;
    int st_n, en_n;                                          // neighbour bounds and own segment of the wave's next pixel, in flight
    pix_bounds(start, x, y, plane_base, W, H, sdx, sdy, l1 < first, st_n, en_n);
    PixBounds own_n = *(const PixBounds*)(start + plane_base + x + W * y);
    for (; x < xend; x += 4) {
        const int st1 = st_n, en1 = en_n;
        const int ds = __builtin_amdgcn_readfirstlane((int)own_n.first), de = __builtin_amdgcn_readfirstlane((int)own_n.end);
        if (x + 4 < xend) {
            pix_bounds(start, x + 4, y, plane_base, W, H, sdx, sdy, l1 < first, st_n, en_n);
            own_n = *(const PixBounds*)(start + plane_base + x + 4 + W * y);
        }
        if (de <= ds) continue;
        if (walk) { pix_walk_events(ev2, start, ds, de, x, y, plane_base, W, H, steps, sp0, Q, dt_max, ell_src, ell_step, deg); continue; }
        // (DEEP) positions in the first pass's neighbour segments, kept from event to event of the pixel
        const int lo1 = max(st1, en1 - Q);
        int dp_p = lo1 - 1, dp_q = lo1;
        bool dp_fresh = true;
        for (int d0 = ds; d0 < de; d0 += PIX_CHUNK) {        // the pixel's destination events, PIX_CHUNK at a time
            const int md = min(PIX_CHUNK, de - d0);
            if (lane < PIX_CHUNK) {
                my_dest[lane] = ev2[min(d0 + lane, N - 1)];
                my_cnt[lane] = lane < md ? 1 : MAXNB;        // sources found so far for event d0 + lane (self edge first)
            }
            ((int4*)my_src)[lane] = make_int4(0, 0, 0, 0);   // PIX_CHUNK * 16 ints = 64 int4
            if (lane < PIX_CHUNK * 2) ((uint4*)my_stp)[lane] = make_uint4(0u, 0u, 0u, 0u);
            PIX_LDS_ORDER();
            if (lane < PIX_CHUNK) my_src[lane * MAXNB] = d0 + lane;
            PIX_LDS_ORDER();
            if (DEEP) ell_pix_pass_deep<GS1, INT_CMP>(ev2, N, Q, dt_max, my_src, my_stp, my_dest, my_cnt, st1, en1, dp_p, dp_q, dp_fresh);
            else ell_pix_pass<GS1, INT_CMP, true, DEPTH>(ev2, start, N, W, H, first, sp0, Q, dt_max, x, y, plane_base, 0, md, my_src, my_stp, my_dest, my_cnt, st1, en1);
            for (int i0 = first; i0 < steps; i0 += 64) {
                if (!__ballot(lane < PIX_CHUNK && my_cnt[lane & (PIX_CHUNK - 1)] < MAXNB)) break;
                ell_pix_pass<64, INT_CMP, false, DEPTH>(ev2, start, N, W, H, steps, sp0, Q, dt_max, x, y, plane_base, i0, md, my_src, my_stp, my_dest, my_cnt, 0, 0);
            }
            if (lane < md * 4) ((int4*)(ell_src + (size_t)d0 * MAXNB))[lane] = ((const int4*)my_src)[lane];
            if (lane < md * 2) ((uint4*)(ell_step + (size_t)d0 * MAXNB))[lane] = ((const uint4*)my_stp)[lane];
            if (lane < md) deg[d0 + lane] = my_cnt[lane];
            PIX_LDS_ORDER();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-per-pixel search for LARGE windows at LOW density (the 1280x720 geometry: radius 13 = 729 spiral steps, half an event
// per pixel).  With lanes = consecutive spiral steps, the 64 neighbour pixels of a pass lie on ring sides: the vertical
// sides are 64 different cache lines, for the bounds and again for the entries, so the kernel is bound by the number of
// line requests (holding four spiral steps per lane, i.e. four times the loads in flight, made it slower: 6.7 against
// 5.0 ms on 4 M events), and more than half of the pixels a wave visits are empty.
// Here the lanes of a wave take the pixels of a RING BAND (whole rings, at most 256 pixels: rings 0..7, 8..10, 11..12, 13
// at radius 13) in ROW-MAJOR order, PR_SPL pixels per lane.  The bounds of a sensor row's pixels are consecutive words and
// their FIFO entries consecutive records of the pixel-sorted array, so a wave-wide load touches a handful of lines
// instead of one per lane.  The price is that lane order is no longer spiral order: per event the lanes drop their
// source counts into a per-wave LDS table indexed by spiral step, the wave scans that table (four steps per lane) and
// every pixel picks up the exclusive sum of its step -- the slot where its sources go, newest first, until 16 are taken.
// Most events are complete after the first band (225 pixels); only the early events of a clip go on.  Each wave first
// looks at the own segments of its strip's 16 pixels (one load) and takes every fourth NON-EMPTY pixel; neighbour bounds
// are requested, one pixel ahead, only for pixels that hold events.  One event at a time per wave (a pixel holds 1.3 on
// average).  A lane keeps the PR_DEPTH newest entries of a pixel in registers; deeper pixels are read on from memory.
#define PR_SPL 4
#define PR_DEPTH 2
#ifndef PR_WAVES
#define PR_WAVES 4
#endif
// bounds of the PR_SPL pixels of this lane in a band: entry idx = k * 64 + lane of the band's row-major list
__device__ __forceinline__ void pixr_bounds(const unsigned* __restrict__ start, int x, int y, int plane_base, int W, int H,
                                            const unsigned (&ent)[PR_SPL], int (&st)[PR_SPL], int (&en)[PR_SPL]) {
#pragma unroll
    for (int k = 0; k < PR_SPL; ++k) {
        const bool live = ent[k] != 0xffffffffu;
        pix_bounds(start, x, y, plane_base, W, H, (int)(ent[k] & 255u) - 128, (int)((ent[k] >> 8) & 255u) - 128, live, st[k], en[k]);
    }
}

template <bool INT_CMP>
__device__ __forceinline__ void ell_pixr_pass(const int2* __restrict__ ev2, int N, int Q, int dt_max, int step0,
                                              const unsigned (&ent)[PR_SPL], const int (&st)[PR_SPL], const int (&en)[PR_SPL],
                                              int* my_src, unsigned short* my_stp, const int2* my_dest, int* my_cnt,
                                              unsigned* my_nq /* [64] words = 256 step counts */, unsigned* my_pre /* [64] */) {
    constexpr int MAXNB = 16;
    const int lane = threadIdx.x & 63;
    const float fdt = (float)dt_max;
    int lo[PR_SPL], lenr[PR_SPL], eold[PR_SPL], eid[PR_SPL][PR_DEPTH], et[PR_SPL][PR_DEPTH];
    bool plain = N >= PR_DEPTH;
#pragma unroll
    for (int k = 0; k < PR_SPL; ++k) {
        lo[k] = max(st[k], en[k] - Q);                       // the FIFO keeps only the newest Q ids
        lenr[k] = min(en[k] - lo[k], PR_DEPTH);
        plain = plain && (en[k] >= PR_DEPTH || en[k] == lo[k]);
    }
    // the PR_DEPTH newest entries end at en - 1: 16-byte loads of two entries each, only the pairs a segment has (an odd
    // length reads one entry of the neighbouring segment, harmless); only a wave that touches the first PR_DEPTH events of
    // the whole array clamps every index instead
    if (__all(plain)) {
        struct __attribute__((aligned(8))) Piece { int4 v; };
#pragma unroll
        for (int k = 0; k < PR_SPL; ++k) {
            const Piece* top = (const Piece*)(ev2 + (lenr[k] > 0 ? en[k] - PR_DEPTH : 0));
            int4 r1 = make_int4(0, 0, 0, 0);
            if (lenr[k] > 0) r1 = top[PR_DEPTH / 2 - 1].v;   // entries en - 2 (older) and en - 1
            eid[k][0] = r1.z; et[k][0] = r1.w; eid[k][1] = r1.x; et[k][1] = r1.y;
            if (PR_DEPTH == 4) {
                int4 r0 = make_int4(0, 0, 0, 0);
                if (lenr[k] > 2) r0 = top[0].v;              // entries en - 4 and en - 3
                eid[k][PR_DEPTH - 2] = r0.z; et[k][PR_DEPTH - 2] = r0.w; eid[k][PR_DEPTH - 1] = r0.x; et[k][PR_DEPTH - 1] = r0.y;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < PR_SPL; ++k)
#pragma unroll
            for (int v = 0; v < PR_DEPTH; ++v) { const int2 c = ev2[min(max(en[k] - 1 - v, lo[k]), max(N - 1, 0))]; eid[k][v] = c.x; et[k][v] = c.y; }
    }
#pragma unroll
    for (int k = 0; k < PR_SPL; ++k) {
        // padding (entries past the segment's length): older than any event and out of any time window, so it is counted
        // neither as "not older" nor as a source
#pragma unroll
        for (int v = 0; v < PR_DEPTH; ++v) { if (v >= lenr[k]) { eid[k][v] = (int)0x80000000; et[k][v] = -(1 << 30); } }
        // oldest entry of a deep pixel: if even that one is not older than the event, nothing below the registers is
        eold[k] = en[k] - lo[k] > PR_DEPTH ? ev2[min(lo[k], N - 1)].x : 0x7fffffff;
    }
    my_nq[lane] = 0u;                                        // steps past the band's last one count nothing
    PIX_LDS_ORDER();
    unsigned char* nq_b = (unsigned char*)my_nq;
    const unsigned char* pre_b = (const unsigned char*)my_pre;
    unsigned long long todo = __ballot(lane < PIX_CHUNK && my_cnt[lane & (PIX_CHUNK - 1)] < MAXNB);   // bit d: event d still lacks sources
    while (todo) {
        const int d = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int base = my_cnt[d];
        const int2 mine = my_dest[d];
        const int me = mine.x, ts = mine.y, t_min = ts - dt_max;
        int nnew[PR_SPL], nq[PR_SPL];
        unsigned undecided = 0;
#pragma unroll
        for (int k = 0; k < PR_SPL; ++k) {
            int cge = 0, nin = 0;
#pragma unroll
            for (int v = 0; v < PR_DEPTH; ++v) {
                cge += eid[k][v] >= me ? 1 : 0;
                if (INT_CMP) nin += et[k][v] >= t_min ? 1 : 0;
                else         nin += (float)(ts - et[k][v]) > fdt ? 0 : 1;
            }
            nnew[k] = cge;
            nq[k] = max(0, nin - cge);
            if (eold[k] < me && cge + nq[k] == PR_DEPTH) undecided |= 1u << k;          // deep pixel, registers undecided
        }
        while (__any(undecided != 0)) {                      // rare: read on from memory, one pixel of a lane at a time
            if (undecided) {
                const int k = __ffs((int)undecided) - 1;
                undecided &= undecided - 1;
                const int enk = k == 0 ? en[0] : k == 1 ? en[1] : k == 2 ? en[2] : en[3];
                const int lok = k == 0 ? lo[0] : k == 1 ? lo[1] : k == 2 ? lo[2] : lo[3];
                int nn = k == 0 ? nnew[0] : k == 1 ? nnew[1] : k == 2 ? nnew[2] : nnew[3];
                int qq = k == 0 ? nq[0] : k == 1 ? nq[1] : k == 2 ? nq[2] : nq[3];
                bool stop = false;
                for (int s = enk - 1 - PR_DEPTH; s >= lok && qq < MAXNB && !stop; s -= 4) {
                    int2 c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) c[u] = ev2[max(s - u, lok)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (stop || s - u < lok || qq >= MAXNB) continue;
                        if (c[u].x >= me) { ++nn; continue; }
                        const int dt = ts - c[u].y;
                        if (INT_CMP ? dt > dt_max : (float)dt > fdt) stop = true; else ++qq;
                    }
                }
#pragma unroll
                for (int j = 0; j < PR_SPL; ++j) if (j == k) { nnew[j] = nn; nq[j] = qq; }
            }
        }
#pragma unroll
        for (int k = 0; k < PR_SPL; ++k)
            if (ent[k] != 0xffffffffu) nq_b[(int)(ent[k] >> 16) - step0] = (unsigned char)nq[k];      // every step of the band is some lane's pixel
        PIX_LDS_ORDER();
        // the band's step table in spiral order, four steps per lane: inclusive sum over the wave (DPP row scans, then row
        // broadcasts), exclusive sums back into the table (clamped to a byte: only sums below 16 matter)
        const unsigned c4 = my_nq[lane];
        const int c0 = (int)(c4 & 255u), c1 = (int)((c4 >> 8) & 255u), c2 = (int)((c4 >> 16) & 255u), c3 = (int)(c4 >> 24);
        const int tot = c0 + c1 + c2 + c3;
        int inc = tot;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);    // row_bcast15 -> rows 1 and 3
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);    // row_bcast31 -> rows 2 and 3
        const int e0 = min(inc - tot, 255), e1 = min(e0 + c0, 255), e2 = min(e1 + c1, 255), e3 = min(e2 + c2, 255);
        my_pre[lane] = (unsigned)e0 | (unsigned)e1 << 8 | (unsigned)e2 << 16 | (unsigned)e3 << 24;
        PIX_LDS_ORDER();
#pragma unroll
        for (int k = 0; k < PR_SPL; ++k) {
            if (nq[k] > 0 && ent[k] != 0xffffffffu) {
                const int pos = base + (int)pre_b[(int)(ent[k] >> 16) - step0];
                const int take = min(nq[k], MAXNB - pos);    // slots left for this pixel's sources
                for (int u = 0; u < take; ++u) {
                    my_src[d * MAXNB + pos + u] = en[k] - 1 - nnew[k] - u;
                    my_stp[d * MAXNB + pos + u] = (unsigned short)(ent[k] >> 16);
                }
            }
        }
        if (lane == 63) my_cnt[d] = min(MAXNB, base + inc);  // the last lane holds the total
        PIX_LDS_ORDER();
    }
}

template <bool INT_CMP>
__global__ __launch_bounds__(256, PR_WAVES) void k_build_ell_pixr(const int2* __restrict__ ev2, const unsigned* __restrict__ start,
                                                       const unsigned* __restrict__ flags_in, const unsigned* __restrict__ band,
                                                       BandPasses passes, int N, int n_strips, int W, int H, int Q, int dt_max,
                                                       int* __restrict__ ell_src, uint16_t* __restrict__ ell_step, int* __restrict__ deg) {
    constexpr int MAXNB = 16;
    __shared__ int s_src[4][PIX_CHUNK * MAXNB];
    __shared__ unsigned short s_stp[4][PIX_CHUNK * MAXNB];
    __shared__ int2 s_dest[4][PIX_CHUNK];
    __shared__ int s_cnt[4][PIX_CHUNK];
    __shared__ unsigned s_nq[4][64], s_pre[4][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* my_src = s_src[wv]; unsigned short* my_stp = s_stp[wv]; int2* my_dest = s_dest[wv]; int* my_cnt = s_cnt[wv];
    const int per_xcd = (n_strips + 7) >> 3;
    const int strip = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (strip >= n_strips) return;                           // (no workgroup barrier below: the waves are independent)
    // (batches whose timestamps are not monotone are left to the per-event walk, launched behind this kernel)
    if (flags_in[0] & EVAD_FLAG_T_NOT_MONOTONE) return;
    const int strips_x = (W + PIX_STRIP - 1) / PIX_STRIP;
    const int row = strip / strips_x, y = row % H, plane_base = (row / H) * W * H, x0 = (strip % strips_x) * PIX_STRIP;
    const int npx = min(PIX_STRIP, W - x0);
    // own segments of the strip's pixels: lane l holds the start of pixel x0 + l (lane npx: the end of the last one)
    const int sb = (int)start[plane_base + W * y + x0 + min(lane, npx)];
    const int sb1 = __shfl_down(sb, 1, 64);
    const unsigned occupied = (unsigned)__ballot(lane < npx && sb1 > sb);
    unsigned mine_px = 0;                                    // this wave's pixels: every fourth non-empty one
    {
        int rank = 0;
        for (unsigned m = occupied; m; m &= m - 1, ++rank) if ((rank & 3) == wv) mine_px |= m & (0u - m);
    }
    if (!mine_px) return;
    // the first band's pixel list is the same for every pixel: read once
    unsigned ent0[PR_SPL];
#pragma unroll
    for (int k = 0; k < PR_SPL; ++k) ent0[k] = k * 64 + lane < passes.cnt[0] ? band[passes.off[0] + k * 64 + lane] : 0xffffffffu;
    int b = __ffs((int)mine_px) - 1;
    mine_px &= mine_px - 1;
    int st_n[PR_SPL], en_n[PR_SPL];                          // first-band bounds of the wave's next pixel, in flight
    pixr_bounds(start, x0 + b, y, plane_base, W, H, ent0, st_n, en_n);
    while (true) {
        const int x = x0 + b;
        const int ds = __builtin_amdgcn_readlane(sb, b), de = __builtin_amdgcn_readlane(sb, b + 1);
        unsigned ent[PR_SPL]; int st1[PR_SPL], en1[PR_SPL];
#pragma unroll
        for (int k = 0; k < PR_SPL; ++k) { ent[k] = ent0[k]; st1[k] = st_n[k]; en1[k] = en_n[k]; }
        const int bn = mine_px ? __ffs((int)mine_px) - 1 : -1;
        mine_px &= mine_px - 1;                              // (0 stays 0)
        if (bn >= 0) pixr_bounds(start, x0 + bn, y, plane_base, W, H, ent0, st_n, en_n);
        for (int d0 = ds; d0 < de; d0 += PIX_CHUNK) {        // the pixel's destination events, PIX_CHUNK at a time
            const int md = min(PIX_CHUNK, de - d0);
            if (lane < PIX_CHUNK) {
                my_dest[lane] = ev2[min(d0 + lane, N - 1)];
                my_cnt[lane] = lane < md ? 1 : MAXNB;        // sources found so far for event d0 + lane (self edge first)
            }
            ((int4*)my_src)[lane] = make_int4(0, 0, 0, 0);   // PIX_CHUNK * 16 ints = 64 int4
            if (lane < PIX_CHUNK * 2) ((uint4*)my_stp)[lane] = make_uint4(0u, 0u, 0u, 0u);
            PIX_LDS_ORDER();
            if (lane < PIX_CHUNK) my_src[lane * MAXNB] = d0 + lane;
            PIX_LDS_ORDER();
            // band after band (ONE copy of the pass): the first with the bounds requested a pixel ahead, the others -- and the
            // first again for a pixel's second chunk of 16 events -- with bounds loaded here
#pragma unroll 1
            for (int p = 0; p < passes.n; ++p) {
                if (p > 0 || d0 > ds) {
                    if (p > 0 && !__ballot(lane < PIX_CHUNK && my_cnt[lane & (PIX_CHUNK - 1)] < MAXNB)) break;
#pragma unroll
                    for (int k = 0; k < PR_SPL; ++k) ent[k] = k * 64 + lane < passes.cnt[p] ? band[passes.off[p] + k * 64 + lane] : 0xffffffffu;
                    pixr_bounds(start, x, y, plane_base, W, H, ent, st1, en1);
                }
                ell_pixr_pass<INT_CMP>(ev2, N, Q, dt_max, passes.step0[p], ent, st1, en1, my_src, my_stp, my_dest, my_cnt, s_nq[wv], s_pre[wv]);
            }
            if (lane < md * 4) ((int4*)(ell_src + (size_t)d0 * MAXNB))[lane] = ((const int4*)my_src)[lane];
            if (lane < md * 2) ((uint4*)(ell_step + (size_t)d0 * MAXNB))[lane] = ((const uint4*)my_stp)[lane];
            if (lane < md) deg[d0 + lane] = my_cnt[lane];
            PIX_LDS_ORDER();
        }
        if (bn < 0) break;
        b = bn;
    }
}

extern "C" int eventad_build_ell(const int32_t* sid, const int32_t* ts_s, const int32_t* ev2, const int32_t* key_s,
                                 const uint32_t* pix_start, uint32_t* counters, int N, int B, int W, int H, int radius_px, int delta_t_us,
                                 int max_num_neighbors, int queue_size, int32_t* ell_src, uint16_t* ell_step,
                                 int32_t* deg, const int32_t* node_slot, const int32_t* node_vox, int nx, int ny,
                                 uint32_t* bitmap, int variant, int bitmap_prezeroed, void* stream) {
    if (!sid || !ts_s || !key_s || !pix_start || !counters || !ell_src || !ell_step || !deg || N < 0 || B <= 0 ||
        W <= 0 || H <= 0 || radius_px < 0 || max_num_neighbors < 1 || max_num_neighbors > 65535 || queue_size < 1 ||
        (bitmap && (!node_slot || !node_vox || nx <= 0 || ny <= 0 || nx > 255 || ny > 255)))
        return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (bitmap && !bitmap_prezeroed) evad_zero_async(bitmap, sizeof(uint32_t) * 3 * (size_t)nx * ny * (B + 1), st);
    if (N == 0) return evad_last_error();
    int steps; const SpiralDev* sp = nullptr;
    const bool fast = max_num_neighbors == 16 && ev2 && radius_px <= SPIRAL_MAX_R;
    if (ensure_spiral(radius_px, &steps, &sp)) return -2;
    if (steps > 65535) return EVAD_ERR_ARG;
    const int sp0 = fast ? spiral_base(radius_px) : 0;
    dim3 grid(evad_blocks(N, 256));
    if (fast && variant == 1 && (radius_px == 1 || radius_px == 3)) {
#define EVAD_RS(RR, MK) hipLaunchKernelGGL((k_build_ell_rowscan<RR, MK>), grid, dim3(256), 0, st, (const int2*)ev2, key_s, pix_start, counters, \
                                            N, W, H, steps, sp0, queue_size, delta_t_us, ell_src, ell_step, deg, node_slot, node_vox, bitmap, counters)
        if (radius_px == 1) { if (bitmap) EVAD_RS(1, true); else EVAD_RS(1, false); }
        else                { if (bitmap) EVAD_RS(3, true); else EVAD_RS(3, false); }
#undef EVAD_RS
    } else if (fast && !bitmap && (variant == 2 || variant == 4 || variant == 5 || variant == 6 || (variant == 0 && (double)N >= 0.4 * (double)B * W * H))) {
        // (measured crossover against the per-event walk: 0.3-0.4 events per pixel at radius 13, below 1.4 at radius 3)
        const long long P = (long long)B * W * H;            // wave per pixel
        const long long strips = (long long)B * H * ((W + PIX_STRIP - 1) / PIX_STRIP);
        if (P > 0x7fffffffLL || strips > 0x7ffffff0LL) return EVAD_ERR_ARG;
        const int n_strips = (int)strips, wgs = (n_strips + 7) / 8 * 8;        // every XCD the same number of workgroups
        const bool int_cmp = delta_t_us > -(1 << 24) + 1 && delta_t_us < (1 << 24) - 1;
        const bool very_dense = (double)N >= 5.0 * (double)P;
#define EVAD_PIX(G1, IC) hipLaunchKernelGGL((k_build_ell_pix<G1, IC, 8>), dim3((unsigned)wgs), dim3(256), 0, st, (const int2*)ev2, pix_start, \
                                            counters, N, n_strips, W, H, steps, sp0, queue_size, delta_t_us, ell_src, ell_step, deg)
        // radius <= 3: all 49 steps in one 64-lane pass, one event at a time (a second pass costs a pixel three dependent
        // round trips, an event more in the loop a few hundred cycles)
        // large windows at low density (or asked for: variant 4): ring bands in row-major lane order, non-empty pixels only
        // (measured equal to the spiral-order form at 1280x720 / 0.54 events per pixel -- 4.99 against 4.96 ms -- and slower at
        // 640x480 / 0.98: both forms are bound by instruction issue, ~600-900 per pixel, not by loads: on request only)
        // deep pixels (from 20 events per pixel, or asked for: variant 5): the first pass keeps two positions per neighbour
        // segment instead of the newest entries (ell_pix_pass_deep).  256x256, radius 3, 8 M events at 30 per pixel: 2.54 ms
        // against 3.03 for the register form with the same 16-step first pass; 0.69 against 0.57 at 9 per pixel; 640x480,
        // radius 7, 13 per pixel: 2.50 against 2.06
        const bool deep = N >= 2 && (variant == 5 || (variant == 0 && (double)N >= 20.0 * (double)P));
        if (deep) {
            if (int_cmp) hipLaunchKernelGGL((k_build_ell_pix<16, true, 8, true>), dim3((unsigned)wgs), dim3(256), 0, st, (const int2*)ev2, pix_start,
                                            counters, N, n_strips, W, H, steps, sp0, queue_size, delta_t_us, ell_src, ell_step, deg);
            else         hipLaunchKernelGGL((k_build_ell_pix<16, false, 8, true>), dim3((unsigned)wgs), dim3(256), 0, st, (const int2*)ev2, pix_start,
                                            counters, N, n_strips, W, H, steps, sp0, queue_size, delta_t_us, ell_src, ell_step, deg);
        }
        else if (sp->passes.n > 0 && variant == 4) {
            if (int_cmp) hipLaunchKernelGGL((k_build_ell_pixr<true>), dim3((unsigned)wgs), dim3(256), 0, st, (const int2*)ev2, pix_start, counters,
                                            sp->band, sp->passes, N, n_strips, W, H, queue_size, delta_t_us, ell_src, ell_step, deg);
            else         hipLaunchKernelGGL((k_build_ell_pixr<false>), dim3((unsigned)wgs), dim3(256), 0, st, (const int2*)ev2, pix_start, counters,
                                            sp->band, sp->passes, N, n_strips, W, H, queue_size, delta_t_us, ell_src, ell_step, deg);
            // non-monotone timestamps (flag on the device): the kernel above returned at once and this exact walk does the work;
            // otherwise its workgroups return at once
            hipLaunchKernelGGL((k_build_ell_reg<false, false>), grid, dim3(256), 0, st, (const int2*)ev2, key_s, pix_start, counters, N, W, H,
                               steps, sp0, queue_size, delta_t_us, ell_src, ell_step, deg, node_slot, node_vox, (unsigned*)nullptr, counters, 1);
        }
        else if (steps > 64 && (double)N < 1.0 * (double)P && int_cmp) {
            // sparse pixels (at most one event per pixel on average: a pixel rarely holds more than two): the two newest entries
            // of a neighbour pixel in registers instead of eight -- a quarter of the entry loads and compares per pass, 61
            // registers and 8 waves per SIMD: 4.96 -> 3.3 ms on 4 M events at 1280x720 (the kernel is bound by instruction
            // issue).  Four entries at 3 events per pixel (256x256, radius 3) lose: 0.49 against 0.41 ms, the reads from
            // memory for pixels deeper than the registers cost more than the compares saved.
            hipLaunchKernelGGL((k_build_ell_pix<64, true, 2>), dim3((unsigned)wgs), dim3(256), 0, st, (const int2*)ev2, pix_start,
                               counters, N, n_strips, W, H, steps, sp0, queue_size, delta_t_us, ell_src, ell_step, deg);
        }
        // from five events per pixel: four events at a time on 16 steps each, whatever the radius (the whole 49-step spiral
        // of radius 3 in one pass, one event at a time, is 0.41 against 0.56 ms at 3 events per pixel -- but 0.99 against 0.57
        // at 9 and 9.25 against 3.03 at 30); variant 6 asks for this form
        else if (variant == 6 || (variant == 0 && very_dense)) { if (int_cmp) EVAD_PIX(16, true); else EVAD_PIX(16, false); }
        else if (steps <= 64 || (double)N < 2.0 * (double)P) { if (int_cmp) EVAD_PIX(64, true); else EVAD_PIX(64, false); }      // (... or few events per pixel)
        else if (very_dense) { if (int_cmp) EVAD_PIX(16, true); else EVAD_PIX(16, false); }
        else                 { if (int_cmp) EVAD_PIX(32, true); else EVAD_PIX(32, false); }
#undef EVAD_PIX
    } else if (fast) {
        const bool by_rank = (double)N >= 2.0 * (double)B * W * H;          // (variant 3 keeps this walk for dense input)
        const int threads = by_rank ? ELL_TILE : 256;
        const dim3 tgrid(evad_blocks(N, threads));
#define EVAD_REG(MK, BR) hipLaunchKernelGGL((k_build_ell_reg<MK, BR>), tgrid, dim3(threads), 0, st, (const int2*)ev2, key_s, pix_start, \
                               counters, N, W, H, steps, sp0, queue_size, delta_t_us, ell_src, ell_step, deg, node_slot, node_vox, bitmap, counters, 0)
        if (bitmap) { if (by_rank) EVAD_REG(true, true); else EVAD_REG(true, false); }
        else        { if (by_rank) EVAD_REG(false, true); else EVAD_REG(false, false); }
#undef EVAD_REG
    } else {                                                 // any neighbour cap / radius: the reference-style walk
        if (bitmap) return EVAD_ERR_ARG;                     // (mark with eventad_pool_edges_mark_ell afterwards)
        hipLaunchKernelGGL(k_build_ell, grid, dim3(256), 0, st, sid, ts_s, key_s, pix_start, sp->table, counters, N, W, H,
                           steps, queue_size, delta_t_us, max_num_neighbors, ell_src, ell_step, deg);
    }
    return evad_last_error();
}

// ELL (sorted numbering) -> the reference's edge_index layout, event numbering: int64[2][16N]
// prefilled with -1, slot [i*16+k] (reference ev_graph.cu:44-45,71-72).  Used by parity tests and
// by callers that want `edge_index` itself.
__global__ void k_ell_to_edges(const int* __restrict__ ell_src, const int* __restrict__ deg, const int* __restrict__ sid,
                               int N, int max_nb, long long K, long long* __restrict__ edges) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * max_nb) return;
    int r = (int)(t / max_nb), k = (int)(t % max_nb);
    int i = sid[r];
    long long o = (long long)i * max_nb + k;
    if (k < deg[r]) { edges[o] = sid[ell_src[t]]; edges[o + K] = i; }
    else            { edges[o] = -1;              edges[o + K] = -1; }
}

extern "C" int eventad_ell_to_edges(const int32_t* ell_src, const int32_t* deg, const int32_t* sid, int N,
                                    int max_num_neighbors, int64_t K, int64_t* edges, void* stream) {
    if (!ell_src || !deg || !sid || !edges || N < 0 || K < (int64_t)N * max_num_neighbors) return EVAD_ERR_ARG;
    if (N == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_ell_to_edges, dim3(evad_blocks((long long)N * max_num_neighbors, 256)), dim3(256), 0,
                       (hipStream_t)stream, ell_src, deg, sid, N, max_num_neighbors, (long long)K, (long long*)edges);
    return evad_last_error();
}
