// eventad_amd/csrc/bwd.hip -- backward of the event-level LUT spline convolution (fine-tuning the DAGR backbone).
//
// Forward (reference MySplineConv, src/dagr/model/layers/spline_conv.py:39-78; csrc/layer1.hip):
//     out_i = sum_{j->i} LUT[cell_ji]^T x_j        (the self edge's matrix LUT[steps] = LUT[0] + W_root carries the root term)
// Backward, given g = dL/d out  (the reference gets both from autograd over its gather + scatter-add, spline_conv.py:44-47,68):
//     dx_j      = sum_{i : j->i} LUT[cell_ji] g_i                      (gather over the TRANSPOSED graph)
//     dLUT[c]   = sum_{j->i, cell_ji = c} x_j (x) g_i                  (one Cin x 16 outer-product sum per spiral step)
// and the host maps dLUT to dL/d weight[25, Cin, 16] and dL/d root through the (linear) LUT construction.
//
// No float atomics anywhere, so both results are bitwise reproducible:
//  * the transposed adjacency (out-edges per source) is built with integer atomics for the COUNTS only; the arrival order
//    they leave inside a source's list is then replaced by a rank sort on (destination, slot), so the list -- and with
//    it the summation order of dx -- is the same in every run;
//  * dLUT is accumulated in two ordered stages: each workgroup walks its destinations in order, every thread owning one
//    (channel, output) element of all matrices in a private LDS column; the workgroups' partial tables are then summed in
//    workgroup order.
#include "common.h"

int evad_exclusive_scan(const unsigned* in, unsigned* out, unsigned* tile_sums, long long n, hipStream_t st);   // graph.hip

// ------------------------------------------------------------------------------------------------ transposed adjacency
__global__ void k_out_count(const int* __restrict__ ell_src, const int* __restrict__ deg, int N, unsigned* __restrict__ cnt,
                            unsigned short* __restrict__ arrival) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * 16) return;
    const int i = (int)(t >> 4), slot = (int)(t & 15);
    if (slot >= deg[i]) return;
    const unsigned a = atomicAdd(&cnt[ell_src[t]], 1u);
    arrival[t] = (unsigned short)min(a, 65535u);
}

__global__ void k_out_scatter(const int* __restrict__ ell_src, const int* __restrict__ deg, int N, const unsigned* __restrict__ start,
                              const unsigned short* __restrict__ arrival, int* __restrict__ tmp) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * 16) return;
    const int i = (int)(t >> 4), slot = (int)(t & 15);
    if (slot >= deg[i]) return;
    tmp[start[ell_src[t]] + arrival[t]] = (int)t;          // key = destination * 16 + slot
}

// one thread per list position: rank of its key inside the source's list (lists are short: mean = mean degree)
__global__ void k_out_rank(const unsigned* __restrict__ start, const int* __restrict__ ell_src, const int* __restrict__ tmp,
                           int* __restrict__ out_edge, int N) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long long)start[N]) return;                  // E = start[N] is known on the device only
    const int key = tmp[p];
    const int src = ell_src[key];
    const unsigned s0 = start[src], s1 = start[src + 1];
    unsigned rank = 0;
    for (unsigned c = s0; c < s1; ++c) rank += (tmp[c] < key);
    out_edge[s0 + rank] = key;
}

// out_start u32[N+1], out_edge i32[E_max = 16N] (key = destination * 16 + slot, ascending inside a source's list).
// scratch: cnt u32[N+1], scan_tmp u32[N/2048+2], arrival u16[16N], tmp i32[16N].  A source may have at most 65535
// out-edges (its arrival index is kept in 16 bits).
extern "C" int eventad_ell_transpose(const int32_t* ell_src, const int32_t* deg, int N, uint32_t* out_start, int32_t* out_edge,
                                     uint32_t* cnt, uint32_t* scan_tmp, uint16_t* arrival, int32_t* tmp, void* stream) {
    if (!ell_src || !deg || !out_start || !out_edge || !cnt || !scan_tmp || !arrival || !tmp || N < 0) return EVAD_ERR_ARG;
    if (N == 0) return EVAD_OK;
    hipStream_t st = (hipStream_t)stream;
    evad_zero_async(cnt, sizeof(uint32_t) * ((size_t)N + 1), st);
    const long long T = (long long)N * 16;
    hipLaunchKernelGGL(k_out_count, dim3(evad_blocks(T, 256)), dim3(256), 0, st, ell_src, deg, N, cnt, (unsigned short*)arrival);
    int rc = evad_exclusive_scan(cnt, out_start, scan_tmp, N, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_out_scatter, dim3(evad_blocks(T, 256)), dim3(256), 0, st, ell_src, deg, N, out_start,
                       (const unsigned short*)arrival, tmp);
    // E is only known on the device (out_start[N]); launch over the bound 16N, positions past E fall out of every list
    hipLaunchKernelGGL(k_out_rank, dim3(evad_blocks(T, 256)), dim3(256), 0, st, out_start, ell_src, tmp, out_edge, N);
    return evad_last_error();
}

// ------------------------------------------------------------------------------------------------ dL/dx
// One quad per source; lane q of the quad owns input channels {4q..4q+3} (+ 16+q for Cin = 20), exactly as the forward
// kernel, and reads its rows of LUT[cell] (row order [i][j], see eventad_amd/spline.py::quad_row_order) from LDS.
template <int CINP, bool LUT_IN_LDS>      // LUT_IN_LDS = false (search radii whose table exceeds LDS): rows are read from memory
__global__ __launch_bounds__(256) void k_conv_bwd_data(const float* __restrict__ g, const unsigned* __restrict__ out_start,
                                                       const int* __restrict__ out_edge, const uint16_t* __restrict__ ell_step,
                                                       const float* __restrict__ lut, float* __restrict__ dx, int N, int steps) {
    extern __shared__ __attribute__((aligned(16))) float s_lut[];
    constexpr int KL = CINP == 20 ? 5 : 4, LS = CINP * 16;
    if (LUT_IN_LDS) {
        for (int t = threadIdx.x; t < (steps + 1) * LS / 4; t += 256) ((float4*)s_lut)[t] = ((const float4*)lut)[t];
        __syncthreads();
    }
    const int j = blockIdx.x * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
    if (j >= N) return;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned p0 = out_start[j], p1 = out_start[j + 1];
    for (unsigned p = p0; p < p1; ++p) {
        const int key = out_edge[p];
        const int i = key >> 4, slot = key & 15;
        const int cell = slot == 0 ? steps : (int)ell_step[key];
        const float4* gr = (const float4*)(g + (size_t)i * 16);
        const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3];
        const float gv[16] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w};
        const float* w = (LUT_IN_LDS ? s_lut : lut) + cell * LS + q * 16;
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < 16; ++o) s = fmaf(w[k * 64 + o], gv[o], s);
            acc[k] += s;
        }
    }
    float* d = dx + (size_t)j * CINP;
    *(float4*)(d + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (CINP == 20) d[16 + q] = acc[4];
}

// lut f32[steps+1][Cin_pad][16] in the [i][j] row order of the forward kernel (row `steps` = self-edge matrix); g f32[N,16];
// dx f32[N, Cin_pad].  cin_pad = 20 or 16.  The table is staged in LDS when its (steps + 1) * Cin_pad * 64 bytes fit in 150 KB
// (radius <= 4 / 5 px), else read from memory.
extern "C" int eventad_l1_conv_bwd_data(const float* g, const uint32_t* out_start, const int32_t* out_edge, const uint16_t* ell_step,
                                        const float* lut, float* dx, int N, int steps, int cin_pad, void* stream) {
    if (!g || !out_start || !out_edge || !ell_step || !lut || !dx || N < 0 || steps < 1 || (cin_pad != 20 && cin_pad != 16)) return EVAD_ERR_ARG;
    const size_t lds = sizeof(float) * (size_t)(steps + 1) * cin_pad * 16;
    if (N == 0) return EVAD_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(evad_blocks(N, 64)), block(256);
    if (lds > 150 * 1024) {                                  // the table does not fit in LDS: read its rows from memory (L2)
        if (cin_pad == 20) hipLaunchKernelGGL((k_conv_bwd_data<20, false>), grid, block, 0, st, g, out_start, out_edge, ell_step, lut, dx, N, steps);
        else               hipLaunchKernelGGL((k_conv_bwd_data<16, false>), grid, block, 0, st, g, out_start, out_edge, ell_step, lut, dx, N, steps);
        return evad_last_error();
    }
    if (cin_pad == 20) {
        auto kern = k_conv_bwd_data<20, true>;
        if (evad_allow_lds((const void*)kern, lds)) return -2;
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, out_start, out_edge, ell_step, lut, dx, N, steps);
    } else {
        auto kern = k_conv_bwd_data<16, true>;
        if (evad_allow_lds((const void*)kern, lds)) return -2;
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, out_start, out_edge, ell_step, lut, dx, N, steps);
    }
    return evad_last_error();
}

// ------------------------------------------------------------------------------------------------ dL/dLUT
// dLUT[c] = sum over the edges with spiral step c of x_src (x) g_dst: per step an [edges_c x Cin]^T . [edges_c x 16] product.
// A workgroup takes 256 consecutive destinations (thread t = destination t of the chunk, its ELL row = 16 slots):
//  (1) stable counting sort of the chunk's valid slots by step, in LDS: per-thread counts per step, a prefix over the
//      threads per step (so the order inside a step is ascending slot id: a fixed summation order), placement;
//  (2) the 4 waves take the steps round-robin; a step's edges go through fp32 MFMA 16x16x4 four at a time -- lane l holds
//      channel l & 15 of the source row and output l & 15 of the gradient row of edge l >> 4, straight from memory -- into one
//      (Cin = 16) or two (Cin = 20: channels 16..19 in a second tile) accumulators, written once per step to the
//      workgroup's partial table.
// Stage 2 sums the partial tables in workgroup order.  (The previous form kept the accumulators in LDS and paid a
// read-modify-write plus three operand reads per FMA: 1.85 ms per convolution at 400k events, against 0.1 ms of traffic.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define BW_CHUNK 256
template <int CINP>
__global__ __launch_bounds__(256) void k_conv_bwd_weight(const float* __restrict__ x, const float* __restrict__ g,
                                                         const int* __restrict__ ell_src, const uint16_t* __restrict__ ell_step,
                                                         const int* __restrict__ deg, float* __restrict__ partial, int N, int steps,
                                                         int band) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int all_bins = steps + 1, T = CINP * 16;
    // The LDS sort takes `band` spiral steps at a time (all of them up to radius 7; at radius 13 the 730 bins go in four
    // bands): the chunk's ELL rows stay in registers, each band counts, sorts and multiplies the slots whose step lies in it.
    unsigned short* s_hist = (unsigned short*)smem;                          // [band][256]: counts, then exclusive prefixes over threads
    unsigned short* s_sorted = s_hist + (size_t)band * BW_CHUNK;            // [4096]: slot index inside the chunk, by step
    int* s_src = (int*)(s_sorted + BW_CHUNK * 16);                          // [4096]: source node of the slot at that position
    unsigned* s_start = (unsigned*)(s_src + BW_CHUNK * 16);                 // [band + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BW_CHUNK, i = n0 + tid;
    const int d = i < N ? min(deg[i], 16) : 0;
    unsigned short cells[16];
    int srcs[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { cells[k] = 0; srcs[k] = 0; }
    if (d > 0) {
        const int4* srow = (const int4*)(ell_src + (size_t)i * 16);
        const int4 q0 = srow[0], q1 = srow[1], q2 = srow[2], q3 = srow[3];
        srcs[0] = q0.x; srcs[1] = q0.y; srcs[2] = q0.z; srcs[3] = q0.w; srcs[4] = q1.x; srcs[5] = q1.y; srcs[6] = q1.z; srcs[7] = q1.w;
        srcs[8] = q2.x; srcs[9] = q2.y; srcs[10] = q2.z; srcs[11] = q2.w; srcs[12] = q3.x; srcs[13] = q3.y; srcs[14] = q3.z; srcs[15] = q3.w;
        const uint4* row = (const uint4*)(ell_step + (size_t)i * 16);       // 16 x u16 = 32 bytes
        const uint4 r0 = row[0], r1 = row[1];
        const unsigned w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) cells[k] = (unsigned short)(k == 0 ? steps : min((int)((w[k >> 1] >> (16 * (k & 1))) & 0xffffu), steps - 1));
    }
    for (int cb = 0; cb < all_bins; cb += band) {
    const int bins = min(band, all_bins - cb);
    for (int c = 0; c < bins; ++c) s_hist[c * BW_CHUNK + tid] = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < d && cells[k] >= cb && cells[k] < cb + bins) s_hist[(cells[k] - cb) * BW_CHUNK + tid] += 1;   // private column: no atomics
    __syncthreads();
    // per step: exclusive prefix of the counts over the 256 threads (lane = 4 consecutive threads), step totals
    for (int c = wave; c < bins; c += 4) {
        unsigned short* h = s_hist + c * BW_CHUNK + 4 * lane;
        const unsigned a0 = h[0], a1 = h[1], a2 = h[2], a3 = h[3], sum = a0 + a1 + a2 + a3;
        unsigned inc = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const unsigned t = __shfl_up(inc, dd, 64); if (lane >= dd) inc += t; }
        const unsigned ex = inc - sum;
        h[0] = (unsigned short)ex; h[1] = (unsigned short)(ex + a0); h[2] = (unsigned short)(ex + a0 + a1); h[3] = (unsigned short)(ex + a0 + a1 + a2);
        if (lane == 63) s_start[c + 1] = inc;                               // total of the step (scanned below)
    }
    __syncthreads();
    if (wave == 0) {                                                         // exclusive scan of the step totals
        unsigned carry = 0;
        for (int c0 = 0; c0 < bins; c0 += 64) {
            const int c = c0 + lane;
            const unsigned v = c < bins ? s_start[c + 1] : 0u;
            unsigned inc = v;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const unsigned t = __shfl_up(inc, dd, 64); if (lane >= dd) inc += t; }
            if (c < bins) s_start[c + 1] = carry + inc;
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) s_start[0] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < d && cells[k] >= cb && cells[k] < cb + bins) {
            const int c = cells[k] - cb;
            const unsigned p = s_start[c] + s_hist[c * BW_CHUNK + tid];
            s_hist[c * BW_CHUNK + tid] += 1;
            s_sorted[p] = (unsigned short)(tid * 16 + k);
            s_src[p] = srcs[k];
        }
    __syncthreads();
    float* out = partial + ((size_t)blockIdx.x * all_bins + cb) * T;
    const int j = lane >> 4, m = lane & 15;
    for (int c = wave; c < bins; c += 4) {
        const unsigned e0 = s_start[c], e1 = s_start[c + 1];
        f32x4 D0 = {0.f, 0.f, 0.f, 0.f}, D1 = {0.f, 0.f, 0.f, 0.f};
        for (unsigned e = e0; e < e1; e += 16) {                             // four MFMA steps (16 edges) per trip: their loads overlap
            float a0[4], a1[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned idx = e + 4 * u + j;
                const bool ok = idx < e1;
                const int slot = ok ? (int)s_sorted[idx] : 0;
                const int src = ok ? s_src[idx] : 0;
                const int dst = n0 + (slot >> 4);
                a0[u] = ok ? x[(size_t)src * CINP + m] : 0.f;
                a1[u] = (CINP == 20 && ok && m < 4) ? x[(size_t)src * CINP + 16 + m] : 0.f;
                b[u] = ok ? g[(size_t)dst * 16 + m] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                D0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b[u], D0, 0, 0, 0);
                if (CINP == 20) D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b[u], D1, 0, 0, 0);
            }
        }
        // D[4 * (lane >> 4) + r][lane & 15]: row = input channel, column = output
        float* oc = out + (size_t)c * T;
#pragma unroll
        for (int r = 0; r < 4; ++r) oc[(4 * j + r) * 16 + m] = D0[r];
        if (CINP == 20 && j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) oc[(16 + r) * 16 + m] = D1[r];
        }
    }
    __syncthreads();                                         // the next band re-uses the tables
    }
}

// 16 elements x 16 groups per workgroup: group q adds the tables q, q + 16, ... (four independent running sums, combined in a
// fixed order), the 16 group sums are then added in group order -- the same bits in every run.
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ partial, int n_part, int len, float* __restrict__ out) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4, t = blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (t < len) {
        int p = q;
        for (; p + 48 < n_part; p += 64) {
            s0 += partial[(size_t)p * len + t]; s1 += partial[(size_t)(p + 16) * len + t];
            s2 += partial[(size_t)(p + 32) * len + t]; s3 += partial[(size_t)(p + 48) * len + t];
        }
        for (; p < n_part; p += 16) s0 += partial[(size_t)p * len + t];
    }
    red[q][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && t < len) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][el];
        out[t] = s;
    }
}

// x f32[N, Cin_pad] (the conv's input rows), g f32[N,16]; dlut f32[steps+1][Cin_pad][16] in NATURAL channel order;
// partial = scratch f32[n_part][steps+1][Cin_pad][16] with n_part = eventad_l1_conv_bwd_weight_parts(N).
extern "C" int eventad_l1_conv_bwd_weight_parts(int N) {
    const int parts = (N + BW_CHUNK - 1) / BW_CHUNK;
    return parts < 1 ? 1 : parts;
}

extern "C" int eventad_l1_conv_bwd_weight(const float* x, const float* g, const int32_t* ell_src, const uint16_t* ell_step,
                                          const int32_t* deg, float* partial, float* dlut, int N, int steps, int cin_pad,
                                          void* stream) {
    if (!x || !g || !ell_src || !ell_step || !deg || !partial || !dlut || N < 0 || steps < 1 || (cin_pad != 20 && cin_pad != 16))
        return EVAD_ERR_ARG;
    const int T = cin_pad * 16, bins = steps + 1;
    // spiral steps sorted at a time: all of them when the tables fit 150 KiB of LDS (radius <= 7), else bands of 226
    const size_t fixed = sizeof(unsigned short) * BW_CHUNK * 16 + sizeof(int) * BW_CHUNK * 16;
    int band = bins;
    while (fixed + sizeof(unsigned short) * (size_t)band * BW_CHUNK + sizeof(unsigned) * (band + 1) > 150 * 1024) band = (band + 1) / 2;
    const size_t lds = fixed + sizeof(unsigned short) * (size_t)band * BW_CHUNK + sizeof(unsigned) * (band + 1);
    hipStream_t st = (hipStream_t)stream;
    const int parts = eventad_l1_conv_bwd_weight_parts(N);
    if (cin_pad == 20) {
        auto kern = k_conv_bwd_weight<20>;
        if (evad_allow_lds((const void*)kern, lds)) return -2;
        hipLaunchKernelGGL(kern, dim3(parts), dim3(256), lds, st, x, g, ell_src, ell_step, deg, partial, N, steps, band);
    } else {
        auto kern = k_conv_bwd_weight<16>;
        if (evad_allow_lds((const void*)kern, lds)) return -2;
        hipLaunchKernelGGL(kern, dim3(parts), dim3(256), lds, st, x, g, ell_src, ell_step, deg, partial, N, steps, band);
    }
    const int len = bins * T;
    hipLaunchKernelGGL(k_sum_partials, dim3(evad_blocks(len, 16)), dim3(256), 0, st, partial, parts, len, dlut);
    return evad_last_error();
}

// ================================================================================ pool1 (voxel max) and the feature sampler
// Forward (csrc/pool.hip k_pool_seg0; reference Pooling.forward with aggr = max, pooling.py:51-97, after sampling_skip,
// net.py:200-228):  feat[slot][c] = max over the events of the voxel of  [ y2 (16) | bilinear sample of the 64-channel map ].
// Backward of the max: the gradient of (slot, c) goes to the ONE member that attains the maximum -- what torch_scatter's
// arg-based scatter_max backward does; among exactly equal members (the reference's choice is then whichever thread
// wrote last) it goes to the smallest sorted position.  One wave per voxel slot walks the slot's members twice over the
// pixel CSR, like the forward pass: first the running (max, argmax), then one row of dy2 / dsamp per member.  Every
// event is a member of exactly one slot, so every row is written exactly once; nothing is accumulated across waves.
__global__ __launch_bounds__(256) void k_pool_seg0_bwd(
    const float* __restrict__ y2, const float* __restrict__ map, int map_h, int map_w, const float* __restrict__ posn_s,
    const int* __restrict__ key_s, const int* __restrict__ node_slot, const unsigned* __restrict__ pix_start,
    const int* __restrict__ vx_lo, const int* __restrict__ vy_lo, const int* __restrict__ ovf_list,
    const unsigned* __restrict__ counters, int B, int W, int H, int nx, int ny, int S, const float* __restrict__ dfeat,
    float* __restrict__ dy2, float* __restrict__ dsamp) {
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= S) return;
    const int lane = threadIdx.x & 63, grp = lane >> 4, q = lane & 15;
    const int plane = slot / (nx * ny), rem = slot % (nx * ny), cy = rem / nx, cx = rem % nx;
    const int n_ovf = (int)counters[2];
    // value of member r in this lane's channels: sampled channels 4q..4q+3 and channel q of y2 (same arithmetic as forward)
    auto value = [&](int r, float4& v, float& yv) -> bool {
        if (node_slot[r] != slot) return false;
        const float px = posn_s[3 * r], py = posn_s[3 * r + 1];
        const int b = key_s[r] / (W * H);
        yv = y2[(size_t)r * 16 + q];
        int x0, y0; float fx, fy;
        evad_bilinear_setup(px, py, (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
        const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
        const bool vx0 = x0 >= 0 && x0 < map_w, vx1 = x0 + 1 >= 0 && x0 + 1 < map_w;
        const bool vy0 = y0 >= 0 && y0 < map_h, vy1 = y0 + 1 >= 0 && y0 + 1 < map_h;
        const float* p00 = map + (((size_t)b * map_h + y0) * map_w + x0) * 64 + 4 * q;
        v = make_float4(0.f, 0.f, 0.f, 0.f);
        auto add = [&](bool ok, const float* p, float w) {
            if (!ok) return;
            const float4 c = *(const float4*)p;
            v.x += c.x * w; v.y += c.y * w; v.z += c.z * w; v.w += c.w * w;
        };
        add(vx0 && vy0, p00, w00); add(vx1 && vy0, p00 + 64, w01);
        add(vx0 && vy1, p00 + (size_t)map_w * 64, w10); add(vx1 && vy1, p00 + (size_t)map_w * 64 + 64, w11);
        return true;
    };
    auto for_members = [&](auto&& fn) {                      // this group's share of the slot's members
        if (plane < B) {
            const int xa = vx_lo[cx], xb = vx_lo[cx + 1], ya = vy_lo[cy], yb = vy_lo[cy + 1];
            for (int yy = ya; yy < yb; ++yy) {
                const int base = W * (yy + H * plane);
                const int ra = (int)pix_start[base + xa], rb = (int)pix_start[base + xb];
                for (int r = ra + grp; r < rb; r += 4) fn(r);
            }
        }
        for (int i = grp; i < n_ovf; i += 4) fn(ovf_list[i]);
    };
    const float NEG = -__builtin_inff();
    float mv[5] = {NEG, NEG, NEG, NEG, NEG};                 // running max of the 4 sampled channels and of y2's channel
    int am[5] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    auto take = [&](int c, float v, int r) { if (v > mv[c] || (v == mv[c] && r < am[c])) { mv[c] = v; am[c] = r; } };
    for_members([&](int r) {
        float4 v; float yv;
        if (!value(r, v, yv)) return;
        take(0, v.x, r); take(1, v.y, r); take(2, v.z, r); take(3, v.w, r); take(4, yv, r);
    });
#pragma unroll
    for (int d = 16; d <= 32; d <<= 1)
#pragma unroll
        for (int c = 0; c < 5; ++c) { const float ov = __shfl_xor(mv[c], d, 64); const int oa = __shfl_xor(am[c], d, 64); take(c, ov, oa); }
    const float* gs = dfeat + (size_t)slot * 80;
    const float gy = gs[q];
    const float4 g4 = *(const float4*)(gs + 16 + 4 * q);
    for_members([&](int r) {
        if (node_slot[r] != slot) return;
        dy2[(size_t)r * 16 + q] = am[4] == r ? gy : 0.f;
        *(float4*)(dsamp + (size_t)r * 64 + 4 * q) = make_float4(am[0] == r ? g4.x : 0.f, am[1] == r ? g4.y : 0.f,
                                                                 am[2] == r ? g4.z : 0.f, am[3] == r ? g4.w : 0.f);
    });
}

/* dfeat f32[S,80] = dL/d(slot feature table of pool1), S = nx*ny*(B+1); the forward inputs as eventad_pool_seg0 takes
 * them.  Out: dy2 f32[N,16], dsamp f32[N,64] (pixel-sorted numbering); every row of both is written. */
extern "C" int eventad_pool_seg0_bwd(const float* y2, const float* map_nhwc, int map_h, int map_w, const float* posn_s,
                                     const int32_t* key_s, const int32_t* node_slot, const uint32_t* pix_start,
                                     const int32_t* vx_lo, const int32_t* vy_lo, const int32_t* ovf_list, const uint32_t* counters,
                                     int N, int B, int W, int H, int nx, int ny, const float* dfeat, float* dy2, float* dsamp,
                                     void* stream) {
    if (!y2 || !map_nhwc || !posn_s || !key_s || !node_slot || !pix_start || !vx_lo || !vy_lo || !ovf_list || !counters ||
        !dfeat || !dy2 || !dsamp || N < 0 || B <= 0 || nx <= 0 || ny <= 0 || map_h <= 0 || map_w <= 0)
        return EVAD_ERR_ARG;
    if (N == 0) return EVAD_OK;
    const int S = nx * ny * (B + 1);
    hipLaunchKernelGGL(k_pool_seg0_bwd, dim3(evad_blocks(S, 4)), dim3(256), 0, (hipStream_t)stream, y2, map_nhwc, map_h, map_w,
                       posn_s, key_s, node_slot, pix_start, vx_lo, vy_lo, ovf_list, counters, B, W, H, nx, ny, S, dfeat, dy2, dsamp);
    return evad_last_error();
}

// Transpose of the bilinear sampler (reference: autograd through F.grid_sample in net.py:211-228):
//     dmap[b][ym][xm][c] = sum over events of clip b with a bilinear corner at (xm, ym) of  weight * g[event][c].
// Gather form, no float atomics: one wave per map pixel.  The events that can touch the pixel lie in a small rectangle of
// sensor pixels, i.e. in one contiguous range of the pixel-sorted array per sensor row; the rectangle is taken one pixel
// too large on every side and each event's corner weights are recomputed exactly as the forward pass does, so the
// membership test is the forward arithmetic itself.  4 groups of 16 lanes take every 4th event of a row range and are
// merged by a fixed tree: the sum order is the same in every run.
template <int CPL>      // channels per lane: the map has 16 * CPL channels
__global__ __launch_bounds__(256) void k_sample_bwd(const float* __restrict__ g, int ldg, const float* __restrict__ posn_s,
                                                    const unsigned* __restrict__ pix_start, int B, int W, int H, int map_h,
                                                    int map_w, float* __restrict__ dmap) {
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= (long long)B * map_h * map_w) return;
    const int lane = threadIdx.x & 63, grp = lane >> 4, q = lane & 15;
    const int xm = (int)(wv % map_w), ym = (int)((wv / map_w) % map_h), b = (int)(wv / ((long long)map_w * map_h));
    const float sx = map_w > 1 ? (float)(W - 1) / (float)(map_w - 1) : (float)W;
    const float sy = map_h > 1 ? (float)(H - 1) / (float)(map_h - 1) : (float)H;
    const int xlo = max(0, (int)floorf((float)(xm - 1) * sx) - 1), xhi = min(W - 1, (int)ceilf((float)(xm + 1) * sx) + 1);
    const int ylo = max(0, (int)floorf((float)(ym - 1) * sy) - 1), yhi = min(H - 1, (int)ceilf((float)(ym + 1) * sy) + 1);
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    for (int yy = ylo; yy <= yhi; ++yy) {
        const int base = W * (yy + H * b);
        const int ra = (int)pix_start[base + xlo], rb = (int)pix_start[base + xhi + 1];
        for (int r = ra + grp; r < rb; r += 4) {
            int x0, y0; float fx, fy;
            evad_bilinear_setup(posn_s[3 * r], posn_s[3 * r + 1], (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
            const bool left = x0 == xm, right = x0 + 1 == xm, top = y0 == ym, bottom = y0 + 1 == ym;
            if (!(left || right) || !(top || bottom)) continue;
            const float w = (left ? 1.f - fx : fx) * (top ? 1.f - fy : fy);       // the forward pass's w00 / w01 / w10 / w11
            const float* gr = g + (size_t)r * ldg + q * CPL;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] += gr[c] * w;
        }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) { acc[c] += __shfl_xor(acc[c], 16, 64); acc[c] += __shfl_xor(acc[c], 32, 64); }
    if (grp == 0) {
        float* o = dmap + (size_t)wv * (16 * CPL) + q * CPL;
#pragma unroll
        for (int c = 0; c < CPL; ++c) o[c] = acc[c];
    }
}

/* g: f32 rows of `ldg` floats in pixel-sorted numbering, the gradient of the sampled channels starting at g[r*ldg]
 * (pass dx0s + 1 with ldg = 20 for the 16-channel map of Layer 1's input, dsamp with ldg = 64 for pool1's 64-channel map).
 * dmap f32[B, map_h, map_w, C] (NHWC like the forward input), C = 16 or 64; every element is written. */
extern "C" int eventad_sample_bwd(const float* g, int ldg, int C, const float* posn_s, const uint32_t* pix_start, int N, int B,
                                  int W, int H, int map_h, int map_w, float* dmap, void* stream) {
    if (!g || !posn_s || !pix_start || !dmap || N < 0 || B <= 0 || W <= 0 || H <= 0 || map_h <= 0 || map_w <= 0 || ldg < C ||
        (C != 16 && C != 64))
        return EVAD_ERR_ARG;
    const long long waves = (long long)B * map_h * map_w;
    const dim3 grid(evad_blocks(waves, 4));
    if (C == 16) hipLaunchKernelGGL(k_sample_bwd<1>, grid, dim3(256), 0, (hipStream_t)stream, g, ldg, posn_s, pix_start, B, W, H, map_h, map_w, dmap);
    else         hipLaunchKernelGGL(k_sample_bwd<4>, grid, dim3(256), 0, (hipStream_t)stream, g, ldg, posn_s, pix_start, B, W, H, map_h, map_w, dmap);
    return evad_last_error();
}
