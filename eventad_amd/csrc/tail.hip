// eventad_amd/csrc/tail.hip -- spline convolutions on the pooled graphs (layers 2-5, <=2240 nodes/clip).
//
// Reference: MySplineConv with the cached LUT (src/dagr/model/layers/spline_conv.py:16-78).  Its LUTs
// for these layers are 12 MB .. 14 GB (net.py:72-89 over spline_conv.py:16-37) because every integer pixel offset gets its
// own Cin x Cout matrix.  A LUT entry is only a 4-tap blend of the 25 kernel matrices
// (degree-1 open B-spline, torch_spline_conv basis), so natively:
//     Z = X . [W_0 | ... | W_24 | W_root (| W_skip)]          one fp32 MFMA GEMM per conv (gemm.hip)
//     out_i = sum_{j->i} sum_{s<4} b_s(dx,dy) * Z[j, idx_s(dx,dy)]  + Z[i, root]      (this file)
// followed by the folded BatchNorm, the skip branch and ReLU (conv.py:23-28,47-56).
// One wave per destination node, lane = output channel (Cout = 64 for every tail layer); edges come
// as CSR by destination with ascending source id = the reference's edge order inside a destination.
#include "common.h"

// (evad_basis1 and evad_tail_blend_row -- the row arithmetic, shared with the fused blend + pooling kernel of pool.hip --
// live in common.h)
__global__ __launch_bounds__(256) void k_tail_blend(const float* __restrict__ Z, int ldz, int root_col,
                                                    const float* __restrict__ Zs, int ldzs, int skip_col,
                                                    const unsigned* __restrict__ rowptr, const int* __restrict__ csr_src,
                                                    const int* __restrict__ csr_dxy, int n_max, const int* __restrict__ n_dev,
                                                    float div_x, float div_y, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float* __restrict__ sscale,
                                                    const float* __restrict__ sshift, float* __restrict__ out, int ldo,
                                                    int ld_pad_to) {
    const int n = min(*n_dev, n_max);
    const int node = blockIdx.x * 4 + (threadIdx.x >> 6), c = threadIdx.x & 63;
    if (node >= n) return;
    const float y = evad_tail_blend_row(Z, ldz, root_col, Zs, ldzs, skip_col, rowptr, csr_src, csr_dxy, node, c, div_x, div_y,
                                        scale, shift, sscale, sshift);
    out[(size_t)node * ldo + c] = y;
    for (int k = 64 + c; k < ld_pad_to; k += 64) out[(size_t)node * ldo + k] = 0.f;
}

extern "C" int eventad_tail_blend(const float* Z, int ldz, int root_col, const float* Zskip, int ldzs, int skip_col,
                                  const uint32_t* rowptr, const int32_t* csr_src, const int32_t* csr_dxy, int n_max,
                                  const int32_t* n_dev, float div_x, float div_y, const float* bn_scale,
                                  const float* bn_shift, const float* skip_scale, const float* skip_shift, float* out,
                                  int ldo, void* stream) {
    if (!Z || !rowptr || !csr_src || !csr_dxy || !n_dev || !bn_scale || !bn_shift || !out || n_max < 0 || ldo < 64 ||
        div_x <= 0 || div_y <= 0 || (Zskip && (!skip_scale || !skip_shift)))
        return EVAD_ERR_ARG;
    if (n_max == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_tail_blend, dim3(evad_blocks(n_max, 4)), dim3(256), 0, (hipStream_t)stream, Z, ldz, root_col,
                       Zskip, ldzs, skip_col, rowptr, csr_src, csr_dxy, n_max, n_dev, div_x, div_y, bn_scale, bn_shift,
                       skip_scale, skip_shift, out, ldo, 64);
    return evad_last_error();
}

// Which kernel cells the blends of a level read, per 64 source nodes: every edge ORs the (up to) four cells of its 4-tap
// blend -- the same evad_basis1 arithmetic as evad_tail_blend_row, zero-weight taps included, since the blend reads them
// too -- into live[src / 64].  live must be zero on entry.  The products of the other cells are not computed
// (eventad_tail_gemm): at 256x256 the edges of pool1's graph use 9 to 12 of the 25 cells.
template <bool IN_LDS>
__global__ __launch_bounds__(256) void k_tail_live_cells(const int* __restrict__ csr_src, const int* __restrict__ csr_dxy,
                                                         const int* __restrict__ e_dev, int e_max, float div_x, float div_y,
                                                         unsigned* __restrict__ live, int n_words) {
    // A few hundred edges share a word, and atomics on one address are served one after the other (some 0.2 us each at the
    // L2): every workgroup gathers its edges' bits in an LDS copy of the words and sends each non-zero word once.
    extern __shared__ unsigned s_live[];
    if (IN_LDS) {
        for (int i = threadIdx.x; i < n_words; i += 256) s_live[i] = 0u;
        __syncthreads();
    }
    const int ne = min(*e_dev, e_max);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) {
        const int s = csr_src[e], pk = csr_dxy[e];
        const int dx = (int)(short)(pk & 0xffff), dy = pk >> 16;
        const float ax = __fadd_rn(__fdiv_rn((float)dx, div_x), 0.5f), ay = __fadd_rn(__fdiv_rn((float)dy, div_y), 0.5f);
        int ix0, ix1, iy0, iy1; float bx0, bx1, by0, by1;
        evad_basis1(ax, &ix0, &ix1, &bx0, &bx1);
        evad_basis1(ay, &iy0, &iy1, &by0, &by1);
        const unsigned bits = 1u << (ix0 + 5 * iy0) | 1u << (ix1 + 5 * iy0) | 1u << (ix0 + 5 * iy1) | 1u << (ix1 + 5 * iy1);
        const int t = s >> 6;
        if (t >= n_words) continue;                          // (a source beyond the node count: not a row of Z)
        if (IN_LDS) { if ((s_live[t] & bits) != bits) atomicOr(&s_live[t], bits); }
        else if ((*(volatile unsigned*)(live + t) & bits) != bits) atomicOr(live + t, bits);
    }
    if (IN_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_words; i += 256) {
            const unsigned bits = s_live[i];
            if (bits && (*(volatile unsigned*)(live + i) & bits) != bits) atomicOr(live + i, bits);
        }
    }
}

extern "C" int eventad_tail_live_cells(const int32_t* csr_src, const int32_t* csr_dxy, const int32_t* e_dev, int e_max, float div_x,
                                       float div_y, uint32_t* live, int n_words, void* stream) {
    if (!csr_src || !csr_dxy || !e_dev || !live || e_max < 0 || n_words < 0 || div_x <= 0 || div_y <= 0) return EVAD_ERR_ARG;
    if (e_max == 0 || n_words == 0) return EVAD_OK;
    unsigned blocks = evad_blocks(e_max, 2048);              // >= 8 edges per thread when the arrays are full
    if (blocks > 256) blocks = 256;
    if (n_words <= 8192)
        hipLaunchKernelGGL(k_tail_live_cells<true>, dim3(blocks), dim3(256), (size_t)n_words * 4, (hipStream_t)stream, csr_src, csr_dxy,
                           e_dev, e_max, div_x, div_y, live, n_words);
    else
        hipLaunchKernelGGL(k_tail_live_cells<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, csr_src, csr_dxy, e_dev, e_max,
                           div_x, div_y, live, n_words);
    return evad_last_error();
}

// NCHW -> NHWC for externally supplied ("precomputed") CNN feature maps: out[b,p,c] = in[b,c,p].
// All (up to 8) map layout changes of a step in ONE launch, as 32x32 LDS-tiled transposes (coalesced on both sides;
// the per-element kernel above reads with a stride of H*W floats).  desc: HOST array, 6 int64 per map:
// [0] in [1] out [2] B [3] C [4] H [5] W.
struct TrMap { const float* in; float* out; int C, P, tiles_c, tiles_p, block0; };
struct TrGroup { TrMap m[8]; int n; };

__global__ __launch_bounds__(256) void k_nchw_to_nhwc_group(TrGroup g) {
    __shared__ float tile[32][33];
    int mi = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.m[i].block0) mi = i;
    const TrMap& M = g.m[mi];
    int local = blockIdx.x - M.block0;
    const int per_b = M.tiles_c * M.tiles_p;
    const int b = local / per_b, rem = local % per_b, tc = rem / M.tiles_p, tp = rem % M.tiles_p;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* in = M.in + (size_t)b * M.C * M.P;
    float* out = M.out + (size_t)b * M.C * M.P;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                           // read: rows = channels, lanes = pixels
        int c = tc * 32 + ty + 8 * r, p = tp * 32 + tx;
        if (c < M.C && p < M.P) tile[ty + 8 * r][tx] = in[(size_t)c * M.P + p];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {                           // write: rows = pixels, lanes = channels
        int p = tp * 32 + ty + 8 * r, c = tc * 32 + tx;
        if (c < M.C && p < M.P) out[(size_t)p * M.C + c] = tile[tx][ty + 8 * r];
    }
}

extern "C" int eventad_nchw_to_nhwc_group(int n, const int64_t* desc, void* stream) {
    if (n < 1 || n > 8 || !desc) return EVAD_ERR_ARG;
    TrGroup g; g.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t* d = desc + 6 * i;
        TrMap& M = g.m[i];
        M.in = (const float*)d[0]; M.out = (float*)d[1];
        long long B = d[2], C = d[3], P = d[4] * d[5];
        if (!M.in || !M.out || B < 0 || C <= 0 || P <= 0 || B * C * P > (1ll << 31)) return EVAD_ERR_ARG;
        M.C = (int)C; M.P = (int)P; M.tiles_c = (int)((C + 31) / 32); M.tiles_p = (int)((P + 31) / 32); M.block0 = blocks;
        long long nb = B * M.tiles_c * M.tiles_p;
        if (blocks + nb > (1ll << 30)) return EVAD_ERR_ARG;
        blocks += (int)nb;
    }
    if (blocks == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_nchw_to_nhwc_group, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    return evad_last_error();
}
