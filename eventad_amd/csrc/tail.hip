// eventad_amd/csrc/tail.hip -- spline convolutions on the pooled graphs (layers 2-5, <=2240 nodes/clip).
//
// Reference: MySplineConv with the cached LUT (src/dagr/model/layers/spline_conv.py:16-78).  Its LUTs
// for these layers are 12 MB .. 14 GB because every integer pixel offset gets its
// own Cin x Cout matrix.  A LUT entry is only a 4-tap blend of the 25 kernel matrices
// (degree-1 open B-spline, torch_spline_conv basis), so natively:
//     Z = X . [W_0 | ... | W_24 | W_root (| W_skip)]          one fp32 MFMA GEMM per conv (gemm.hip)
//     out_i = sum_{j->i} sum_{s<4} b_s(dx,dy) * Z[j, idx_s(dx,dy)]  + Z[i, root]      (this file)
// followed by the folded BatchNorm, the skip branch and ReLU (conv.py:23-28,47-56).
// One wave per destination node, lane = output channel (Cout = 64 for every tail layer); edges come
// as CSR by destination with ascending source id = the reference's edge order inside a destination.
#include "common.h"

// torch_spline_conv degree-1 basis for one pseudo-coordinate: v = attr*(kernel_size-1);
// index k_mod in {0,1}: ((int64)v + k_mod) % 5,  weight 1 - frac - k_mod + 2*frac*k_mod.
__device__ __forceinline__ void basis1(float attr, int* i0, int* i1, float* b0, float* b1) {
    float v = __fmul_rn(attr, 4.f);
    int t = (int)v;                                    // truncation, as the (int64_t) cast
    float fr = __fsub_rn(v, floorf(v));
    int a = t % 5, b = (t + 1) % 5;
    *i0 = a < 0 ? a + 5 : a; *i1 = b < 0 ? b + 5 : b;  // advanced indexing wraps negative indices
    *b0 = __fsub_rn(1.f, fr);
    *b1 = __fadd_rn(__fsub_rn(__fsub_rn(1.f, fr), 1.f), __fmul_rn(2.f, fr));
}

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
    float acc = 0.f;
    const unsigned e0 = rowptr[node], e1 = rowptr[node + 1];
    // Up to 64 edges at a time: lane l fetches edge l's source and offset (ONE coalesced load each instead of a
    // dependent pair per edge) and works out that edge's 4 taps; the taps are then broadcast edge by edge, so the
    // Z gathers of consecutive edges are independent loads in flight together.  Messages are still added in
    // ascending edge order (= the reference's order inside a destination).
    for (unsigned eb = e0; eb < e1; eb += 64) {
        const int ne = min(64u, e1 - eb);
        int s = 0, o00 = 0, o10 = 0, o01 = 0, o11 = 0;
        float w00 = 0.f, w10 = 0.f, w01 = 0.f, w11 = 0.f;
        if (c < ne) {
            s = csr_src[eb + c];
            const int pk = csr_dxy[eb + c];
            const int dx = (int)(short)(pk & 0xffff), dy = pk >> 16;
            // LUT grid attribute (spline_conv.py:27-30): d / (2*M*size) + 0.5
            const float ax = __fadd_rn(__fdiv_rn((float)dx, div_x), 0.5f), ay = __fadd_rn(__fdiv_rn((float)dy, div_y), 0.5f);
            int ix0, ix1, iy0, iy1; float bx0, bx1, by0, by1;
            basis1(ax, &ix0, &ix1, &bx0, &bx1);
            basis1(ay, &iy0, &iy1, &by0, &by1);
            o00 = (ix0 + 5 * iy0) * 64; o10 = (ix1 + 5 * iy0) * 64; o01 = (ix0 + 5 * iy1) * 64; o11 = (ix1 + 5 * iy1) * 64;
            w00 = __fmul_rn(bx0, by0); w10 = __fmul_rn(bx1, by0); w01 = __fmul_rn(bx0, by1); w11 = __fmul_rn(bx1, by1);
        }
        for (int j0 = 0; j0 < ne; j0 += 4) {                 // 4 edges = 16 independent gathers per trip
            float z[4][4], ww[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + u, ne - 1);
                const float* zr = Z + (size_t)__shfl(s, j, 64) * ldz + c;
                z[u][0] = zr[__shfl(o00, j, 64)]; z[u][1] = zr[__shfl(o10, j, 64)];
                z[u][2] = zr[__shfl(o01, j, 64)]; z[u][3] = zr[__shfl(o11, j, 64)];
                ww[u][0] = __shfl(w00, j, 64); ww[u][1] = __shfl(w10, j, 64); ww[u][2] = __shfl(w01, j, 64); ww[u][3] = __shfl(w11, j, 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u < ne) {
                    float m = __fmul_rn(ww[u][0], z[u][0]);
                    m = fmaf(ww[u][1], z[u][1], m);
                    m = fmaf(ww[u][2], z[u][2], m);
                    m = fmaf(ww[u][3], z[u][3], m);
                    acc += m;
                }
            }
        }
    }
    acc += Z[(size_t)node * ldz + root_col * 64 + c];
    float y = fmaf(acc, scale[c], shift[c]);
    if (Zs) y += fmaf(Zs[(size_t)node * ldzs + skip_col * 64 + c], sscale[c], sshift[c]);
    y = fmaxf(y, 0.f);
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
