// eventad_amd/csrc/bwd_pooled.hip -- backward of the pooled levels of the DAGR backbone (fine-tuning; EventAD itself
// freezes the backbone, models/EventAD.py:148-150, so nothing here runs in its training step).
//
// Forward of a pooled Layer (reference conv.py:59-72 over MySplineConv, spline_conv.py:39-78; csrc/tail.hip + gemm.hip):
//     Z = X . [W_0 | ... | W_24 | W_root (| W_skip)]                       fp32 MFMA GEMM
//     pre_i = sum_{j->i} sum_{s<4} b_s(dx,dy) Z[j, cell_s(dx,dy)] + Z[i, root]
//     y_i = ReLU(scale * pre_i + shift (+ sscale * Z[i, skip] + sshift))    folded eval-mode BatchNorm
// Backward, given g = dL/dy:  d = g * [y > 0] * scale (eventad_relu_scale_bwd);
//     dZ[j, c] = sum_{i : j->i} sum_{s : cell_s = c} b_s d_i ,  dZ[i, root] = d_i ,  dZ[i, skip] = g * [y > 0] * sscale
// (eventad_tail_blend_bwd: a gather over the TRANSPOSED graph, eventad_csr_transpose), and dX = dZ . W^T, dW = X^T . dZ are
// two more GEMMs (eventad_gemm_f32_ex).  Pooling (pooling.py:51-97): the gradient of a voxel's max goes to the member that
// attains it (eventad_pool_max_bwd), of a mean to every member / count; the sampled CNN channels' gradient goes back to
// the feature map through the transposed bilinear sampler (eventad_sample_nodes_bwd).
// The reference obtains all of this from autograd.  No float atomics: out-edge lists are built with integer atomics for
// the counts only and then ordered by edge id; every sum has a fixed order.
#include "common.h"

int evad_exclusive_scan(const unsigned* in, unsigned* out, unsigned* tile_sums, long long n, hipStream_t st);   // graph.hip

// ------------------------------------------------------------------------------------------------ transposed adjacency
// (the bounds on e, on the source id and on the list position below only matter for a caller whose rowptr, csr_src and E
// do not belong together: nothing is then written outside the arrays)
__global__ void k_csr_count(const unsigned* __restrict__ rowptr, const int* __restrict__ csr_src, int n, int E, unsigned* __restrict__ cnt,
                            int* __restrict__ edge_dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned e1 = min(rowptr[i + 1], (unsigned)E);
    for (unsigned e = rowptr[i]; e < e1; ++e) {
        edge_dst[e] = i;
        const unsigned s = (unsigned)csr_src[e];
        if (s < (unsigned)n) atomicAdd(&cnt[s], 1u);
    }
}
__global__ void k_csr_scatter(const int* __restrict__ csr_src, int n, int E, const unsigned* __restrict__ start, unsigned* __restrict__ fill,
                              int* __restrict__ tmp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const unsigned s = (unsigned)csr_src[e];
    if (s >= (unsigned)n) return;
    const unsigned p = start[s] + atomicAdd(&fill[s], 1u);
    if (p < (unsigned)E) tmp[p] = e;
}
// arrival order -> ascending edge id inside each source's list (lists are <= 75 long on the pooled graphs)
__global__ void k_csr_rank(const unsigned* __restrict__ start, const int* __restrict__ csr_src, const int* __restrict__ tmp, int n, int E,
                           int* __restrict__ out_edge) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E) return;
    const int e = tmp[t];
    if (e < 0 || e >= E) return;
    const unsigned s = (unsigned)csr_src[e];
    if (s >= (unsigned)n) return;
    const unsigned a = start[s], b = min(start[s + 1], (unsigned)E);
    unsigned rank = 0;
    for (unsigned q = a; q < b; ++q) rank += tmp[q] < e;
    if (a + rank < (unsigned)E) out_edge[a + rank] = e;
}

extern "C" int eventad_csr_transpose(const uint32_t* rowptr, const int32_t* csr_src, int n, int E, uint32_t* out_start,
                                     int32_t* out_edge, int32_t* edge_dst, uint32_t* cnt, uint32_t* scan_tmp, int32_t* tmp,
                                     void* stream) {
    if (!rowptr || !csr_src || !out_start || !out_edge || !edge_dst || !cnt || !scan_tmp || !tmp || n < 0 || E < 0) return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    evad_zero_async(cnt, sizeof(uint32_t) * (size_t)(n + 1), st);
    if (n > 0) hipLaunchKernelGGL(k_csr_count, dim3(evad_blocks(n, 256)), dim3(256), 0, st, rowptr, csr_src, n, E, cnt, edge_dst);
    int rc = evad_exclusive_scan(cnt, out_start, scan_tmp, n, st);              // out_start[n] = E
    if (rc) return rc;
    evad_zero_async(cnt, sizeof(uint32_t) * (size_t)(n + 1), st);
    if (E > 0) {
        hipLaunchKernelGGL(k_csr_scatter, dim3(evad_blocks(E, 256)), dim3(256), 0, st, csr_src, n, E, out_start, cnt, tmp);
        hipLaunchKernelGGL(k_csr_rank, dim3(evad_blocks(E, 256)), dim3(256), 0, st, out_start, csr_src, tmp, n, E, out_edge);
    }
    return evad_last_error();
}

// ------------------------------------------------------------------------------------------------ element-wise pieces
// (basis: evad_basis1, common.h -- the forward's)

// out_a = g * [y > 0] * scale_a (, out_b = g * [y > 0] * scale_b): through the ReLU and the folded BatchNorm scale(s)
__global__ void k_relu_scale_bwd(const float* __restrict__ g, int ldg, const float* __restrict__ y, int ldy, const float* __restrict__ sa,
                                 const float* __restrict__ sb, int n, int C, float* __restrict__ oa, float* __restrict__ ob) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * C) return;
    const int i = (int)(t / C), c = (int)(t % C);
    const float v = y[(size_t)i * ldy + c] > 0.f ? g[(size_t)i * ldg + c] : 0.f;
    oa[t] = v * sa[c];
    if (ob) ob[t] = v * sb[c];
}

extern "C" int eventad_relu_scale_bwd(const float* g, int ldg, const float* y, int ldy, const float* scale_a, const float* scale_b,
                                      int n, int C, float* out_a, float* out_b, void* stream) {
    if (!g || !y || !scale_a || !out_a || n < 0 || C <= 0 || ldg < C || ldy < C || (out_b && !scale_b)) return EVAD_ERR_ARG;
    if (n == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_relu_scale_bwd, dim3(evad_blocks((long long)n * C, 256)), dim3(256), 0, (hipStream_t)stream, g, ldg, y, ldy,
                       scale_a, scale_b, n, C, out_a, out_b);
    return evad_last_error();
}

// One wave per SOURCE node j, lane = channel: the 25 cell accumulators of the node live in the wave's LDS column; the
// node's out-edges are walked in ascending edge id (= a fixed summation order), 4 taps each.
__global__ __launch_bounds__(256) void k_tail_blend_bwd(const float* __restrict__ d, int ldd, const unsigned* __restrict__ out_start,
                                                        const int* __restrict__ out_edge, const int* __restrict__ edge_dst,
                                                        const int* __restrict__ csr_dxy, int n, float div_x, float div_y,
                                                        float* __restrict__ dZ, int ldz, int root_col, const float* __restrict__ dskip,
                                                        int skip_col) {
    __shared__ float s_acc[4][25][64];
    const int w = threadIdx.x >> 6, c = threadIdx.x & 63, j = blockIdx.x * 4 + w;
    if (j >= n) return;                                      // wave-uniform; no workgroup barrier below
#pragma unroll
    for (int k = 0; k < 25; ++k) s_acc[w][k][c] = 0.f;
    for (unsigned q = out_start[j]; q < out_start[j + 1]; ++q) {
        const int e = out_edge[q], i = edge_dst[e], pk = csr_dxy[e];
        const int dx = (int)(short)(pk & 0xffff), dy = pk >> 16;
        const float ax = __fadd_rn(__fdiv_rn((float)dx, div_x), 0.5f), ay = __fadd_rn(__fdiv_rn((float)dy, div_y), 0.5f);
        int ix0, ix1, iy0, iy1; float bx0, bx1, by0, by1;
        evad_basis1(ax, &ix0, &ix1, &bx0, &bx1);
        evad_basis1(ay, &iy0, &iy1, &by0, &by1);
        const float v = d[(size_t)i * ldd + c];
        s_acc[w][ix0 + 5 * iy0][c] += __fmul_rn(bx0, by0) * v;
        s_acc[w][ix1 + 5 * iy0][c] += __fmul_rn(bx1, by0) * v;
        s_acc[w][ix0 + 5 * iy1][c] += __fmul_rn(bx0, by1) * v;
        s_acc[w][ix1 + 5 * iy1][c] += __fmul_rn(bx1, by1) * v;
    }
    float* row = dZ + (size_t)j * ldz;
#pragma unroll
    for (int k = 0; k < 25; ++k) row[k * 64 + c] = s_acc[w][k][c];
    row[root_col * 64 + c] = d[(size_t)j * ldd + c];
    if (dskip) row[skip_col * 64 + c] = dskip[(size_t)j * 64 + c];
}

extern "C" int eventad_tail_blend_bwd(const float* d, int ldd, const uint32_t* out_start, const int32_t* out_edge,
                                      const int32_t* edge_dst, const int32_t* csr_dxy, int n, float div_x, float div_y, float* dZ,
                                      int ldz, int root_col, const float* dskip, int skip_col, void* stream) {
    if (!d || !out_start || !out_edge || !edge_dst || !csr_dxy || !dZ || n < 0 || ldd < 64 || div_x <= 0 || div_y <= 0 ||
        root_col < 25 || ldz < (root_col + 1) * 64 || (dskip && (skip_col < 25 || skip_col == root_col || ldz < (skip_col + 1) * 64)))
        return EVAD_ERR_ARG;
    if (n == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_tail_blend_bwd, dim3(evad_blocks(n, 4)), dim3(256), 0, (hipStream_t)stream, d, ldd, out_start, out_edge,
                       edge_dst, csr_dxy, n, div_x, div_y, dZ, ldz, root_col, dskip, skip_col);
    return evad_last_error();
}

// ------------------------------------------------------------------------------------------------ pooling
// value of channel c of input node m as the forward pooled it: c < Ca from xa, else the bilinear sample of the map
__device__ __forceinline__ float bp_member_value(const float* __restrict__ xa, int lda, int Ca, const float* __restrict__ map, int map_h,
                                                 int map_w, int Cm, const float* __restrict__ npos, const int* __restrict__ nbatch,
                                                 int W, int H, int m, int c) {
    if (c < Ca) return xa[(size_t)m * lda + c];
    int x0, y0; float fx, fy;
    evad_bilinear_setup(npos[3 * m], npos[3 * m + 1], (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    const bool vx0 = x0 >= 0 && x0 < map_w, vx1 = x0 + 1 >= 0 && x0 + 1 < map_w;
    const bool vy0 = y0 >= 0 && y0 < map_h, vy1 = y0 + 1 >= 0 && y0 + 1 < map_h;
    const float* p00 = map + ((size_t)nbatch[m] * map_h * map_w + (size_t)y0 * map_w + x0) * Cm + (c - Ca);
    float v = 0.f;                                           // same order of the four terms as k_pool_scatter
    if (vx0 && vy0) v += p00[0] * w00;
    if (vx1 && vy0) v += p00[Cm] * w01;
    if (vx0 && vy1) v += p00[(size_t)map_w * Cm] * w10;
    if (vx1 && vy1) v += p00[(size_t)map_w * Cm + Cm] * w11;
    return v;
}

// pass 1 (max): smallest member index among the members that attain the pooled maximum, per (output node, channel), kept
// as 0x7fffffff - index under atomicMax so that the table starts from zeros (cleared by a kernel: capturable)
__global__ void k_pool_argmax(const float* __restrict__ xa, int lda, int Ca, const float* __restrict__ map, int map_h, int map_w, int Cm,
                              const float* __restrict__ npos, const int* __restrict__ nbatch, const int* __restrict__ node_slot,
                              const int* __restrict__ slot2node, int n_in, int W, int H, const float* __restrict__ xout, int ldo,
                              int* __restrict__ argm) {
    const int C = Ca + Cm;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_in * C) return;
    const int m = (int)(t / C), c = (int)(t % C);
    const int node = slot2node[node_slot[m]];
    const float v = bp_member_value(xa, lda, Ca, map, map_h, map_w, Cm, npos, nbatch, W, H, m, c);
    if (v == xout[(size_t)node * ldo + c]) atomicMax(&argm[(size_t)node * C + c], 0x7fffffff - m);
}
// pass 2: the gradient of (output node, channel) to that member (max), or to every member divided by the count (mean)
__global__ void k_pool_route(const int* __restrict__ node_slot, const int* __restrict__ slot2node, const unsigned* __restrict__ cnt,
                             int n_in, int Ca, int Cm, int mean, const int* __restrict__ argm, const float* __restrict__ dxout,
                             int ldd, float* __restrict__ dxa, float* __restrict__ dsamp) {
    const int C = Ca + Cm;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_in * C) return;
    const int m = (int)(t / C), c = (int)(t % C);
    const int slot = node_slot[m], node = slot2node[slot];
    const float g = dxout[(size_t)node * ldd + c];
    const float v = mean ? g / (float)cnt[slot] : (argm[(size_t)node * C + c] == 0x7fffffff - m ? g : 0.f);
    if (c < Ca) dxa[(size_t)m * Ca + c] = v; else dsamp[(size_t)m * Cm + (c - Ca)] = v;
}

extern "C" int eventad_pool_max_bwd(const float* xa, int lda, int Ca, const float* map_nhwc, int map_h, int map_w, int Cm,
                                    const float* npos, const int32_t* nbatch, const int32_t* node_slot, const int32_t* slot2node,
                                    const uint32_t* cnt, int n_in, int n_out, int W, int H, int mean, const float* xout, int ldo,
                                    const float* dxout, int ldd, int32_t* argm, float* dxa, float* dsamp, void* stream) {
    if (!xa || !npos || !nbatch || !node_slot || !slot2node || !cnt || !xout || !dxout || !argm || !dxa || n_in < 0 || n_out < 0 ||
        Ca <= 0 || Cm < 0 || lda < Ca || (Cm > 0 && (!map_nhwc || !dsamp || map_h <= 0 || map_w <= 0)) || ldo < Ca + Cm || ldd < Ca + Cm)
        return EVAD_ERR_ARG;
    if (n_in == 0) return EVAD_OK;
    hipStream_t st = (hipStream_t)stream;
    const int C = Ca + Cm;
    const unsigned blocks = evad_blocks((long long)n_in * C, 256);
    if (!mean) {
        evad_zero_async(argm, sizeof(int32_t) * (size_t)n_out * C, st);
        hipLaunchKernelGGL(k_pool_argmax, dim3(blocks), dim3(256), 0, st, xa, lda, Ca, map_nhwc, map_h, map_w, Cm, npos, nbatch,
                           node_slot, slot2node, n_in, W, H, xout, ldo, argm);
    }
    hipLaunchKernelGGL(k_pool_route, dim3(blocks), dim3(256), 0, st, node_slot, slot2node, cnt, n_in, Ca, Cm, mean, argm, dxout, ldd,
                       dxa, dsamp);
    return evad_last_error();
}

// ------------------------------------------------------------------------------------------------ sampler at pooled nodes
// dmap[b, y, x, :] = sum over the nodes of clip b whose bilinear footprint contains pixel (x, y) of weight * dsamp[node, :].
// One wave per map pixel.  Nodes are numbered in ascending voxel-slot order, so the nodes of slot plane p are the range
// [occ[p * plane], occ[(p + 1) * plane]); a node of clip b lies in plane b or b + 1 (an event at t == time_window is
// clustered into the next plane, pooling.py:51-58 over grid_cluster).  The 64 lanes test 64 nodes of the range at a time;
// the hits (ballot) are then added in ascending node order, lane = channel: a fixed summation order.
__global__ __launch_bounds__(64) void k_sample_nodes_bwd(const float* __restrict__ dsamp, int C, const float* __restrict__ npos,
                                                         const int* __restrict__ nbatch, const unsigned* __restrict__ occ, int plane, int S,
                                                         int W, int H, int map_h, int map_w, float* __restrict__ dmap) {
    const int pix = blockIdx.x, lane = threadIdx.x;
    const int b = pix / (map_h * map_w), rem = pix % (map_h * map_w), py = rem / map_w, px = rem % map_w;
    const int m0 = (int)occ[min(b * plane, S)], m1 = (int)occ[min((b + 2) * plane, S)];
    float acc = 0.f;
    for (int base = m0; base < m1; base += 64) {
        const int m = base + lane;
        bool hit = false; float wgt = 0.f;
        if (m < m1 && nbatch[m] == b) {
            int x0, y0; float fx, fy;
            evad_bilinear_setup(npos[3 * m], npos[3 * m + 1], (float)W, (float)H, map_w, map_h, &x0, &y0, &fx, &fy);
            const int ddx = px - x0, ddy = py - y0;
            hit = ddx >= 0 && ddx <= 1 && ddy >= 0 && ddy <= 1;
            wgt = (ddx ? fx : 1.f - fx) * (ddy ? fy : 1.f - fy);
        }
        unsigned long long hits = __ballot(hit);
        while (hits) {
            const int l = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const float w = __shfl(wgt, l, 64);
            if (lane < C) acc += w * dsamp[(size_t)(base + l) * C + lane];
        }
    }
    if (lane < C) dmap[(size_t)pix * C + lane] = acc;
}

extern "C" int eventad_sample_nodes_bwd(const float* dsamp, int C, const float* npos, const int32_t* nbatch, const uint32_t* occ,
                                        int nx, int ny, int B, int W, int H, int map_h, int map_w, float* dmap, void* stream) {
    if (!dsamp || !npos || !nbatch || !occ || !dmap || B <= 0 || C <= 0 || C > 64 || nx <= 0 || ny <= 0 || map_h <= 0 || map_w <= 0)
        return EVAD_ERR_ARG;
    hipLaunchKernelGGL(k_sample_nodes_bwd, dim3((unsigned)(B * map_h * map_w)), dim3(64), 0, (hipStream_t)stream, dsamp, C, npos, nbatch,
                       occ, nx * ny, nx * ny * (B + 1), W, H, map_h, map_w, dmap);
    return evad_last_error();
}
