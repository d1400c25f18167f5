// eventad_amd/csrc/gemm.hip -- exact-fp32 MFMA GEMM for the dense parts of the path
// (tail spline convolutions as X.[W_0|..|W_24|W_root|W_skip], 1x1 reducers, head projections).
//
//   C[M,N] = epilogue( A[M,K] . B[K,N] )        all row-major fp32
//
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: bitwise an fmaf chain in k order, so parity with
// the fp32 CPU oracle is at rounding level).  128x128 block tile, 4 waves (2x2), each wave a
// 64x64 sub-tile = 2x2 MFMA tiles; K is staged through LDS in slabs of 16, stored k-major so that
// the operand read of lane l -- A[m0 + (l&31)][k0 + (l>>5)] -- is a conflict-free ds_read_b32.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GBM 128
#define GBN 128
#define GBK 16

// epilogue: v = acc (+ bias[n]); v = v*scale[n] + shift[n] (if scale); v += residual[m,n] (if residual); relu
struct GemmEpi {
    const float* bias; const float* scale; const float* shift; const float* residual; int ldr; int relu;
};

__global__ __launch_bounds__(256) void k_gemm_f32(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                  float* __restrict__ C, int ldc, int M, const int* __restrict__ m_dev,
                                                  int N, int K, GemmEpi epi) {
    __shared__ float As[GBK][GBM + 4];
    __shared__ float Bs[GBK][GBN + 4];
    if (m_dev) M = min(M, *m_dev);
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += GBK) {
        // A slab: 128 rows x 16 k  -> 512 float4 (4 k-values of one row), 2 per thread, stored transposed
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int f = tid + t * 256;
            int row = f >> 2, kq = (f & 3) * 4;
            int gm = m0 + row, gk = k0 + kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M) {
                const float* p = A + (size_t)gm * lda + gk;
                if (gk + 3 < K) v = *(const float4*)p;
                else { if (gk < K) v.x = p[0]; if (gk + 1 < K) v.y = p[1]; if (gk + 2 < K) v.z = p[2]; }
            }
            As[kq][row] = v.x; As[kq + 1][row] = v.y; As[kq + 2][row] = v.z; As[kq + 3][row] = v.w;
        }
        // B slab: 16 k x 128 cols -> 512 float4 along n
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int f = tid + t * 256;
            int kr = f >> 5, nq = (f & 31) * 4;
            int gk = k0 + kr, gn = n0 + nq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < K) {
                const float* p = Bm + (size_t)gk * ldb + gn;
                if (gn + 3 < N) v = *(const float4*)p;
                else { if (gn < N) v.x = p[0]; if (gn + 1 < N) v.y = p[1]; if (gn + 2 < N) v.z = p[2]; }
            }
            *(float4*)&Bs[kr][nq] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            int kr = kk + (lane >> 5), c = lane & 31;
            float a0 = As[kr][wm + c], a1 = As[kr][wm + 32 + c];
            float b0 = Bs[kr][wn + c], b1 = Bs[kr][wn + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int gn = n0 + wn + j * 32 + (lane & 31);
            if (gn >= N) continue;
            float bia = epi.bias ? epi.bias[gn] : 0.f;
            float sc = epi.scale ? epi.scale[gn] : 1.f, sh = epi.scale ? epi.shift[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int gm = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm >= M) continue;
                float v = acc[i][j][r] + bia;
                if (epi.scale) v = fmaf(v, sc, sh);
                if (epi.residual) v += epi.residual[(size_t)gm * epi.ldr + gn];
                if (epi.relu) v = fmaxf(v, 0.f);
                C[(size_t)gm * ldc + gn] = v;
            }
        }
}

int evad_gemm(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, const int* m_dev, int N, int K,
              GemmEpi epi, hipStream_t st) {
    if (M <= 0 || N <= 0) return EVAD_OK;
    dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM);
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, st, A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi);
    return evad_last_error();
}

extern "C" int eventad_gemm_f32(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M,
                                const int32_t* m_dev, int N, int K, const float* bias, const float* scale,
                                const float* shift, const float* residual, int ldr, int relu, void* stream) {
    if (!A || !Bm || !C || M < 0 || N < 0 || K < 0 || lda < K || ldb < N || ldc < N || (lda & 3) || (ldb & 3) ||
        ((uintptr_t)A & 15) || ((uintptr_t)Bm & 15) || (scale && !shift))
        return EVAD_ERR_ARG;
    GemmEpi epi{bias, scale, shift, residual, ldr, relu};
    return evad_gemm(A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi, (hipStream_t)stream);
}
