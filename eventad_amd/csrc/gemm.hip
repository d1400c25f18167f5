// eventad_amd/csrc/gemm.hip -- exact-fp32 MFMA GEMMs for the dense parts of the path
// (tail spline convolutions as X.[W_0|..|W_24|W_root|W_skip], head GRU / fusion projections and
// their weight gradients).
//
//   C[M,N] (+)= epilogue( op(A)[M,K] . op(B)[K,N] )        fp32, row-major storage
//       op(A) = A        (A stored [M,K], lda)   or  A^T  (A stored [K,M], lda)      -- trans_a
//       op(B) = B        (B stored [K,N], ldb)   or  B^T  (B stored [N,K], ldb)      -- trans_b
//
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: bitwise an fmaf chain in k order, so parity with
// the fp32 CPU oracle is at rounding level).  K is staged through LDS in slabs, stored k-major so
// that the operand read of lane l -- A[m0 + (l&31)][k0 + (l>>5)] -- is a conflict-free ds_read_b32.
//   k_gemm_f32<TM,TN>   4 waves (2x2), each wave TMxTN MFMA tiles: 128x128 block (TM=TN=2, big M)
//                       or 64x64 block (TM=TN=1, the head's 256-row GEMMs)
//   k_gemm_m32          M <= 32 rows (one clip step of the head): 32x64 block, the 4 waves split K
//                       and reduce through LDS, so a [32,256]x[256,768] product is ~2 us of MFMA.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GBK 16

// epilogue: v = acc (+ bias[n]); v = v*scale[n] + shift[n] (if scale); v += residual[m,n] (if residual);
//           relu; C = accumulate ? C + v : v
struct GemmEpi {
    const float* bias; const float* scale; const float* shift; const float* residual; int ldr; int relu; int accumulate;
};

__device__ __forceinline__ float4 ld4_guard(const float* p, int i, int n) {
    // p points at element i of a row with n valid elements
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i + 3 < n) v = *(const float4*)p;
    else { if (i < n) v.x = p[0]; if (i + 1 < n) v.y = p[1]; if (i + 2 < n) v.z = p[2]; }
    return v;
}

__device__ __forceinline__ float epilogue(float v, int gm, int gn, const GemmEpi& e, const float* C, int ldc) {
    if (e.bias) v += e.bias[gn];
    if (e.scale) v = fmaf(v, e.scale[gn], e.shift[gn]);
    if (e.residual) v += e.residual[(size_t)gm * e.ldr + gn];
    if (e.relu) v = fmaxf(v, 0.f);
    if (e.accumulate) v += C[(size_t)gm * ldc + gn];
    return v;
}

// Software pipeline: the global loads of slab k+1 are issued into registers before the MFMAs of slab k
// and written to the OTHER LDS buffer afterwards, so global latency hides under the 8 x TM x TN MFMAs of a
// slab and there is one barrier per slab instead of two.
template <int TM, int TN, bool AT, bool BT>
__global__ __launch_bounds__(256, TM * TN == 4 ? 4 : 1) void k_gemm_f32(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                  float* __restrict__ C, int ldc, int M, const int* __restrict__ m_dev,
                                                  int N, int K, GemmEpi epi, const unsigned* __restrict__ live, int live_blocks) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    // live (64x64 tiles only): word i = which of the first live_blocks 64-column blocks anyone reads of rows 64i .. 64i+63;
    // a tile nobody reads is not computed (its part of C keeps whatever it held)
    if (TM * TN == 1 && live && (int)blockIdx.x < live_blocks && !((live[blockIdx.y] >> blockIdx.x) & 1u)) return;
    constexpr int AL = BM * (GBK / 4) / 256, BL = BN * (GBK / 4) / 256;     // float4 loads per thread per slab
    __shared__ float As[2][GBK][BM + 4];
    __shared__ float Bs[2][GBK][BN + 4];
    if (m_dev) M = min(M, *m_dev);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32 * TM, wn = (wave & 1) * 32 * TN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[AL], rb[BL];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int t = 0; t < AL; ++t) {
            int f = tid + t * 256;
            ra[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!AT) { int row = f >> 2, kq = (f & 3) * 4, gm = m0 + row, gk = k0 + kq;
                       if (gm < M) ra[t] = ld4_guard(A + (size_t)gm * lda + gk, gk, K); }
            else     { int kr = f / (BM / 4), mq = (f % (BM / 4)) * 4, gk = k0 + kr, gm = m0 + mq;
                       if (gk < K) ra[t] = ld4_guard(A + (size_t)gk * lda + gm, gm, M); }
        }
#pragma unroll
        for (int t = 0; t < BL; ++t) {
            int f = tid + t * 256;
            rb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!BT) { int kr = f / (BN / 4), nq = (f % (BN / 4)) * 4, gk = k0 + kr, gn = n0 + nq;
                       if (gk < K) rb[t] = ld4_guard(Bm + (size_t)gk * ldb + gn, gn, N); }
            else     { int col = f >> 2, kq = (f & 3) * 4, gn = n0 + col, gk = k0 + kq;
                       if (gn < N) rb[t] = ld4_guard(Bm + (size_t)gn * ldb + gk, gk, K); }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int t = 0; t < AL; ++t) {
            int f = tid + t * 256; float4 v = ra[t];
            if (!AT) { int row = f >> 2, kq = (f & 3) * 4;
                       As[buf][kq][row] = v.x; As[buf][kq + 1][row] = v.y; As[buf][kq + 2][row] = v.z; As[buf][kq + 3][row] = v.w; }
            else     { int kr = f / (BM / 4), mq = (f % (BM / 4)) * 4; *(float4*)&As[buf][kr][mq] = v; }
        }
#pragma unroll
        for (int t = 0; t < BL; ++t) {
            int f = tid + t * 256; float4 v = rb[t];
            if (!BT) { int kr = f / (BN / 4), nq = (f % (BN / 4)) * 4; *(float4*)&Bs[buf][kr][nq] = v; }
            else     { int col = f >> 2, kq = (f & 3) * 4;
                       Bs[buf][kq][col] = v.x; Bs[buf][kq + 1][col] = v.y; Bs[buf][kq + 2][col] = v.z; Bs[buf][kq + 3][col] = v.w; }
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += GBK) {
        const bool more = k0 + GBK < K;
        if (more) fetch(k0 + GBK);
        {   // the operands of k-step kk + 2 are requested from LDS before the MFMAs of step kk are issued
            const int c = lane & 31, kh2 = lane >> 5;
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = As[cur][kh2][wm + 32 * i + c];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = Bs[cur][kh2][wn + 32 * j + c];
#pragma unroll
            for (int kk = 0; kk < GBK; kk += 2) {
                const int q = (kk >> 1) & 1;
                if (kk + 2 < GBK) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[q ^ 1][i] = As[cur][kk + 2 + kh2][wm + 32 * i + c];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[q ^ 1][j] = Bs[cur][kk + 2 + kh2][wn + 32 * j + c];
                }
                __builtin_amdgcn_sched_barrier(0);           // keep the reads above the MFMAs (the scheduler sinks them to their use)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int gn = n0 + wn + j * 32 + (lane & 31);
            if (gn >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int gm = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm >= M) continue;
                C[(size_t)gm * ldc + gn] = epilogue(acc[i][j][r], gm, gn, epi, C, ldc);
            }
        }
}

// ---- M <= 32: one 32x64 output block per workgroup, K split over the 4 waves inside each slab
#define SBK 32
template <bool BT>
__global__ __launch_bounds__(256) void k_gemm_m32(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                  float* __restrict__ C, int ldc, int M, int N, int K, GemmEpi epi) {
    __shared__ float As[SBK][32 + 1];
    __shared__ float Bs[SBK][64 + 4];
    __shared__ float Red[3][32][64 + 1];
    const int n0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += SBK) {
        {   // A slab: 32 rows x 32 k = 256 float4
            int row = tid >> 3, kq = (tid & 7) * 4, gk = k0 + kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) v = ld4_guard(A + (size_t)row * lda + gk, gk, K);
            As[kq][row] = v.x; As[kq + 1][row] = v.y; As[kq + 2][row] = v.z; As[kq + 3][row] = v.w;
        }
        if (!BT) {
            for (int f = tid; f < SBK * 16; f += 256) {
                int kr = f >> 4, nq = (f & 15) * 4, gk = k0 + kr, gn = n0 + nq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gk < K) v = ld4_guard(Bm + (size_t)gk * ldb + gn, gn, N);
                *(float4*)&Bs[kr][nq] = v;
            }
        } else {
            for (int f = tid; f < 64 * (SBK / 4); f += 256) {
                int col = f >> 3, kq = (f & 7) * 4, gn = n0 + col, gk = k0 + kq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gn < N) v = ld4_guard(Bm + (size_t)gn * ldb + gk, gk, K);
                Bs[kq][col] = v.x; Bs[kq + 1][col] = v.y; Bs[kq + 2][col] = v.z; Bs[kq + 3][col] = v.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SBK / 4; kk += 2) {                   // this wave's quarter of the slab
            int kr = wave * (SBK / 4) + kk + (lane >> 5), c = lane & 31;
            float a = As[kr][c];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[kr][c], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[kr][32 + c], acc[1], 0, 0, 0);
        }
        __syncthreads();
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Red[wave - 1][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][j * 32 + (lane & 31)] = acc[j][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int col = j * 32 + (lane & 31), gn = n0 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int gm = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gn >= N || gm >= M) continue;
                float v = ((acc[j][r] + Red[0][gm][col]) + Red[1][gm][col]) + Red[2][gm][col];
                C[(size_t)gm * ldc + gn] = epilogue(v, gm, gn, epi, C, ldc);
            }
        }
    }
}

template <int TM, int TN>
static void launch_big(bool ta, bool tb, dim3 grid, hipStream_t st, const float* A, int lda, const float* Bm, int ldb,
                       float* C, int ldc, int M, const int* m_dev, int N, int K, GemmEpi epi, const unsigned* live = nullptr,
                       int live_blocks = 0) {
    if (!ta && !tb) hipLaunchKernelGGL((k_gemm_f32<TM, TN, false, false>), grid, dim3(256), 0, st, A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi, live, live_blocks);
    else if (!ta && tb) hipLaunchKernelGGL((k_gemm_f32<TM, TN, false, true>), grid, dim3(256), 0, st, A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi, live, live_blocks);
    else if (ta && !tb) hipLaunchKernelGGL((k_gemm_f32<TM, TN, true, false>), grid, dim3(256), 0, st, A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi, live, live_blocks);
    else hipLaunchKernelGGL((k_gemm_f32<TM, TN, true, true>), grid, dim3(256), 0, st, A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi, live, live_blocks);
}

int evad_gemm(bool ta, bool tb, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M,
              const int* m_dev, int N, int K, GemmEpi epi, hipStream_t st) {
    if (M <= 0 || N <= 0) return EVAD_OK;
    if (M <= 32 && !ta && !m_dev) {
        dim3 grid((N + 63) / 64);
        if (tb) hipLaunchKernelGGL(k_gemm_m32<true>, grid, dim3(256), 0, st, A, lda, Bm, ldb, C, ldc, M, N, K, epi);
        else    hipLaunchKernelGGL(k_gemm_m32<false>, grid, dim3(256), 0, st, A, lda, Bm, ldb, C, ldc, M, N, K, epi);
    } else if ((long long)M * N >= 128ll * 128 * 256 && K >= 256) {   // shallow K (the tail's 64..132): 64x64 tiles, measured 10-20 % faster
        launch_big<2, 2>(ta, tb, dim3((N + 127) / 128, (M + 127) / 128), st, A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi);
    } else {
        launch_big<1, 1>(ta, tb, dim3((N + 63) / 64, (M + 63) / 64), st, A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi);
    }
    return evad_last_error();
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Z = X . [W_0 | .. | W_24 | W_root (| W_skip)] of a pooled-level spline convolution, computed only where a blend reads it:
// live[i] (eventad_tail_live_cells, tail.hip) names the kernel cells that the out-edges of nodes 64i .. 64i+63 use.  The
// other 64x64 tiles of the first 25 column blocks are skipped -- their part of Z is never read.
extern "C" int eventad_tail_gemm(const float* X, int ldx, const float* Wc, int ldw, float* Z, int ldz, int n_max,
                                 const int32_t* n_dev, int N, int K, const uint32_t* live, void* stream) {
    if (!X || !Wc || !Z || !n_dev || !live || n_max < 0 || N < 25 * 64 || N > 32 * 64 || (N & 63) || K < 0 || ldx < K || ldw < N ||
        ldz < N || (ldx & 3) || (ldw & 3) || !al16(X) || !al16(Wc))
        return EVAD_ERR_ARG;
    if (n_max == 0) return EVAD_OK;
    GemmEpi epi{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    launch_big<1, 1>(false, false, dim3(N / 64, (n_max + 63) / 64), (hipStream_t)stream, X, ldx, Wc, ldw, Z, ldz, n_max, n_dev, N, K, epi,
                     live, 25);
    return evad_last_error();
}

extern "C" int eventad_gemm_f32(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M,
                                const int32_t* m_dev, int N, int K, const float* bias, const float* scale,
                                const float* shift, const float* residual, int ldr, int relu, void* stream) {
    if (!A || !Bm || !C || M < 0 || N < 0 || K < 0 || lda < K || ldb < N || ldc < N || (lda & 3) || (ldb & 3) ||
        !al16(A) || !al16(Bm) || (scale && !shift))
        return EVAD_ERR_ARG;
    GemmEpi epi{bias, scale, shift, residual, ldr, relu, 0};
    return evad_gemm(false, false, A, lda, Bm, ldb, C, ldc, M, m_dev, N, K, epi, (hipStream_t)stream);
}

/* General form: trans_a / trans_b select the stored layout of A / B (see file header);
 * accumulate != 0 adds into C.  Leading dimensions must be multiples of 4 and bases 16-byte aligned. */
extern "C" int eventad_gemm_f32_ex(int trans_a, int trans_b, const float* A, int lda, const float* Bm, int ldb, float* C,
                                   int ldc, int M, int N, int K, const float* bias, int relu, int accumulate,
                                   void* stream) {
    if (!A || !Bm || !C || M < 0 || N < 0 || K < 0 || (lda & 3) || (ldb & 3) || !al16(A) || !al16(Bm) || ldc < N ||
        lda < (trans_a ? M : K) || ldb < (trans_b ? K : N))
        return EVAD_ERR_ARG;
    GemmEpi epi{bias, nullptr, nullptr, nullptr, 0, relu, accumulate};
    return evad_gemm(trans_a != 0, trans_b != 0, A, lda, Bm, ldb, C, ldc, M, nullptr, N, K, epi, (hipStream_t)stream);
}

// =============================================================================================
// Grouped tile GEMM -- the head's workhorse.
// The head's products are small ([32..768] x [4..768] x [4..768]); as single launches of a 64x64- or
// 128x128-tiled kernel they fill 4..48 workgroups of a 256-CU chip and walk K serially (measured
// 15-30 us each).  Here up to 8 independent products go out in ONE launch.  A workgroup (4 waves) owns
// one 32x32 output tile; wave w takes the 64-deep K chunks w, w+4, ... (all 16 loads of a chunk in
// flight at once, private LDS staging, 32 MFMAs per chunk) and the four partial accumulators are
// summed through LDS in wave order (bitwise reproducible, no float atomics, no second launch).
// =============================================================================================
#define TG_MAXP 8
struct TProb {
    const float* A; const float* B; float* C; const float* bias; float* part; int* cnt;
    int lda, ldb, ldc, M, N, K, ta, tb, acc, relu, tiles_m, tiles_n, ksplit, kgroups, block0, keep_slabs;
};
struct TGroup { TProb p[TG_MAXP]; int n; };

__global__ __launch_bounds__(256) void k_tile_gemm(TGroup g) {
    // 16.9 KB of LDS: a wave stages its 64-deep K chunk as four 16-deep quarters through its own [16][33] A and B
    // buffers (all 16 loads of the chunk are issued first; a wave's LDS traffic is ordered, so no barrier between the
    // quarters), and the cross-wave reduction re-uses the staging memory.  With 80 KB (whole chunk staged at once) a
    // workgroup of this kernel displaced a layer-1 convolution workgroup of its CU whenever the two streams overlapped.
    __shared__ float Stage[8][16][33];
    float (*Red)[32][33] = (float (*)[32][33])Stage;       // [3][32][33] = 12.7 KB <= 16.9 KB, valid after the barrier below
    int pi = 0;
#pragma unroll
    for (int i = 1; i < TG_MAXP; ++i) if (i < g.n && (int)blockIdx.x >= g.p[i].block0) pi = i;
    const TProb& P = g.p[pi];
    const int local = blockIdx.x - P.block0;
    const int per_k = P.tiles_m * P.tiles_n;
    const int kg = local / per_k, rem = local % per_k, mt = rem / P.tiles_n, nt = rem % P.tiles_n;
    const int m0 = mt * 32, n0 = nt * 32, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*as)[33] = Stage[wave];
    float (*bs)[33] = Stage[4 + wave];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {   // one round: this workgroup's 4 K chunks (K > 256 is spread over several workgroups per tile)
        const int ks = kg * 4 + wave, k0 = ks * 64;
        const bool live = ks < P.ksplit;
        // quarter q covers k0 + 16q .. +15.  k-major operands ([K,M] / [K,N]): 16 k-rows x 8 float4; row-major
        // operands ([M,K] / [N,K]): 32 rows x 4 float4 -- 128 float4 per operand per quarter = 2 per lane.
        // Two halves of 8 loads: the second half is issued once the first sits in LDS, under the first half's MFMAs --
        // 32 load registers live at a time instead of 64, so a wave of this kernel (~60 VGPRs) also fits on a SIMD that
        // the backbone's row scan (112 x 4) or pool (88 x 5) has filled up to the last 64-72 registers.
        float4 ra[2][2], rb[2][2];
        auto fetch = [&](int q, int i, float4& va, float4& vb) {
            const int f = lane + 64 * i, kb = k0 + 16 * q;
            const int kr = f >> 3, kl = (f & 7) * 4;        // k-major: k row, 4 consecutive m/n
            const int rw = f >> 2, rl = (f & 3) * 4;        // row-major: m/n row, 4 consecutive k
            va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va;
            if (live) {
                if (P.ta) { if (kb + kr < P.K) va = ld4_guard(P.A + (size_t)(kb + kr) * P.lda + m0 + kl, m0 + kl, P.M); }
                else      { if (m0 + rw < P.M) va = ld4_guard(P.A + (size_t)(m0 + rw) * P.lda + kb + rl, kb + rl, P.K); }
                if (P.tb) { if (n0 + rw < P.N) vb = ld4_guard(P.B + (size_t)(n0 + rw) * P.ldb + kb + rl, kb + rl, P.K); }
                else      { if (kb + kr < P.K) vb = ld4_guard(P.B + (size_t)(kb + kr) * P.ldb + n0 + kl, n0 + kl, P.N); }
            }
        };
        auto stage = [&](int i, const float4& va, const float4& vb) {
            const int f = lane + 64 * i, kr = f >> 3, kl = (f & 7) * 4, rw = f >> 2, rl = (f & 3) * 4;
            if (P.ta) { as[kr][kl] = va.x; as[kr][kl + 1] = va.y; as[kr][kl + 2] = va.z; as[kr][kl + 3] = va.w; }
            else      { as[rl][rw] = va.x; as[rl + 1][rw] = va.y; as[rl + 2][rw] = va.z; as[rl + 3][rw] = va.w; }
            if (P.tb) { bs[rl][rw] = vb.x; bs[rl + 1][rw] = vb.y; bs[rl + 2][rw] = vb.z; bs[rl + 3][rw] = vb.w; }
            else      { bs[kr][kl] = vb.x; bs[kr][kl + 1] = vb.y; bs[kr][kl + 2] = vb.z; bs[kr][kl + 3] = vb.w; }
        };
        auto mfma16 = [&]() {
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                const int kr = kk + (lane >> 5), c = lane & 31;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[kr][c], bs[kr][c], acc, 0, 0, 0);
            }
        };
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int i = 0; i < 2; ++i) fetch(2 * half + qq, i, ra[qq][i], rb[qq][i]);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
                for (int i = 0; i < 2; ++i) stage(i, ra[qq][i], rb[qq][i]);
                mfma16();
            }
        }
        __syncthreads();                                    // every wave is done with its staging memory: Red may overwrite it
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Red[wave - 1][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int col = lane & 31, gn = n0 + col;
        if (gn < P.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gm = m0 + row;
                if (gm >= P.M) continue;
                float v = ((acc[r] + Red[0][row][col]) + Red[1][row][col]) + Red[2][row][col];
                if (P.kgroups > 1) { P.part[((size_t)kg * P.M + gm) * P.N + gn] = v; continue; }
                if (P.bias) v += P.bias[gn];
                if (P.relu) v = fmaxf(v, 0.f);
                if (P.acc) v += P.C[(size_t)gm * P.ldc + gn];
                P.C[(size_t)gm * P.ldc + gn] = v;
            }
        }
        // K > 256 with a tile counter: the workgroup that arrives last at the tile sums the slabs in slab order
        // (same order, hence same bits, as k_tile_reduce) -- no second launch, no float atomics, no spinning.
        if (P.kgroups > 1 && P.cnt) {
            __threadfence();                                 // release this wave's slab
            int old = 0;
            if (lane == 0) old = atomicAdd(&P.cnt[rem], 1);
            old = __shfl(old, 0, 64);
            if (old == P.kgroups - 1) {
                __threadfence();                             // acquire the other slabs
                if (gn < P.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gm = m0 + row;
                        if (gm >= P.M) continue;
                        float v = 0.f;
                        for (int q = 0; q < P.kgroups; ++q) v += __builtin_nontemporal_load(&P.part[((size_t)q * P.M + gm) * P.N + gn]);
                        if (P.bias) v += P.bias[gn];
                        if (P.relu) v = fmaxf(v, 0.f);
                        if (P.acc) v += P.C[(size_t)gm * P.ldc + gn];
                        P.C[(size_t)gm * P.ldc + gn] = v;
                    }
                }
                if (lane == 0) P.cnt[rem] = 0;               // ready for the next launch
            }
        }
    }
}

__global__ void k_tile_reduce(TGroup g) {
    int pi = blockIdx.y;
    if (pi >= g.n) return;
    const TProb& P = g.p[pi];
    if (P.kgroups == 1 || P.cnt || P.keep_slabs) return;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.M * P.N) return;
    int m = t / P.N, n = t % P.N;
    float v = 0.f;
    for (int q = 0; q < P.kgroups; ++q) v += P.part[((size_t)q * P.M + m) * P.N + n];
    if (P.bias) v += P.bias[n];
    if (P.relu) v = fmaxf(v, 0.f);
    if (P.acc) v += P.C[(size_t)m * P.ldc + n];
    P.C[(size_t)m * P.ldc + n] = v;
}

/* desc: HOST array, 16 int64 per problem:
 *   [0] A  [1] B  [2] part scratch f32[ceil(K/256)][M][N] (needed when K > 256)  [3] C  [4] bias (0 = none)  [5] lda  [6] ldb  [7] ldc  [8] M  [9] N  [10] K
 *   [11] trans_a  [12] trans_b  [13] accumulate  [14] relu
 *   [15] tile counters i32[ceil(M/32)*ceil(N/32)] (0 = none): zero on first use, left zero by every launch; with them the
 *        K > 256 slabs are summed by the last workgroup to reach each tile instead of by a second launch;
 *        the value 1 = leave the slabs of a K > 256 product in `part` (f32[ceil(K/256)][M][N]) un-summed: the consumer adds them
 *        (+ bias/accumulate are then the consumer's business too) */
extern "C" int eventad_gemm_group(int n_prob, const int64_t* desc, void* stream) {
    if (n_prob < 1 || n_prob > TG_MAXP || !desc) return EVAD_ERR_ARG;
    TGroup g;
    g.n = n_prob;
    int blocks = 0, max_mn = 0;
    for (int i = 0; i < n_prob; ++i) {
        const int64_t* d = desc + 16 * i;
        TProb& P = g.p[i];
        P.A = (const float*)d[0]; P.B = (const float*)d[1]; P.part = (float*)d[2]; P.C = (float*)d[3]; P.bias = (const float*)d[4];
        P.lda = (int)d[5]; P.ldb = (int)d[6]; P.ldc = (int)d[7]; P.M = (int)d[8]; P.N = (int)d[9]; P.K = (int)d[10];
        P.ta = (int)d[11]; P.tb = (int)d[12]; P.acc = (int)d[13]; P.relu = (int)d[14];
        P.keep_slabs = d[15] == 1; P.cnt = P.keep_slabs ? nullptr : (int*)d[15];
        if (!P.A || !P.B || !P.C || P.M < 1 || P.N < 1 || P.K < 1 || (P.lda & 3) || (P.ldb & 3) || !al16(P.A) || !al16(P.B) ||
            P.ldc < P.N || P.lda < (P.ta ? P.M : P.K) || P.ldb < (P.tb ? P.K : P.N))
            return EVAD_ERR_ARG;
        P.tiles_m = (P.M + 31) / 32; P.tiles_n = (P.N + 31) / 32; P.ksplit = (P.K + 63) / 64; P.block0 = blocks;
        P.kgroups = (P.ksplit + 3) / 4;
        if (P.kgroups > 1) { if (!P.part) return EVAD_ERR_ARG; if (!P.cnt && !P.keep_slabs && P.M * P.N > max_mn) max_mn = P.M * P.N; }
        blocks += P.tiles_m * P.tiles_n * P.kgroups;
    }
    hipLaunchKernelGGL(k_tile_gemm, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    if (max_mn) hipLaunchKernelGGL(k_tile_reduce, dim3(evad_blocks(max_mn, 256), n_prob), dim3(256), 0, (hipStream_t)stream, g);
    return evad_last_error();
}

// =============================================================================================
// Grouped GEMM on 64x64 tiles -- for the groups that are bound by MFMA throughput, not by latency.
// On the head's 32 CUs the batched weight gradients (dW = dG^T . X over all clips: six products, up to 768 x 256 x R)
// are 630 tiles of 32x32 in k_tile_gemm: every tile's operands are fetched by four waves that each hold a quarter of K,
// staged with 4-byte LDS writes and summed across the waves through LDS.  Here a workgroup owns a 64x64 tile, each of
// its 4 waves a 32x32 quadrant over the WHOLE K (a plain fmaf chain in k order: no cross-wave sum, no slabs, no second
// launch whatever K is), the operands go through LDS once per workgroup -- half the loads per flop -- as 16-deep slabs,
// double buffered, with the global loads of the next TWO slabs in flight (2 x 8 registers) while a slab multiplies.
// =============================================================================================
__global__ __launch_bounds__(256) void k_tile_gemm64(TGroup g) {
    __shared__ float As[2][GBK][64 + 4];
    __shared__ float Bs[2][GBK][64 + 4];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < TG_MAXP; ++i) if (i < g.n && (int)blockIdx.x >= g.p[i].block0) pi = i;
    const TProb& P = g.p[pi];
    const int local = blockIdx.x - P.block0;
    const int m0 = (local / P.tiles_n) * 64, n0 = (local % P.tiles_n) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int M = P.M, N = P.N, K = P.K;
    const bool ta = P.ta != 0, tb = P.tb != 0;
    // k-major operand ([K, M] / [K, N]): thread -> k row tid / 16, 4 consecutive m / n; row-major ([M, K] / [N, K]): row tid / 4,
    // 4 consecutive k
    const int kr = tid >> 4, cq = (tid & 15) * 4, rw = tid >> 2, kq = (tid & 3) * 4;
    const float* pa = ta ? P.A + (size_t)kr * P.lda + m0 + cq : P.A + (size_t)(m0 + rw) * P.lda + kq;
    const float* pb = tb ? P.B + (size_t)(n0 + rw) * P.ldb + kq : P.B + (size_t)kr * P.ldb + n0 + cq;
    const size_t sa = ta ? (size_t)GBK * P.lda : (size_t)GBK, sb = tb ? (size_t)GBK : (size_t)GBK * P.ldb;
    auto fetch = [&](int k0, float4& va, float4& vb) {
        va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va;
        if (k0 >= K) return;
        if (ta) { if (k0 + kr < K) va = ld4_guard(pa, m0 + cq, M); } else { if (m0 + rw < M) va = ld4_guard(pa, k0 + kq, K); }
        if (tb) { if (n0 + rw < N) vb = ld4_guard(pb, k0 + kq, K); } else { if (k0 + kr < K) vb = ld4_guard(pb, n0 + cq, N); }
        pa += sa; pb += sb;
    };
    auto stage = [&](int buf, const float4& va, const float4& vb) {
        if (ta) *(float4*)&As[buf][kr][cq] = va;
        else { As[buf][kq][rw] = va.x; As[buf][kq + 1][rw] = va.y; As[buf][kq + 2][rw] = va.z; As[buf][kq + 3][rw] = va.w; }
        if (tb) { Bs[buf][kq][rw] = vb.x; Bs[buf][kq + 1][rw] = vb.y; Bs[buf][kq + 2][rw] = vb.z; Bs[buf][kq + 3][rw] = vb.w; }
        else *(float4*)&Bs[buf][kr][cq] = vb;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int c = lane & 31, kh2 = lane >> 5;
    auto slab = [&](int buf) {
        float a[2], b[2];
        a[0] = As[buf][kh2][wm + c]; b[0] = Bs[buf][kh2][wn + c];
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            const int q = (kk >> 1) & 1;
            if (kk + 2 < GBK) { a[q ^ 1] = As[buf][kk + 2 + kh2][wm + c]; b[q ^ 1] = Bs[buf][kk + 2 + kh2][wn + c]; }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
        }
    };
    // slab s multiplies out of LDS buffer s & 1 while the registers hold slabs s + 1 (va1/vb1) and s + 2 (va2/vb2)
    float4 va0, vb0, va1, vb1, va2, vb2;
    fetch(0, va0, vb0);
    fetch(GBK, va1, vb1);
    stage(0, va0, vb0);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += 2 * GBK) {
        fetch(k0 + 2 * GBK, va2, vb2);
        slab(0);
        if (k0 + GBK < K) stage(1, va1, vb1);
        __syncthreads();
        if (k0 + GBK >= K) break;
        fetch(k0 + 3 * GBK, va1, vb1);
        slab(1);
        if (k0 + 2 * GBK < K) stage(0, va2, vb2);
        __syncthreads();
    }
    const int gn = n0 + wn + c;
    if (gn >= N) return;
    const float bias = P.bias ? P.bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh2;
        if (gm >= M) continue;
        float v = acc[r] + bias;
        if (P.relu) v = fmaxf(v, 0.f);
        if (P.acc) v += P.C[(size_t)gm * P.ldc + gn];
        P.C[(size_t)gm * P.ldc + gn] = v;
    }
}

/* Same descriptors as eventad_gemm_group ([2] and [15] are not used: there are no slabs). */
extern "C" int eventad_gemm_group64(int n_prob, const int64_t* desc, void* stream) {
    if (n_prob < 1 || n_prob > TG_MAXP || !desc) return EVAD_ERR_ARG;
    TGroup g;
    g.n = n_prob;
    int blocks = 0;
    for (int i = 0; i < n_prob; ++i) {
        const int64_t* d = desc + 16 * i;
        TProb& P = g.p[i];
        P.A = (const float*)d[0]; P.B = (const float*)d[1]; P.part = nullptr; P.C = (float*)d[3]; P.bias = (const float*)d[4];
        P.lda = (int)d[5]; P.ldb = (int)d[6]; P.ldc = (int)d[7]; P.M = (int)d[8]; P.N = (int)d[9]; P.K = (int)d[10];
        P.ta = (int)d[11]; P.tb = (int)d[12]; P.acc = (int)d[13]; P.relu = (int)d[14];
        P.keep_slabs = 0; P.cnt = nullptr;
        if (!P.A || !P.B || !P.C || P.M < 1 || P.N < 1 || P.K < 1 || (P.lda & 3) || (P.ldb & 3) || !al16(P.A) || !al16(P.B) ||
            P.ldc < P.N || P.lda < (P.ta ? P.M : P.K) || P.ldb < (P.tb ? P.K : P.N))
            return EVAD_ERR_ARG;
        P.tiles_m = (P.M + 63) / 64; P.tiles_n = (P.N + 63) / 64; P.ksplit = 1; P.kgroups = 1; P.block0 = blocks;
        blocks += P.tiles_m * P.tiles_n;
    }
    hipLaunchKernelGGL(k_tile_gemm64, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    return evad_last_error();
}
