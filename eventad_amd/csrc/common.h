// eventad_amd/csrc/common.h -- shared helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EVAD_OK 0
#define EVAD_ERR_ARG (-1)
#define EVENTAD_ERR_TOO_LARGE (-3)
#define EVAD_SCAN_TILE 1024         // slots per workgroup of the tiled slot scans (pool.hip)

// error-flag bits written by kernels into a caller-owned u32 word (checked lazily by the host)
#define EVAD_FLAG_T_NOT_MONOTONE 1u   // timestamps not non-decreasing in event order inside a clip
#define EVAD_FLAG_POS_OUT_OF_FOV 2u   // an event fell outside [0,W)x[0,H) or batch outside [0,B)
#define EVAD_FLAG_POOL_WINDOW    4u   // a pooled edge fell outside the 5x5x3 neighbour window
#define EVAD_FLAG_T_OUT_OF_RANGE 8u   // normalised t outside [0,1]

static inline int evad_last_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? EVAD_OK : (int)e;
}

static inline unsigned evad_blocks(long long n, int threads) {
    long long b = (n + threads - 1) / threads;
    return (unsigned)(b < 1 ? 1 : b);
}

// Clear device memory with a fill KERNEL rather than a memset node: every entry point of this library may be recorded
// into a hipGraph by its caller, and memset nodes recorded on this platform have replayed wrongly after the first replay.
// p must be 4-byte aligned and bytes a multiple of 4.
static __global__ void k_evad_zero(uint32_t* __restrict__ p, size_t words) {
    const size_t head = min(words, (size_t)((16 - ((uintptr_t)p & 15)) & 15) >> 2);      // words before 16-byte alignment
    const size_t vec = (words - head) >> 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4* q = (uint4*)(p + head);
    for (size_t k = i; k < vec; k += (size_t)gridDim.x * blockDim.x) q[k] = make_uint4(0u, 0u, 0u, 0u);
    if (i < head) p[i] = 0u;
    const size_t tail0 = head + (vec << 2);
    if (i < words - tail0) p[tail0 + i] = 0u;
}
static inline void evad_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return;
    const size_t words = bytes >> 2;
    size_t blocks = (words / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_evad_zero, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, words);
}

// order-preserving float <-> uint map so float max can use integer atomicMax (0 < every real)
__device__ __forceinline__ unsigned evad_f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float evad_ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Bilinear sampling set-up of F.grid_sample(align_corners=True) as the reference calls it (net.py:211-228): normalised
// event position -> top-left map pixel + fractions, every step rounded like the reference's separate torch ops.
__device__ __forceinline__ void evad_bilinear_setup(float px_norm, float py_norm, float Wf, float Hf, int w_in, int h_in,
                                               int* x0, int* y0, float* fx, float* fy) {
    float x = __fmul_rn(px_norm, Wf), y = __fmul_rn(py_norm, Hf);
    float xn = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, x), Wf - 1.f), 1.f);
    float yn = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, y), Hf - 1.f), 1.f);
    float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xn, 1.f), 2.f), (float)(w_in - 1));
    float iy = __fmul_rn(__fdiv_rn(__fadd_rn(yn, 1.f), 2.f), (float)(h_in - 1));
    float fx0 = floorf(ix), fy0 = floorf(iy);
    *x0 = (int)fx0; *y0 = (int)fy0; *fx = ix - fx0; *fy = iy - fy0;
}

// Pooled-edge dedup (pool.hip): bit index of a source slot relative to its destination slot in the
// destination's 5x5x3 neighbour-window bitmap: ((dp+1)*5 + (dy+2))*5 + (dx+2).
__device__ __forceinline__ void evad_mark_edge(int ss, int sd, int nx, int ny, unsigned* __restrict__ bitmap, unsigned* flags) {
    if (ss == sd) return;                                   // self loop after clustering: dropped (pooling.py:62-64)
    int plane = nx * ny;
    int dx = ss % nx - sd % nx, dy = (ss / nx) % ny - (sd / nx) % ny, dp = ss / plane - sd / plane;
    if (dx < -2 || dx > 2 || dy < -2 || dy > 2 || dp < -1 || dp > 1) { atomicOr(flags, EVAD_FLAG_POOL_WINDOW); return; }
    int bit = ((dp + 1) * 5 + (dy + 2)) * 5 + (dx + 2);
    unsigned* w = bitmap + (size_t)sd * 3 + (bit >> 5);
    unsigned m = 1u << (bit & 31);
    if (!(*(volatile unsigned*)w & m)) atomicOr(w, m);
}

// same with pre-decomposed voxel coordinates packed as cx | cy << 8 | plane << 16 (no integer divisions)
__device__ __forceinline__ void evad_mark_edge_packed(int vs, int vd, int slot_d, unsigned* __restrict__ bitmap, unsigned* flags) {
    if (vs == vd) return;
    int dx = (vs & 255) - (vd & 255), dy = ((vs >> 8) & 255) - ((vd >> 8) & 255), dp = (vs >> 16) - (vd >> 16);
    if (dx < -2 || dx > 2 || dy < -2 || dy > 2 || dp < -1 || dp > 1) { atomicOr(flags, EVAD_FLAG_POOL_WINDOW); return; }
    int bit = ((dp + 1) * 5 + (dy + 2)) * 5 + (dx + 2);
    unsigned* w = bitmap + (size_t)slot_d * 3 + (bit >> 5);
    unsigned m = 1u << (bit & 31);
    if (!(*(volatile unsigned*)w & m)) atomicOr(w, m);
}

// The same for the <= 15 edges of one ELL row at once: all bitmap words are fetched first (independent loads -- the
// volatile test-then-set above serialises one round trip per edge), then only the missing bits are OR-ed in.
// A stale read costs at most a redundant atomicOr.
template <int MAXNB>
__device__ __forceinline__ void evad_mark_row_packed(const int (&vs)[MAXNB], int nn, int vd, int slot_d, unsigned* __restrict__ bitmap,
                                                     unsigned* flags) {
    unsigned* w[MAXNB]; unsigned m[MAXNB], cur[MAXNB];
    bool bad = false;
#pragma unroll
    for (int e = 1; e < MAXNB; ++e) {
        w[e] = nullptr; m[e] = 0; cur[e] = 0;
        if (e < nn && vs[e] != vd) {
            int dx = (vs[e] & 255) - (vd & 255), dy = ((vs[e] >> 8) & 255) - ((vd >> 8) & 255), dp = (vs[e] >> 16) - (vd >> 16);
            if (dx < -2 || dx > 2 || dy < -2 || dy > 2 || dp < -1 || dp > 1) bad = true;
            else {
                int bit = ((dp + 1) * 5 + (dy + 2)) * 5 + (dx + 2);
                w[e] = bitmap + (size_t)slot_d * 3 + (bit >> 5); m[e] = 1u << (bit & 31);
                cur[e] = __builtin_nontemporal_load(w[e]);
            }
        }
    }
    if (bad) atomicOr(flags, EVAD_FLAG_POOL_WINDOW);
    unsigned set0 = 0, set1 = 0, set2 = 0;                   // several edges usually hit the same bit: OR each word once
    unsigned* base = bitmap + (size_t)slot_d * 3;
#pragma unroll
    for (int e = 1; e < MAXNB; ++e) if (w[e] && !(cur[e] & m[e])) { int wi = (int)(w[e] - base); if (wi == 0) set0 |= m[e]; else if (wi == 1) set1 |= m[e]; else set2 |= m[e]; }
    // events are pixel-sorted, so the lanes of a wave fall into a few runs with the same destination voxel: OR the
    // masks down each run and let its first lane issue the atomics (3 words per run instead of per lane).
    // Must be reached by every lane of the wave (callers pass nn = 0 for idle lanes).
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int os = __shfl_down(slot_d, d, 64);
        const unsigned o0 = __shfl_down(set0, d, 64), o1 = __shfl_down(set1, d, 64), o2 = __shfl_down(set2, d, 64);
        if (lane + d < 64 && os == slot_d) { set0 |= o0; set1 |= o1; set2 |= o2; }
    }
    const int prev = __shfl_up(slot_d, 1, 64);
    if (lane == 0 || prev != slot_d) {
        if (set0) atomicOr(base, set0);
        if (set1) atomicOr(base + 1, set1);
        if (set2) atomicOr(base + 2, set2);
    }
}
