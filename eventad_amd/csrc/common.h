// eventad_amd/csrc/common.h -- shared helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EVAD_OK 0
#define EVAD_ERR_ARG (-1)

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

// order-preserving float <-> uint map so float max can use integer atomicMax (0 < every real)
__device__ __forceinline__ unsigned evad_f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float evad_ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
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
