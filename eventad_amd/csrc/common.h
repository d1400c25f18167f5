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
