// What a grid-wide barrier between 32 workgroups costs (the shape of the head's stage kernels: 32 x 1024 threads, one
// workgroup per CU because of its LDS, on a CU-masked stream of 4 CUs per XCD): per round every workgroup writes 4 KB, all
// meet at a counter in device memory (release fence, atomic add, bounded spin, acquire fence), then every thread reads
// what another workgroup wrote in this round and checks it.  The alternative is a kernel boundary per round (measured
// with the same body as one launch per round).  Every wait is bounded: a workgroup that waits longer than 50 ms sets a
// flag and every workgroup leaves.   usage: grid_barrier [rounds] [masked 0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define WGS 32
#define TPB 1024
__device__ __forceinline__ bool grid_sync(unsigned* counter, unsigned target, unsigned* timeout_flag) {
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __threadfence();                                     // release: this workgroup's stores before the arrival
        atomicAdd(counter, 1u);
        const long long t0 = wall_clock64();                 // 100 MHz
        int good = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { good = 0; break; }
            if (wall_clock64() - t0 > 5000000ll) { atomicExch(timeout_flag, 1u); good = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        ok = good;
    }
    __syncthreads();
    __threadfence();                                         // acquire: the other workgroups' stores after the arrival
    return ok != 0;
}
__global__ __launch_bounds__(TPB) void k_rounds(float* buf, unsigned* counter, unsigned* flags, int rounds, int round0, int use_barrier) {
    extern __shared__ float lds[];
    const int g = blockIdx.x, tid = threadIdx.x;
    lds[tid] = 0.f;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const int it = round0 + r;
        float* cur = buf + (size_t)(it & 1) * WGS * TPB;
        cur[g * TPB + tid] = (float)(it * 64 + g);
        if (use_barrier) { if (!grid_sync(counter, (unsigned)WGS * (unsigned)(it + 1), flags)) break; }
        else break;                                          // one round per launch: the kernel boundary is the barrier
        const int src = (g + 1 + (tid % (WGS - 1))) % WGS;
        const float v = __builtin_nontemporal_load(cur + src * TPB + tid);
        if (v != (float)(it * 64 + src)) ++bad;
        lds[tid] += v;
    }
    if (!use_barrier && round0 > 0) {                         // check the previous launch's round
        const int it = round0 - 1, src = (g + 1 + (tid % (WGS - 1))) % WGS;
        const float v = buf[(size_t)(it & 1) * WGS * TPB + src * TPB + tid];
        if (v != (float)(it * 64 + src)) ++bad;
    }
    if (bad) atomicAdd(flags + 1, bad);
    if (lds[tid] == 12345.f) buf[0] = 1.f;
}
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200, masked = argc > 2 ? atoi(argv[2]) : 1;
    float* buf; unsigned* ctl;
    hipMalloc(&buf, 2 * WGS * TPB * sizeof(float)); hipMalloc(&ctl, 64);
    hipStream_t st;
    if (masked) { uint32_t mask[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}; if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("mask failed\n"); return 1; } }
    else hipStreamCreate(&st);
    const int lds = 145 * 1024;
    if (hipFuncSetAttribute((const void*)k_rounds, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("lds attr failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {                   // 0: one launch per round, 1: one launch, barrier per round
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(ctl, 0, 64, st); hipMemsetAsync(buf, 0, 2 * WGS * TPB * sizeof(float), st);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            if (mode == 0) for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_rounds, dim3(WGS), dim3(TPB), lds, st, buf, ctl, ctl + 1, 1, r, 0);
            else hipLaunchKernelGGL(k_rounds, dim3(WGS), dim3(TPB), lds, st, buf, ctl, ctl + 1, rounds, 0, 1);
            hipEventRecord(e1, st);
            if (hipStreamSynchronize(st) != hipSuccess) { printf("sync failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned h[3]; hipMemcpy(h, ctl, 12, hipMemcpyDeviceToHost);
            printf("%s masked=%d: %.2f us per round (%d rounds), timeout flag %u, wrong values %u\n", mode ? "grid barrier" : "launch per round", masked,
                   ms * 1e3f / rounds, rounds, h[1], h[2]);
            if (h[1]) return 1;
        }
    }
    return 0;
}
