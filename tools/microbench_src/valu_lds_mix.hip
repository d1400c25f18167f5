// What the instruction mix of the layer-1 convolution can reach on the chip: every lane runs steps of R ds_read_b128 (16
// bytes each, conflict-free, a different address per step) feeding 4*R non-packed v_fma_f32, plus X integer VALU instructions
// per step, 16 waves per CU (2 workgroups of 8 waves, 64 KiB of LDS each), nothing else -- no global loads, no gathers.
// usage: valu_lds_mix [R reads per step = 20] [X extra VALU per step = 0] [reads on: 1|0] [mode 0|1|2, see k_mix]
// Prints the time per step and wave, the chip's FMA rate, and the shader clock measured over the run (s_memtime ticks per
// 100 MHz s_memrealtime tick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// MODE 0: reads written next to their products, two channels per scheduling region (the compiler interleaves them);
// MODE 1: one channel's four reads requested while the previous channel's products run, the pipeline drained at the end of
//         every step (what the kernel does per edge);  MODE 2: the same, but the next step's first channel is requested
//         during the last channel's products (no drain between steps).
template <int R, int X, bool READS, int MODE>
__global__ __launch_bounds__(512, 4) void k_mix(float* out, unsigned long long* clk, int steps, unsigned stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int t = threadIdx.x; t < 16384; t += 512) lds[t] = 1e-6f * (float)t;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    constexpr int KL = R / 4;
    float4 acc[4];
    for (int s = 0; s < 4; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    float x[5] = {1.0f + lane * 1e-3f, 0.5f, 0.25f, 0.125f, 2.0f};
    unsigned a = (unsigned)lane * 16u, junk = (unsigned)lane;
    auto rd = [&](unsigned base, int i, int s) {
        float4 w;
        if (READS) w = *(const float4*)((const char*)lds + ((base + (unsigned)(i * 4 + s) * 1024u) & 0xFFFFu));
        else { w = make_float4(x[1], x[2], x[3], x[4]); asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w)); }
        return w;
    };
    auto fma4 = [&](int i, int s, const float4& w) {
        acc[s].x = fmaf(x[i], w.x, acc[s].x); acc[s].y = fmaf(x[i], w.y, acc[s].y);
        acc[s].z = fmaf(x[i], w.z, acc[s].z); acc[s].w = fmaf(x[i], w.w, acc[s].w);
    };
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float4 w[2][4];
    if (MODE == 2) {
        a = (a + stride) & 0xFFFFu;
#pragma unroll
        for (int s = 0; s < 4; ++s) w[0][s] = rd(a, 0, s);
    }
    for (int it = 0; it < steps; ++it) {
        if (MODE == 0) {
            a = (a + stride) & 0xFFFFu;                      // a new 64 KiB-wrapped base per step (a multiple of 1 KiB apart: conflict-free)
#pragma unroll
            for (int i = 0; i < KL; ++i) {
#pragma unroll
                for (int s = 0; s < 4; ++s) fma4(i, s, rd(a, i, s));
                if (i & 1) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            unsigned an = a;
            if (MODE == 1) {
                a = (a + stride) & 0xFFFFu;
#pragma unroll
                for (int s = 0; s < 4; ++s) w[0][s] = rd(a, 0, s);
            } else an = (a + stride) & 0xFFFFu;
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                // buffers alternate per channel; with an odd channel count the roles swap every step, so the step is unrolled
                // twice below for MODE 2 (KL even: no swap)
                if (i + 1 < KL) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) w[(i + 1) & 1][s] = rd(a, i + 1, s);
                } else if (MODE == 2) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) w[(i + 1) & 1][s] = rd(an, 0, s);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) fma4(i, s, w[i & 1][s]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 2) {
                a = an;
                if (KL & 1) {                                // odd channel count: the prefetched channel 0 sits in buffer 1: copy (4 moves
#pragma unroll                                               // per chunk; a real kernel would unroll two steps instead)
                    for (int s = 0; s < 4; ++s) w[0][s] = w[1][s];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < X; ++k) { junk = junk * 2u + (unsigned)k; asm volatile("" : "+v"(junk)); }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float sum = (float)junk;
    for (int s = 0; s < 4; ++s) sum += acc[s].x + acc[s].y + acc[s].z + acc[s].w;
    if (MODE == 2) sum += w[0][0].x;
    if (sum == 12345.678f) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int R, int X, bool READS, int MODE> static void run(int cus) {
    const int blocks = cus * 2, steps = 20000;
    float* out; unsigned long long* clk; hipMalloc(&out, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mix<R, X, READS, MODE>), dim3(blocks), dim3(512), 65536, 0, out, clk, steps, 1024u * 5u);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double waves = (double)blocks * 8, fma = waves * steps * (double)R * 4 * 64;
        printf("mode %d R=%d reads(%s) X=%d: %.3f ms, %.1f ns per step and wave, %.2f TFLOP/s FMA, LDS %.1f TB/s, shader clock %.0f MHz\n", MODE, R,
               READS ? "on" : "off", X, ms, ms * 1e6 / steps, 2.0 * fma / ms / 1e9, READS ? waves * steps * (double)R * 1024 / ms / 1e9 : 0.0,
               100.0 * (double)h[0] / (double)h[1]);
    }
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 20, X = argc > 2 ? atoi(argv[2]) : 0, reads = argc > 3 ? atoi(argv[3]) : 1, mode = argc > 4 ? atoi(argv[4]) : 0;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
#define CASE(r, x) if (R == r && X == x) { \
        if (!reads) run<r, x, false, 0>(cus); else if (mode == 0) run<r, x, true, 0>(cus); else if (mode == 1) run<r, x, true, 1>(cus); else run<r, x, true, 2>(cus); \
        return 0; }
    CASE(20, 0) CASE(20, 16) CASE(20, 32) CASE(20, 60) CASE(16, 0) CASE(16, 16) CASE(16, 32)
    printf("no such case\n");
    return 1;
}
