// What takes the fp32 MFMA rate of a 128x128 tile loop from the register-only 155 TFLOP/s down to the ~100 TFLOP/s of the
// convolution kernel?  The loop of k_conv_f32<2,2> rebuilt one ingredient at a time:
//   mode 0: 32 MFMAs per slab, operands in registers          mode 1: + operands read from LDS (4 ds_read_b32 per 4 MFMAs)
//   mode 2: + one workgroup barrier per slab                   mode 3: + the slab's LDS staging writes (8 b32 + 2 b128 per thread)
//   mode 4: + the slab's global loads (4 x 16 B per thread, a buffer that stays in L2)
//   mode 5: the same loads streaming through 1 MiB per workgroup (HBM latency)   mode 6: + 64 dependent integer VALU ops per slab
//   (the loader's address arithmetic)
// usage: mfma_steps <blocks per CU>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_steps(float* out, const float4* __restrict__ src, int slabs, int rnd) {
    __shared__ float As[2][16][132];
    __shared__ float Bs[2][16][132];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, c = lane & 31, kh2 = lane >> 5;
    for (int i = tid; i < 2 * 16 * 132; i += 256) {
        if (rnd) {                                           // operands with busy mantissas: the data a real product sees
            unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            (&As[0][0][0])[i] = ((h & 0xffffff) / 16777216.f - 0.5f) * 1e-2f;
            h *= 3266489917u; h ^= h >> 16;
            (&Bs[0][0][0])[i] = ((h & 0xffffff) / 16777216.f - 0.5f) * 1e-2f;
        } else { (&As[0][0][0])[i] = 1e-3f * (i & 15); (&Bs[0][0][0])[i] = 1e-3f * (i & 7); }
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[2], rb[2];
    ra[0] = ra[1] = rb[0] = rb[1] = make_float4(1.f, 2.f, 3.f, 4.f);
    const float4* p = src + (size_t)blockIdx.x * 65536 + tid;
    int cur = 0;
    for (int s = 0; s < slabs; ++s) {
        if (MODE >= 4) {
            const int o = MODE >= 5 ? (s & 63) * 1024 : (s & 3) * 1024;
            ra[0] = p[o]; ra[1] = p[o + 256]; rb[0] = p[o + 512]; rb[1] = p[o + 768];
        }
        if (MODE >= 6) {
            int v = s + tid;
#pragma unroll
            for (int u = 0; u < 64; ++u) v = v * 3 + u;
            if (v == 123456789) rb[1].x += 1.f;
        }
        float a[2][2], b[2][2];
        if (MODE >= 1) {
            a[0][0] = As[cur][kh2][wm + c]; a[0][1] = As[cur][kh2][wm + 32 + c];
            b[0][0] = Bs[cur][kh2][wn + c]; b[0][1] = Bs[cur][kh2][wn + 32 + c];
        } else { a[0][0] = 1.f + lane; a[0][1] = 2.f; b[0][0] = 0.5f; b[0][1] = 0.25f; a[1][0] = 1.5f; a[1][1] = 2.5f; b[1][0] = 0.75f; b[1][1] = 0.125f; }
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const int q = (kk >> 1) & 1;
            if (MODE >= 1 && kk + 2 < 16) {
                a[q ^ 1][0] = As[cur][kk + 2 + kh2][wm + c]; a[q ^ 1][1] = As[cur][kk + 2 + kh2][wm + 32 + c];
                b[q ^ 1][0] = Bs[cur][kk + 2 + kh2][wn + c]; b[q ^ 1][1] = Bs[cur][kk + 2 + kh2][wn + 32 + c];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
        }
        if (MODE >= 3) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int f = tid + t * 256, row = f >> 2, kq = (f & 3) * 4, kr = f / 32, nq = (f % 32) * 4;
                const float4 v = ra[t];
                As[cur ^ 1][kq][row] = v.x; As[cur ^ 1][kq + 1][row] = v.y; As[cur ^ 1][kq + 2][row] = v.z; As[cur ^ 1][kq + 3][row] = v.w;
                *(float4*)&Bs[cur ^ 1][kr][nq] = rb[t];
            }
        }
        if (MODE >= 2) __syncthreads();
        if (MODE >= 3) cur ^= 1;
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 12345.678f) out[0] = sum;
}
template <int MODE> static void run(int blocks, float* out, const float4* src, int slabs, int rnd = 0) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_steps<MODE>, dim3(blocks), dim3(256), 0, 0, out, src, slabs, rnd);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * slabs * 32.0 * (32.0 * 32 * 2 * 2);
    printf("mode %d%s: %.2f ms, %.1f TFLOP/s\n", MODE, rnd ? " (random operands)" : "", best, flops / best / 1e9);
}
int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount * per_cu, slabs = 4000;
    float* out; (void)hipMalloc(&out, 4);
    float4* src; (void)hipMalloc(&src, (size_t)blocks * 65536 * sizeof(float4)); (void)hipMemset(src, 0, (size_t)blocks * 65536 * sizeof(float4));
    printf("%d CUs, %d workgroups of 4 waves per CU\n", pr.multiProcessorCount, per_cu);
    run<0>(blocks, out, src, slabs); run<1>(blocks, out, src, slabs); run<2>(blocks, out, src, slabs);
    run<3>(blocks, out, src, slabs); run<4>(blocks, out, src, slabs); run<5>(blocks, out, src, slabs); run<6>(blocks, out, src, slabs);
    run<2>(blocks, out, src, slabs, 1); run<2>(blocks, out, src, 4 * slabs, 1);
    return 0;
}
