// eventad_amd/csrc/asy.hip -- index-masked row kernels of DAGR's asynchronous (event-by-event) inference.
//
// Reference: the CUDA extension `asy_tools` (src/dagr/asynchronous/asy_tools/main.cu:14-188), called from
// asynchronous/batch_norm.py:40, conv.py:236-238, linear.py:46-48 and base/utils.py:42: after a new event arrives only
// the rows listed in an index vector are recomputed.  Same results bit for bit as the serial restatement in
// oracle/asy_ref.c (IEEE division and square root -- sqrtf, not __fsqrt_rn, which this toolchain maps to the approximate
// native square root --, products and sums in the reference's order, a*b+c contracted to one
// fused multiply-add as nvcc does by default).
//
// Mapping: one 64-lane wave per listed row, lanes = output channels (the reference runs one thread per (row, channel)
// with a stride-Cin weight walk per thread; here the row of x is read once per wave and broadcast).
#include "common.h"

#define ASY_WAVES 4

// masked_isdiff (main.cu:14-41): indices[i] <- -1 unless some channel of row indices[i] differs between x_old and x_new
// by more than atol + rtol * x_new (no absolute value on x_new, as in the reference).
__global__ __launch_bounds__(64 * ASY_WAVES) void k_masked_isdiff(long long* __restrict__ indices, const float* __restrict__ x_old,
                                                                  const float* __restrict__ x_new, int K, int C, float atol, float rtol) {
    const int i = blockIdx.x * ASY_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= K) return;
    const long long row = indices[i];
    bool diff = false;
    for (int c = lane; c < C; c += 64) {
        const float a = x_old[row * C + c], b = x_new[row * C + c];
        diff |= fabsf(a - b) > fmaf(rtol, b, atol);
    }
    if (!__any(diff) && lane == 0) indices[i] = -1;
}

// masked_inplace_BN (main.cu:43-67): x_out[row] = (x[row] - mean) / sqrt(var + eps) * weight + bias for the listed rows.
__global__ void k_masked_bn(const long long* __restrict__ indices, const float* __restrict__ x, float* __restrict__ x_out,
                            const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ weight,
                            const float* __restrict__ bias, int K, int C, float eps) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)K * C) return;
    const int i = (int)(t / C), c = (int)(t % C);
    const long long o = indices[i] * C + c;
    x_out[o] = fmaf(__fdiv_rn(__fsub_rn(x[o], mean[c]), sqrtf(__fadd_rn(var[c], eps))), weight[c], bias[c]);
}

// masked_lin / masked_lin_no_bias (main.cu:128-188): x_out[row] = (add ? x_out[row] : 0) + weight . x_in[row] (+ bias),
// weight f32[Cout, Cin] (nn.Linear layout), channel sums accumulated in ascending input-channel order.
__global__ __launch_bounds__(64 * ASY_WAVES) void k_masked_lin(const long long* __restrict__ indices, const float* __restrict__ x_in,
                                                               float* __restrict__ x_out, const float* __restrict__ weight,
                                                               const float* __restrict__ bias, int K, int Cin, int Cout, int add) {
    const int i = blockIdx.x * ASY_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= K) return;
    const long long row = indices[i];
    const float* xr = x_in + row * Cin;
    for (int co = lane; co < Cout; co += 64) {
        float acc = add ? x_out[row * Cout + co] : 0.f;
        const float* w = weight + (size_t)co * Cin;
        for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xr[ci], w[ci], acc);
        if (bias) acc = __fadd_rn(acc, bias[co]);
        x_out[row * Cout + co] = acc;
    }
}

extern "C" int eventad_masked_isdiff(int64_t* indices, const float* x_old, const float* x_new, int K, int C, float atol,
                                     float rtol, void* stream) {
    if (K < 0 || C <= 0 || (K > 0 && (!indices || !x_old || !x_new))) return EVAD_ERR_ARG;
    if (K == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_masked_isdiff, dim3(evad_blocks(K, ASY_WAVES)), dim3(64 * ASY_WAVES), 0, (hipStream_t)stream,
                       (long long*)indices, x_old, x_new, K, C, atol, rtol);
    return evad_last_error();
}

extern "C" int eventad_masked_inplace_bn(const int64_t* indices, const float* x, float* x_out, const float* running_mean,
                                         const float* running_var, const float* weight, const float* bias, int K, int C,
                                         float eps, void* stream) {
    if (K < 0 || C <= 0 || (K > 0 && (!indices || !x || !x_out || !running_mean || !running_var || !weight || !bias))) return EVAD_ERR_ARG;
    if (K == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_masked_bn, dim3(evad_blocks((long long)K * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)indices, x, x_out, running_mean, running_var, weight, bias, K, C, eps);
    return evad_last_error();
}

extern "C" int eventad_masked_lin(const int64_t* indices, const float* x_in, float* x_out, const float* weight,
                                  const float* bias, int K, int Cin, int Cout, int add, void* stream) {
    if (K < 0 || Cin <= 0 || Cout <= 0 || (K > 0 && (!indices || !x_in || !x_out || !weight))) return EVAD_ERR_ARG;
    if (K == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_masked_lin, dim3(evad_blocks(K, ASY_WAVES)), dim3(64 * ASY_WAVES), 0, (hipStream_t)stream,
                       (const long long*)indices, x_in, x_out, weight, bias, K, Cin, Cout, add);
    return evad_last_error();
}
