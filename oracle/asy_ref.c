/*
 * oracle/asy_ref.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement, in plain C, of the four kernels of the reference's asynchronous-inference extension `asy_tools`
 * (paths relative to /root/reference):
 *
 *   ora_masked_isdiff       src/dagr/asynchronous/asy_tools/main.cu:14-41   (host wrapper :96-125 drops the -1 entries)
 *   ora_masked_inplace_bn   src/dagr/asynchronous/asy_tools/main.cu:43-67
 *   ora_masked_lin          src/dagr/asynchronous/asy_tools/main.cu:128-188 (bias == NULL: masked_lin_no_bias)
 *
 * One GPU thread per (row, channel) in the reference; the same bodies run in serial loops here.  a*b+c is one fused
 * multiply-add (nvcc contracts by default); the file must be compiled with -ffp-contract=off so that nothing else is.
 *
 * PARITY UNPINNED w.r.t. reference outputs: the reference holds no tests or fixtures for these kernels and the
 * extension cannot be compiled here (needs CUDA); pinned only by hand-derived cases in tests/test_oracle_asy.py.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

void ora_masked_isdiff(int64_t* indices, const float* x_old, const float* x_new, int K, int C, float atol, float rtol) {
    for (int i = 0; i < K; ++i) {
        int64_t temp = indices[i];
        indices[i] = -1;
        int64_t offset = temp * C;
        for (int c = 0; c < C; ++c) {
            float input = x_old[offset + c], other = x_new[offset + c];
            if (fabsf(input - other) > fmaf(rtol, other, atol)) { indices[i] = temp; break; }
        }
    }
}

void ora_masked_inplace_bn(const int64_t* indices, const float* x, float* x_out, const float* running_mean,
                           const float* running_var, const float* weight, const float* bias, int K, int C, float eps) {
    for (int i = 0; i < K; ++i)
        for (int c = 0; c < C; ++c) {
            int64_t o = (int64_t)C * indices[i] + c;
            x_out[o] = fmaf((x[o] - running_mean[c]) / sqrtf(running_var[c] + eps), weight[c], bias[c]);
        }
}

void ora_masked_lin(const int64_t* indices, const float* x_in, float* x_out, const float* weight, const float* bias,
                    int K, int Cin, int Cout, int add) {
    for (int i = 0; i < K; ++i)
        for (int co = 0; co < Cout; ++co) {
            int64_t o = (int64_t)Cout * indices[i] + co, xi = (int64_t)Cin * indices[i];
            if (!add) x_out[o] = 0;
            for (int ci = 0; ci < Cin; ++ci) x_out[o] = fmaf(x_in[xi + ci], weight[(size_t)co * Cin + ci], x_out[o]);
            if (bias) x_out[o] += bias[co];
        }
}
