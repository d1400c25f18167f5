// eventad_amd/csrc/cnn.hip -- the CNN RGB-feature branch (ResNet-50 trunk + 1x1 reducers) as implicit GEMM.
//
// Reference: HookModule.forward over torchvision resnet50 (src/dagr/model/networks/net_img.py:122-134,
// net.py:41-49): cuDNN convolutions + BatchNorm + ReLU in NCHW.  Here activations are NHWC fp32 and
// every convolution is one exact-fp32 MFMA GEMM
//      out[m, co] = epilogue( sum_k A[m, k] * Wk[k, co] ),   m = (b, oh, ow),  k = (kh, kw, ci)
// whose A operand is never materialised: the loader computes the input address of each (m, k) on the
// fly (zero outside the image).  Eval-mode BatchNorm is folded into the epilogue's per-channel
// scale/shift, ReLU and the bottleneck's residual add are fused there too.  Same tile structure as
// gemm.hip (128x128 or 64x64 block tile, v_mfma_f32_32x32x2_f32, k-major LDS slabs of 16).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CBK 16

struct ConvP {
    const float* in; const float* w; float* out;
    const float* bias; const float* scale; const float* shift; const float* residual;
    int B, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, relu;
    float* part; int ksplit, kchunk;       // split-K: slab z covers k in [z*kchunk, (z+1)*kchunk) and goes to part[z][M][N]
};

// C16: Cin % 16 == 0 (every convolution of the trunk but the stem).  A 16-deep K slab then lies inside ONE (kh, kw) tap, so
// the tap and the first channel of the slab are wave-uniform and advance incrementally from slab to slab: the loader's
// per-thread integer divisions (about 200 VALU instructions per slab against its 32 MFMAs) become a few scalar ones.
template <int TM, int TN, bool C16>
__global__ __launch_bounds__(256, TM * TN == 4 ? 4 : 1) void k_conv_f32(ConvP p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int AL = BM * (CBK / 4) / 256, BL = BN * (CBK / 4) / 256;
    __shared__ float As[2][CBK][BM + 4];
    __shared__ float Bs[2][CBK][BN + 4];
    const int M = p.B * p.OH * p.OW, N = p.Cout, K = p.KH * p.KW * p.Cin;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32 * TM, wn = (wave & 1) * 32 * TN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // every thread stages the same A rows in every slab: decompose them once
    int a_base[AL], a_ih[AL], a_iw[AL];
#pragma unroll
    for (int t = 0; t < AL; ++t) {
        int row = (tid + t * 256) >> 2, gm = m0 + row;
        if (gm < M) {
            int b = gm / (p.OH * p.OW), r2 = gm % (p.OH * p.OW), oh = r2 / p.OW, ow = r2 % p.OW;
            a_base[t] = b * p.IH * p.IW; a_ih[t] = oh * p.stride - p.pad; a_iw[t] = ow * p.stride - p.pad;
        } else { a_base[t] = -1; a_ih[t] = 0; a_iw[t] = 0; }
    }
    float4 ra[AL], rb[BL];
    const int kend_ = p.ksplit > 1 ? min(K, ((int)blockIdx.z + 1) * p.kchunk) : K;
    int s_kh = 0, s_kw = 0, s_ci = 0;                         // C16: tap and first channel of the slab that fetch() loads next
    if (C16) {
        const int kb = p.ksplit > 1 ? (int)blockIdx.z * p.kchunk : 0;
        const int kk = kb / p.Cin;
        s_ci = kb % p.Cin; s_kh = kk / p.KW; s_kw = kk % p.KW;
    }
    auto fetch = [&](int k0) {                                 // slab k0 -> registers (global latency hides under MFMAs)
        if (C16) {                                             // slabs are fetched in ascending order, 16 apart
#pragma unroll
            for (int t = 0; t < AL; ++t) {
                const int kq = ((tid + t * 256) & 3) * 4;
                const int ih = a_ih[t] + s_kh, iw = a_iw[t] + s_kw;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a_base[t] >= 0 && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW)
                    v = *(const float4*)(p.in + ((size_t)(a_base[t] + ih * p.IW + iw)) * p.Cin + s_ci + kq);
                ra[t] = v;
            }
            s_ci += CBK;
            if (s_ci >= p.Cin) { s_ci = 0; if (++s_kw == p.KW) { s_kw = 0; ++s_kh; } }
        } else {
#pragma unroll
        for (int t = 0; t < AL; ++t) {
            int f = tid + t * 256, kq = (f & 3) * 4, gk = k0 + kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_base[t] >= 0 && gk < kend_) {               // Cin % 4 == 0: the 4 k's share (kh, kw)
                int kk = gk / p.Cin, ci = gk % p.Cin, kh = kk / p.KW, kw = kk % p.KW;
                int ih = a_ih[t] + kh, iw = a_iw[t] + kw;
                if (ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW)
                    v = *(const float4*)(p.in + ((size_t)(a_base[t] + ih * p.IW + iw)) * p.Cin + ci);
            }
            ra[t] = v;
        }
        }
#pragma unroll
        for (int t = 0; t < BL; ++t) {
            int f = tid + t * 256, kr = f / (BN / 4), nq = (f % (BN / 4)) * 4, gk = k0 + kr, gn = n0 + nq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((C16 || gk < kend_) && gn < N) v = *(const float4*)(p.w + (size_t)gk * N + gn);  // Cout % 4 == 0
            rb[t] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int t = 0; t < AL; ++t) {
            int f = tid + t * 256, row = f >> 2, kq = (f & 3) * 4; float4 v = ra[t];
            As[buf][kq][row] = v.x; As[buf][kq + 1][row] = v.y; As[buf][kq + 2][row] = v.z; As[buf][kq + 3][row] = v.w;
        }
#pragma unroll
        for (int t = 0; t < BL; ++t) {
            int f = tid + t * 256, kr = f / (BN / 4), nq = (f % (BN / 4)) * 4;
            *(float4*)&Bs[buf][kr][nq] = rb[t];
        }
    };
    const int kbeg = p.ksplit > 1 ? (int)blockIdx.z * p.kchunk : 0;
    const int kend = p.ksplit > 1 ? min(K, kbeg + p.kchunk) : K;
    fetch(kbeg);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += CBK) {
        const bool more = k0 + CBK < kend;
        if (more) fetch(k0 + CBK);
        {   // the operands of k-step kk + 2 are requested from LDS before the MFMAs of step kk are issued
            const int c = lane & 31, kh2 = lane >> 5;
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = As[cur][kh2][wm + 32 * i + c];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = Bs[cur][kh2][wn + 32 * j + c];
#pragma unroll
            for (int kk = 0; kk < CBK; kk += 2) {
                const int q = (kk >> 1) & 1;
                if (kk + 2 < CBK) {
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
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int gn = n0 + wn + j * 32 + (lane & 31);
            if (gn >= N) continue;
            float bi = p.bias ? p.bias[gn] : 0.f, sc = p.scale ? p.scale[gn] : 1.f, sh = p.scale ? p.shift[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int gm = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm >= M) continue;
                if (p.ksplit > 1) { p.part[((size_t)blockIdx.z * M + gm) * N + gn] = acc[i][j][r]; continue; }
                float v = acc[i][j][r] + bi;
                if (p.scale) v = fmaf(v, sc, sh);
                if (p.residual) v += p.residual[(size_t)gm * N + gn];
                if (p.relu) v = fmaxf(v, 0.f);
                p.out[(size_t)gm * N + gn] = v;
            }
        }
}

// split-K epilogue: slabs summed in slab order (deterministic), then the same epilogue as the one-pass kernel
__global__ void k_conv_reduce(ConvP p, long long MN) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= MN) return;
    const int gn = (int)(t % p.Cout);
    float v = 0.f;
    for (int z = 0; z < p.ksplit; ++z) v += p.part[(size_t)z * MN + t];
    if (p.bias) v += p.bias[gn];
    if (p.scale) v = fmaf(v, p.scale[gn], p.shift[gn]);
    if (p.residual) v += p.residual[t];
    if (p.relu) v = fmaxf(v, 0.f);
    p.out[t] = v;
}

// Tile size and split-K factor of one convolution.  Shapes whose output has too few tiles to fill the chip (the 3x3 and
// wide 1x1 convolutions of ResNet layers 3-4: 64-128 tiles for 256 CUs) run split-K when a workspace of `have` floats
// can hold the slabs; have < 0: as many as the shape wants (the size query).
static long long conv_plan(long long M, int Cout, int K, long long have, bool& big, int& ksplit, int& kchunk) {
    const long long MN = M * Cout;
    big = MN >= 128ll * 128 * 64 && Cout >= 128;
    const int bt = big ? 128 : 64;
    const long long tiles = ((Cout + bt - 1) / bt) * ((M + bt - 1) / bt);
    ksplit = 1; kchunk = 0;
    if (have != 0 && tiles < 640 && K >= 512) {
        int ks = (int)((768 + tiles - 1) / tiles);
        if (ks > 8) ks = 8;
        while (ks > 1 && (K / ks < 128 || (have > 0 && (long long)ks * MN > have))) --ks;
        if (ks > 1) {
            kchunk = ((K + ks - 1) / ks + CBK - 1) / CBK * CBK;
            ksplit = (K + kchunk - 1) / kchunk;
            return (long long)ks * MN;                        // what the choice of ks was checked against
        }
    }
    return 0;
}

/* Floats of split-K workspace eventad_conv2d_nhwc uses for this shape when it is given at least that many (0: the shape
 * runs in one pass).  Passes that may overlap in time on one GPU need a workspace each. */
extern "C" int eventad_conv2d_workspace_floats(int B, int OH, int OW, int Cin, int Cout, int KH, int KW, int64_t* floats) {
    if (!floats || B <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return EVAD_ERR_ARG;
    bool big; int ks, kc;
    const long long M = (long long)B * OH * OW;
    *floats = conv_plan(M, Cout, KH * KW * Cin, -1, big, ks, kc);
    return 0;
}

/* in f32[B,IH,IW,Cin] (NHWC, Cin % 4 == 0), w f32[KH*KW*Cin, Cout] (k = (kh*KW+kw)*Cin+ci, Cout % 4 == 0),
 * out f32[B,OH,OW,Cout]; epilogue: +bias, *scale+shift (folded BatchNorm), +residual (same shape as out), ReLU.
 * workspace (nullable) f32[workspace_floats]: split-K slabs (conv_plan), summed in slab order by a second launch. */
extern "C" int eventad_conv2d_nhwc(const float* in, const float* w, float* out, const float* bias, const float* scale,
                                   const float* shift, const float* residual, int B, int IH, int IW, int Cin, int OH,
                                   int OW, int Cout, int KH, int KW, int stride, int pad, int relu, float* workspace,
                                   int64_t workspace_floats, void* stream) {
    if (!in || !w || !out || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 3) ||
        (Cout & 3) || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || (scale && !shift) ||
        ((uintptr_t)in & 15) || ((uintptr_t)w & 15))
        return EVAD_ERR_ARG;
    ConvP p{in, w, out, bias, scale, shift, residual, B, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, relu, nullptr, 1, 0};
    const long long M = (long long)B * OH * OW, MN = M * Cout;
    const int K = KH * KW * Cin;
    bool big;
    conv_plan(M, Cout, K, (workspace && workspace_floats > 0) ? workspace_floats : 0, big, p.ksplit, p.kchunk);
    if (p.ksplit > 1) p.part = workspace;
    const int bt = big ? 128 : 64;
    dim3 grid((Cout + bt - 1) / bt, (unsigned)((M + bt - 1) / bt), p.ksplit);
    const bool c16 = Cin % CBK == 0;                         // then K and every split-K chunk are multiples of the slab depth
    if (big) { if (c16) hipLaunchKernelGGL((k_conv_f32<2, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
               else     hipLaunchKernelGGL((k_conv_f32<2, 2, false>), grid, dim3(256), 0, (hipStream_t)stream, p); }
    else     { if (c16) hipLaunchKernelGGL((k_conv_f32<1, 1, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
               else     hipLaunchKernelGGL((k_conv_f32<1, 1, false>), grid, dim3(256), 0, (hipStream_t)stream, p); }
    if (p.ksplit > 1) hipLaunchKernelGGL(k_conv_reduce, dim3(evad_blocks(MN, 256)), dim3(256), 0, (hipStream_t)stream, p, MN);
    return evad_last_error();
}

// bn1 + ReLU + 3x3/2 max-pool (pad 1) of the stem, NHWC, one thread per 4 output channels
__global__ void k_bn_relu_maxpool(const float* __restrict__ in, const float* __restrict__ scale, const float* __restrict__ shift,
                                  float* __restrict__ out, int B, int IH, int IW, int C, int OH, int OW) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int c4 = C / 4;
    if (t >= (long long)B * OH * OW * c4) return;
    int c = (int)(t % c4) * 4;
    long long m = t / c4;
    int ow = (int)(m % OW), oh = (int)((m / OW) % OH), b = (int)(m / ((long long)OW * OH));
    float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
    float4 best = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
    for (int dh = 0; dh < 3; ++dh) {
        int ih = oh * 2 - 1 + dh;
        if (ih < 0 || ih >= IH) continue;
        for (int dw = 0; dw < 3; ++dw) {
            int iw = ow * 2 - 1 + dw;
            if (iw < 0 || iw >= IW) continue;
            float4 v = *(const float4*)(in + (((size_t)b * IH + ih) * IW + iw) * C + c);
            v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
            v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
            best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
        }
    }
    *(float4*)(out + (size_t)m * C + c) = best;
}

extern "C" int eventad_bn_relu_maxpool(const float* in, const float* scale, const float* shift, float* out, int B, int IH,
                                       int IW, int C, int OH, int OW, void* stream) {
    if (!in || !scale || !shift || !out || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || (C & 3)) return EVAD_ERR_ARG;
    long long total = (long long)B * OH * OW * (C / 4);
    hipLaunchKernelGGL(k_bn_relu_maxpool, dim3(evad_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, in, scale, shift, out,
                       B, IH, IW, C, OH, OW);
    return evad_last_error();
}

// image f32[B,3,H,W] (NCHW, already /255 by format_data) -> f32[B,H,W,4] with a zero 4th channel
__global__ void k_image_to_nhwc4(const float* __restrict__ in, float* __restrict__ out, int B, int HW) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * HW) return;
    long long b = t / HW; int p = (int)(t % HW);
    const float* s = in + b * 3 * (long long)HW + p;
    *(float4*)(out + t * 4) = make_float4(s[0], s[HW], s[2 * (long long)HW], 0.f);
}

extern "C" int eventad_image_to_nhwc4(const float* in, float* out, int B, int H, int W, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0) return EVAD_ERR_ARG;
    hipLaunchKernelGGL(k_image_to_nhwc4, dim3(evad_blocks((long long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H * W);
    return evad_last_error();
}
