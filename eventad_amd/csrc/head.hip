// eventad_amd/csrc/head.hip -- the EventAD anomaly head (forward + backward) and its optimiser.
//
// Reference: EventADModel.forward (models/EventAD.py:178-333) -- for every clip of the batch, for
// every track id 1..30: GRU(64->256, 2 layers) on the ROI feature, GRU(4->32) on the box, fusion MLP,
// cross-entropy; hidden states handed from clip to clip through a soft attention over the tracks of
// the clip (:320-323).  Reference cost: B*30 Python iterations with >=3 host syncs each and M=1 GEMVs.
//
// Native formulation: 32 track slots per clip (slot 0 and 31 never valid), a device-side validity
// mask instead of Python `continue`s, [32,H] state arrays instead of dicts.  Everything that does not
// depend on the clip-to-clip chain is batched over all B*32 rows (input projections, fusion MLP, all
// weight gradients -> gemm.hip); the chain itself is 4 small launches per clip forward and 5 backward.
// The kernels here are the element-wise / reduction pieces between the GEMMs.
#include "common.h"

#define NT 32          // track slots per clip
#define FEAT_T 31      // rows of the ROI feature table per (clip, frame)  (reference: max_boxes + 1)

// v_exp_f32-based forms (relative error ~1e-7, saturate correctly at +-inf): the libm tanhf/expf expand to
// ~50 instructions each and dominated the attention kernels
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f / (__expf(2.f * x) + 1.f); }

// ------------------------------------------------------------------------------------------------
// (clip, track) lookup tables: replaces EventAD.py:229-255 (first matching box row, normalised box,
// label) and gathers the current-frame ROI features into the GEMM operand X[B*32, 64].
__global__ void k_head_tables(const float* __restrict__ bbox, const long long* __restrict__ bbox_batch, int M,
                              const float* __restrict__ labels, int label_col, const float* __restrict__ feats, int B, float Wimg,
                              float Himg, float* __restrict__ X, float* __restrict__ CO, int* __restrict__ target,
                              int* __restrict__ valid) {
    int row = blockIdx.x, b = row / NT, k = row % NT, j = threadIdx.x;      // 64 threads = 64 channels
    float f = 0.f;
    if (k >= 1 && k < FEAT_T) f = feats[(((size_t)b * 2 + 1) * FEAT_T + k) * 64 + j];
    X[(size_t)row * 64 + j] = f;
    float s = f;                                                            // torch.sum(curr_feature) == 0 ?
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    int found = -1;                                          // FIRST matching box row: 64 rows tested per step
    if (k >= 1 && k < FEAT_T)
        for (int base = 0; base < M && found < 0; base += 64) {
            const int m = base + j;
            const bool hit = m < M && (int)bbox_batch[m] == b && bbox[(size_t)m * 6 + 5] == (float)k;
            const unsigned long long mask = __ballot(hit);
            if (mask) found = base + __ffsll((long long)mask) - 1;
        }
    if (j == 0) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f; int t = 0;
        if (found >= 0) {
            const float* bb = bbox + (size_t)found * 6;
            c0 = (float)((double)bb[0] / (double)Wimg); c1 = (float)((double)bb[1] / (double)Himg);
            c2 = (float)((double)bb[2] / (double)Wimg); c3 = (float)((double)bb[3] / (double)Himg);
            t = labels ? (int)labels[found] : (label_col >= 0 ? (int)bb[label_col] : 0);
        }
        CO[(size_t)row * 4] = c0; CO[(size_t)row * 4 + 1] = c1; CO[(size_t)row * 4 + 2] = c2; CO[(size_t)row * 4 + 3] = c3;
        target[row] = t;
        valid[row] = (found >= 0 && s != 0.f) ? 1 : 0;
    }
}

extern "C" int eventad_head_tables(const float* bbox, const int64_t* bbox_batch, int M, const float* labels, int label_col,
                                   const float* feats, int B, int width, int height, float* X, float* CO,
                                   int32_t* target, int32_t* valid, void* stream) {
    if (!feats || !X || !CO || !target || !valid || B <= 0 || M < 0 || (M > 0 && (!bbox || !bbox_batch)) || label_col > 5) return EVAD_ERR_ARG;
    hipLaunchKernelGGL(k_head_tables, dim3(B * NT), dim3(64), 0, (hipStream_t)stream, bbox, (const long long*)bbox_batch, M,
                       labels, label_col, feats, B, (float)width, (float)height, X, CO, target, valid);
    return evad_last_error();
}

// ------------------------------------------------------------------------------------------------
// GRU cell (PyTorch gate order r,z,n): h' = (1-z)*n + z*h,  n = tanh(gi_n + r*gh_n); the gates (r,z,n) are kept for
// the backward pass; xdrop = h' * mask is the dropout'd input of the next GRU layer (nn.GRU(dropout=0.3),
// EventAD.py:86).  Backward: dh = dHn (+ dx_extra*mask); dn = dh*(1-z), dz = dh*(h-n), dh_prev = dSpass + dh*z;
// dgi = (da_r, da_z, da_n), dgh = (da_r, da_z, da_n*r) with da_n = dn*(1-n^2), da_z = dz*z(1-z), da_r = da_n*gh_n*r(1-r).
// SpatialAttention (EventAD.py:100-138) over the valid tracks of one clip:
//   s_k = tanh(Hn_k).w ; alpha = softmax_valid(s) ; S'_k = valid ? Hn_k*alpha_k : S_k (state kept).
// ------------------------------------------------------------------------------------------------
// Fused recurrent step: GRU cell + SpatialAttention of one clip in ONE workgroup per problem (1024 threads,
// thread = (track row k = tid>>5, lane c = tid&31) owning hidden columns c, c+32, ...).  The head is a chain
// of B dependent clip steps over 32 rows each, i.e. pure launch latency: fusing the element-wise cell with
// the row-reduction attention that consumes it (and, on the way back, the attention backward with the cell
// backward) halves the number of dependent launches; eventad_amd/head_native.py additionally runs GRU layer 1
// one clip behind layer 0 so both layers share each launch.
__device__ __forceinline__ float wave_sum(float v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64); return v; }

struct StepProb { const float* gi; const float* gh; const float* hprev; float* gates; float* hn; const float* mask;
                  float* xdrop; const float* w; float* alpha; float* snext; const int* valid; int H; };
struct StepGroup { StepProb p[3]; int n; };

template <int CH>
__device__ __forceinline__ void step_fwd_body(const StepProb& P) {
    constexpr int H = CH * 32;
    __shared__ float s_s[NT];
    __shared__ int s_v[NT];
    const int tid = threadIdx.x, k = tid >> 5, c = tid & 31;
    if (tid < NT) s_v[tid] = P.valid[tid];
    const float* a = P.gi + (size_t)k * 3 * H; const float* b = P.gh + (size_t)k * 3 * H;
    float* gt = P.gates + (size_t)k * 3 * H;
    float hn[CH], hp[CH];
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int j = c + 32 * i;
        const float r = sigmoidf_(a[j] + b[j]);
        const float z = sigmoidf_(a[H + j] + b[H + j]);
        const float n = tanhf_(a[2 * H + j] + r * b[2 * H + j]);
        hp[i] = P.hprev[(size_t)k * H + j];
        hn[i] = (1.f - z) * n + z * hp[i];
        gt[j] = r; gt[H + j] = z; gt[2 * H + j] = n;
        P.hn[(size_t)k * H + j] = hn[i];
        if (P.xdrop) P.xdrop[(size_t)k * H + j] = P.mask ? hn[i] * P.mask[(size_t)k * H + j] : hn[i];
        part += tanhf_(hn[i]) * P.w[j];
    }
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    if (c == 0) s_s[k] = part;
    __syncthreads();
    // softmax over the valid tracks: 32 lanes of wave 0 (one exp per track instead of 64 per thread), shared through LDS
    __shared__ float s_al[NT];
    if (tid < 64) {
        const int q = tid & 31;
        const bool vq = s_v[q] != 0;
        float m = vq ? s_s[q] : -3.4e38f;
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        const float e = vq ? expf(s_s[q] - m) : 0.f;
        float sum = e;
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if (tid < NT) s_al[q] = vq ? e / sum : 0.f;
    }
    __syncthreads();
    const bool v = s_v[k] != 0;
    const float al = s_al[k];
    if (c == 0) P.alpha[k] = al;
#pragma unroll
    for (int i = 0; i < CH; ++i) P.snext[(size_t)k * H + c + 32 * i] = v ? hn[i] * al : hp[i];
}

__global__ __launch_bounds__(1024) void k_step_fwd(StepGroup g) {
    const StepProb& P = g.p[blockIdx.x];
    if (P.H == 256) step_fwd_body<8>(P); else step_fwd_body<1>(P);
}

/* desc: HOST array, 12 int64 per problem: [0] gi [1] gh [2] hprev (= the attention's kept state) [3] gates out
 * [4] hn out [5] dropout mask (0 = none) [6] xdrop out (0 = none) [7] attention w [8] alpha out [9] snext out
 * [10] valid i32[32] [11] H (256 or 32). */
extern "C" int eventad_gru_att_fwd_group(int n, const int64_t* d, void* stream) {
    if (n < 1 || n > 3 || !d) return EVAD_ERR_ARG;
    StepGroup g; g.n = n;
    for (int i = 0; i < n; ++i) {
        const int64_t* q = d + 12 * i; StepProb& P = g.p[i];
        P.gi = (const float*)q[0]; P.gh = (const float*)q[1]; P.hprev = (const float*)q[2]; P.gates = (float*)q[3];
        P.hn = (float*)q[4]; P.mask = (const float*)q[5]; P.xdrop = (float*)q[6]; P.w = (const float*)q[7];
        P.alpha = (float*)q[8]; P.snext = (float*)q[9]; P.valid = (const int*)q[10]; P.H = (int)q[11];
        if (!P.gi || !P.gh || !P.hprev || !P.gates || !P.hn || !P.w || !P.alpha || !P.snext || !P.valid || (P.H != 256 && P.H != 32))
            return EVAD_ERR_ARG;
    }
    hipLaunchKernelGGL(k_step_fwd, dim3(n), dim3(1024), 0, (hipStream_t)stream, g);
    return evad_last_error();
}

struct StepBwdProb { const float* hn; const float* w; const float* alpha; const float* dsnext; const float* dhn_out;
                     const int* valid; float* dw; const float* dx_extra; const float* mask; const float* gates;
                     const float* gh; const float* hprev; float* dgi; float* dgh; float* dsprev; int H;
                     const float* ds_part; int ds_slabs; const float* dx_part; int dx_slabs; };
struct StepBwdGroup { StepBwdProb p[3]; int n; };

template <int CH>
__device__ __forceinline__ void step_bwd_body(const StepBwdProb& P) {
    constexpr int H = CH * 32;
    // s_c holds the per-row terms of d(attention w) for the column sums: the two rows of a wave are added in registers
    // first, so 16 row pairs are staged (16 KB instead of 33 KB -- small enough to sit beside two layer-1 convolution
    // workgroups on a CU when the two streams overlap)
    __shared__ float s_da[NT], s_al[NT], s_c[NT / 2][H + 1];
    __shared__ int s_v[NT];
    const int tid = threadIdx.x, k = tid >> 5, c = tid & 31;
    if (tid < NT) { s_v[tid] = P.valid[tid]; s_al[tid] = P.alpha[tid]; }
    float hn[CH], gs[CH];
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const size_t t = (size_t)k * H + c + 32 * i;
        hn[i] = P.hn[t]; gs[i] = P.dsnext[t];
        if (P.ds_part) {                                     // un-summed split-K slabs of the previous stage's dS product:
            float v = 0.f;                                   // same order as k_tile_reduce (slabs ascending, then + C)
            for (int q = 0; q < P.ds_slabs; ++q) v += P.ds_part[(size_t)q * NT * H + t];
            gs[i] = v + gs[i];
        }
        part += gs[i] * hn[i];
    }
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    __syncthreads();
    if (c == 0) s_da[k] = s_v[k] ? part : 0.f;
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < NT; ++q) if (s_v[q]) dot += s_al[q] * s_da[q];
    const bool v = s_v[k] != 0;
    const float al = s_al[k], ds = v ? al * (s_da[k] - dot) : 0.f;
    const float* gt = P.gates + (size_t)k * 3 * H;
    float* a = P.dgi + (size_t)k * 3 * H; float* cg = P.dgh + (size_t)k * 3 * H;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int j = c + 32 * i;
        const size_t t = (size_t)k * H + j;
        const float th = tanhf_(hn[i]);
        float dh = 0.f, dspass = gs[i];                      // state kept: the gradient passes through
        if (v) {
            dh = gs[i] * al + ds * P.w[j] * (1.f - th * th);
            if (P.dhn_out) dh += P.dhn_out[t];
            dspass = 0.f;
        }
        {
            const float mine = ds * th, other = __shfl_xor(mine, 32, 64);        // rows k (even, low half-wave) and k+1
            if (!(k & 1)) s_c[k >> 1][j] = mine + other;
        }
        // GRU cell backward (see k_gru_cell_bwd)
        if (P.dx_part) {
            float dx = 0.f;
            for (int q = 0; q < P.dx_slabs; ++q) dx += P.dx_part[(size_t)q * NT * H + t];
            dh += P.mask ? dx * P.mask[t] : dx;
        } else if (P.dx_extra) dh += P.mask ? P.dx_extra[t] * P.mask[t] : P.dx_extra[t];
        const float r = gt[j], z = gt[H + j], n = gt[2 * H + j];
        const float ghn = P.gh[(size_t)k * 3 * H + 2 * H + j];
        const float h = P.hprev[t];
        const float dn = dh * (1.f - z), dz = dh * (h - n);
        const float dan = dn * (1.f - n * n), daz = dz * z * (1.f - z), dar = dan * ghn * r * (1.f - r);
        a[j] = dar; a[H + j] = daz; a[2 * H + j] = dan;
        cg[j] = dar; cg[H + j] = daz; cg[2 * H + j] = dan * r;
        P.dsprev[t] = dspass + dh * z;
    }
    __syncthreads();
    if (tid < H) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) acc += s_c[q][tid];
        P.dw[tid] += acc;                                   // every problem of a launch owns its own dw accumulator
    }
}

// <= 64 VGPRs (no spills): a 16-wave workgroup then needs 256 registers per SIMD -- what ONE retiring layer-1 convolution
// workgroup frees -- instead of waiting for both convolution workgroups of a CU to drain
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_step_bwd(StepBwdGroup g) {
    const StepBwdProb& P = g.p[blockIdx.x];
    if (P.H == 256) step_bwd_body<8>(P); else step_bwd_body<1>(P);
}

/* desc: HOST array, 20 int64 per problem: [0] hn [1] attention w [2] alpha [3] dsnext [4] dhn_out (0 = none)
 * [5] valid [6] dw accumulator [7] dx_extra (0 = none) [8] mask (0 = none) [9] gates [10] gh [11] hprev
 * [12] dgi out [13] dgh out [14] dsprev out [15] H [16] dsnext slabs (0 = none) [17] their number [18] dx_extra slabs
 * (0 = none; replace [7]) [19] their number.  Slabs are f32[n][32][H], the un-summed split-K output of eventad_gemm_group. */
extern "C" int eventad_gru_att_bwd_group(int n, const int64_t* d, void* stream) {
    if (n < 1 || n > 3 || !d) return EVAD_ERR_ARG;
    StepBwdGroup g; g.n = n;
    for (int i = 0; i < n; ++i) {
        const int64_t* q = d + 20 * i; StepBwdProb& P = g.p[i];
        P.hn = (const float*)q[0]; P.w = (const float*)q[1]; P.alpha = (const float*)q[2]; P.dsnext = (const float*)q[3];
        P.dhn_out = (const float*)q[4]; P.valid = (const int*)q[5]; P.dw = (float*)q[6]; P.dx_extra = (const float*)q[7];
        P.mask = (const float*)q[8]; P.gates = (const float*)q[9]; P.gh = (const float*)q[10]; P.hprev = (const float*)q[11];
        P.dgi = (float*)q[12]; P.dgh = (float*)q[13]; P.dsprev = (float*)q[14]; P.H = (int)q[15];
        P.ds_part = (const float*)q[16]; P.ds_slabs = (int)q[17]; P.dx_part = (const float*)q[18]; P.dx_slabs = (int)q[19];
        if ((P.ds_part && (P.ds_slabs < 1 || P.ds_slabs > 8)) || (P.dx_part && (P.dx_slabs < 1 || P.dx_slabs > 8))) return EVAD_ERR_ARG;
        if (!P.hn || !P.w || !P.alpha || !P.dsnext || !P.valid || !P.dw || !P.gates || !P.gh || !P.hprev || !P.dgi || !P.dgh ||
            !P.dsprev || (P.H != 256 && P.H != 32))
            return EVAD_ERR_ARG;
    }
    hipLaunchKernelGGL(k_step_bwd, dim3(n), dim3(1024), 0, (hipStream_t)stream, g);
    return evad_last_error();
}

// ------------------------------------------------------------------------------------------------
// Output layer + cross-entropy (Fuse_GRUNet.fusion[3], EventAD.py:27,280-281): one wave per row.
// F1r = ReLU(fusion.0(...)) [R,256]; D2 = dropout mask (nullable).  Writes logits, d(loss)/d(logits)
// (masked by validity) and the per-row loss; k_loss_sum adds the rows in (clip, track) order, which
// is the order of the reference's `losses['cross_entropy'] += loss`.
__global__ void k_fuse_out(const float* __restrict__ F1r, const float* __restrict__ D2, const float* __restrict__ Wo,
                           const float* __restrict__ bo, const int* __restrict__ target, const int* __restrict__ valid,
                           int R, float* __restrict__ logits, float* __restrict__ dlog, float* __restrict__ loss_rows,
                           float* __restrict__ dPre) {
    int row = blockIdx.x, lane = threadIdx.x;
    if (row >= R) return;
    float a0 = 0.f, a1 = 0.f;
    float fr[4], dm[4], w0[4], w1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        fr[i] = F1r[(size_t)row * 256 + j];
        dm[i] = D2 ? D2[(size_t)row * 256 + j] : 1.f;
        w0[i] = Wo[j]; w1[i] = Wo[256 + j];
        const float f = D2 ? fr[i] * dm[i] : fr[i];
        a0 += f * w0[i]; a1 += f * w1[i];
    }
    for (int d = 32; d > 0; d >>= 1) { a0 += __shfl_down(a0, d, 64); a1 += __shfl_down(a1, d, 64); }
    float g0 = 0.f, g1 = 0.f;
    if (lane == 0) {
        float l0 = a0 + bo[0], l1 = a1 + bo[1];
        logits[2 * row] = l0; logits[2 * row + 1] = l1;
        float m = fmaxf(l0, l1), e0 = expf(l0 - m), e1 = expf(l1 - m), s = e0 + e1;
        int t = target[row], v = valid[row];
        float lse = logf(s) + m;
        loss_rows[row] = v ? lse - (t ? l1 : l0) : 0.f;
        g0 = v ? e0 / s - (t == 0 ? 1.f : 0.f) : 0.f;
        g1 = v ? e1 / s - (t == 1 ? 1.f : 0.f) : 0.f;
        dlog[2 * row] = g0; dlog[2 * row + 1] = g1;
    }
    if (dPre) {      // backward of the output layer into the hidden layer, while the row is in registers (= k_out_bwd_pre)
        g0 = __shfl(g0, 0, 64); g1 = __shfl(g1, 0, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dPre[(size_t)row * 256 + lane + 64 * i] = fr[i] > 0.f ? (g0 * w0[i] + g1 * w1[i]) * dm[i] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_loss_sum(const float* __restrict__ loss_rows, int R, float* __restrict__ loss) {
    __shared__ float s_l[1024];                              // rows are fetched in parallel, added in row order by one thread
    float s = 0.f;
    for (int base = 0; base < R; base += 1024) {
        for (int i = threadIdx.x; i < 1024 && base + i < R; i += 256) s_l[i] = loss_rows[base + i];
        __syncthreads();
        if (threadIdx.x == 0) { const int m = min(1024, R - base); for (int r = 0; r < m; ++r) s += s_l[r]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = s;
}

extern "C" int eventad_fuse_out(const float* F1r, const float* D2, const float* Wo, const float* bo, const int32_t* target,
                                const int32_t* valid, int R, float* logits, float* dlog, float* loss_rows, float* loss,
                                float* dPre, void* stream) {
    if (!F1r || !Wo || !bo || !target || !valid || !logits || !dlog || !loss_rows || R <= 0) return EVAD_ERR_ARG;
    hipLaunchKernelGGL(k_fuse_out, dim3(R), dim3(64), 0, (hipStream_t)stream, F1r, D2, Wo, bo, target, valid, R, logits, dlog, loss_rows, dPre);
    if (loss) hipLaunchKernelGGL(k_loss_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_rows, R, loss);   // else: eventad_out_bwd adds the rows
    return evad_last_error();
}

// backward of the output layer: dPre[R,256] = (dlog.Wo) * D2 * [F1r > 0]  (one thread per element),
// dWo[c][j] = sum_r dlog[r][c] * F1[r][j], dbo[c] = sum_r dlog[r][c]  (one wave per column j, rows in
// lane-strided order then a fixed shuffle tree: reproducible).
__global__ void k_out_bwd_pre(const float* __restrict__ dlog, const float* __restrict__ Wo, const float* __restrict__ F1r,
                              const float* __restrict__ D2, int R, float* __restrict__ dPre) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * 256) return;
    int r = t >> 8, j = t & 255;
    float m = D2 ? D2[t] : 1.f;
    dPre[t] = F1r[t] > 0.f ? (dlog[2 * r] * Wo[j] + dlog[2 * r + 1] * Wo[256 + j]) * m : 0.f;
}

__global__ __launch_bounds__(256) void k_out_bwd_w(const float* __restrict__ dlog, const float* __restrict__ F1r,
                                                   const float* __restrict__ D2, int R, float* __restrict__ dWo,
                                                   float* __restrict__ dbo, float* __restrict__ clear, int n_clear,
                                                   const float* __restrict__ loss_rows, float* __restrict__ loss) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_clear; i += 64 * 256) clear[i] = 0.f;
    if (loss && blockIdx.x == 63) {                          // the forward's loss rows, added in row order as k_loss_sum does
        __shared__ float s_l[1024];
        float s = 0.f;
        for (int base = 0; base < R; base += 1024) {
            for (int i = threadIdx.x; i < 1024 && base + i < R; i += 256) s_l[i] = loss_rows[base + i];
            __syncthreads();
            if (threadIdx.x == 0) { const int m = min(1024, R - base); for (int r = 0; r < m; ++r) s += s_l[r]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) *loss = s;
    }
    int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;       // 64 blocks x 4 waves = 256 columns
    float g0 = 0.f, g1 = 0.f, b0 = 0.f, b1 = 0.f;
    for (int r = lane; r < R; r += 64) {
        float d0 = dlog[2 * r], d1 = dlog[2 * r + 1];
        float f = F1r[(size_t)r * 256 + j] * (D2 ? D2[(size_t)r * 256 + j] : 1.f);
        g0 += d0 * f; g1 += d1 * f; b0 += d0; b1 += d1;
    }
    g0 = wave_sum(g0); g1 = wave_sum(g1); b0 = wave_sum(b0); b1 = wave_sum(b1);
    if (lane == 0) { dWo[j] = g0; dWo[256 + j] = g1; if (j == 0) { dbo[0] = b0; dbo[1] = b1; } }
}

extern "C" int eventad_out_bwd(const float* dlog, const float* Wo, const float* F1r, const float* D2, int R, float* dPre,
                               float* dWo, float* dbo, float* clear, int n_clear, const float* loss_rows, float* loss,
                               void* stream) {
    if (!dlog || !Wo || !F1r || !dWo || !dbo || R <= 0 || n_clear < 0 || (n_clear > 0 && !clear) || (loss && !loss_rows)) return EVAD_ERR_ARG;
    if (dPre)
        hipLaunchKernelGGL(k_out_bwd_pre, dim3(evad_blocks((long long)R * 256, 256)), dim3(256), 0, (hipStream_t)stream, dlog, Wo,
                           F1r, D2, R, dPre);
    hipLaunchKernelGGL(k_out_bwd_w, dim3(64), dim3(256), 0, (hipStream_t)stream, dlog, F1r, D2, R, dWo, dbo, clear, n_clear, loss_rows, loss);
    return evad_last_error();
}

// column sums (bias gradients): out[n] = sum_r A[r, n]; 64 columns per workgroup, the 4 waves split the
// rows, fixed-order LDS reduction.
#define COL_MAXP 12
struct ColGroup { const float* A[COL_MAXP]; float* out[COL_MAXP]; int lda[COL_MAXP], N[COL_MAXP], R[COL_MAXP]; int n; };
__global__ __launch_bounds__(256) void k_colsum_g(ColGroup g) {
    __shared__ float red[4][64];
    const int pi = blockIdx.y, N = g.N[pi], R = g.R[pi];
    int n = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    if (blockIdx.x * 64 >= N) return;
    float s = 0.f;
    if (n < N) {
        const float* a = g.A[pi] + n; const size_t ld = (size_t)g.lda[pi];
        int r = w;
        for (; r + 28 < R; r += 32) {                        // 8 rows in flight, added in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a[(size_t)(r + 4 * u) * ld];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < R; r += 4) s += a[(size_t)r * ld];
    }
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && n < N) g.out[pi][n] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

/* desc: HOST array, 5 int64 per problem: [0] A [1] out [2] lda [3] N [4] rows (0 = R, the call's common row count). */
extern "C" int eventad_colsum_group(int n, const int64_t* desc, int R, void* stream) {
    if (n < 1 || n > COL_MAXP || !desc || R <= 0) return EVAD_ERR_ARG;
    ColGroup g; g.n = n; int nmax = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t* d = desc + 5 * i;
        g.A[i] = (const float*)d[0]; g.out[i] = (float*)d[1]; g.lda[i] = (int)d[2]; g.N[i] = (int)d[3]; g.R[i] = d[4] > 0 ? (int)d[4] : R;
        if (!g.A[i] || !g.out[i] || g.N[i] <= 0 || g.lda[i] < g.N[i]) return EVAD_ERR_ARG;
        if (g.N[i] > nmax) nmax = g.N[i];
    }
    hipLaunchKernelGGL(k_colsum_g, dim3(evad_blocks(nmax, 64), n), dim3(256), 0, (hipStream_t)stream, g);
    return evad_last_error();
}


// ------------------------------------------------------------------------------------------------
// Optimiser: clip_grad_norm_(max_norm) + AdamW on flat fp32 buffers (utils/train.py:27-53,141-152).
//   state[0]=step  [1]=grad norm  [2]=clip coefficient  [3]=1-beta1^t  [4]=1-beta2^t  [5]=non-finite count  [6]=this update was skipped
//   [7]=step count of the update in flight ([3], [4] belong to it)
#define NORM_BLOCKS 256
__global__ __launch_bounds__(256) void k_sqnorm_partial(const float* __restrict__ g, int n, float* __restrict__ partial,
                                                        double beta1, double beta2, float* __restrict__ state) {
    __shared__ float red[256];
    float s = 0.f;
    // 16-byte loads (g is 16-byte aligned: eventad_clip_adamw checks); a thread adds its elements in ascending order
    const int n4 = n >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += NORM_BLOCKS * 256) {
        const float4 v = ((const float4*)g)[i];
        s += v.x * v.x; s += v.y * v.y; s += v.z * v.z; s += v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[4 * n4 + threadIdx.x]; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {               // what the update needs if it is not skipped: the next step count
        const float step = state[0] + 1.f;                   // ([7], committed to [0] by k_adamw) and its bias corrections
        state[7] = step;
        state[3] = (float)(1.0 - pow(beta1, (double)step)); state[4] = (float)(1.0 - pow(beta2, (double)step));
    }
}

// Every workgroup adds up the 256 partial sums itself (4 per lane of its first wave, then a fixed shuffle tree: the same
// bits in every workgroup and every run) instead of waiting for a one-wave launch that does it once.
//   The reference raises on a non-finite gradient BEFORE optimizer.step() (utils/train.py:144-149), so its weights stay
//   clean; and a batch without any valid object leaves every p.grad None, which AdamW skips entirely.  Both cases become a
//   skipped update here: parameters, moments and the step count stay as they are ([6] = 1), the non-finite case is also
//   counted in [5] for the host's lazy check.
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n,
                                               float lr, float beta1, float omb1, float beta2, float omb2, float eps, float decay,
                                               float max_norm, const float* __restrict__ partial, float* __restrict__ state,
                                               const float* __restrict__ loss_in, float* __restrict__ result_out) {
    __shared__ float s_norm;
    // A thread's elements (two groups of four consecutive ones, a grid apart: 16-byte loads and stores; the update is pure
    // streaming, 7 floats of traffic per parameter, and runs on the head's CU share) are requested BEFORE the norm is formed
    // from the partial sums, so that round trip and the barrier hide under the loads.
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
    const int i0 = 4 * tid, i1 = 4 * (tid + nthr);
    const bool full0 = i0 + 3 < n, full1 = i1 + 3 < n;
    float4 p4[2], g4[2], m4[2], v4[2];
    if (full0) { p4[0] = *(const float4*)(p + i0); g4[0] = *(const float4*)(g + i0); m4[0] = *(const float4*)(m + i0); v4[0] = *(const float4*)(v + i0); }
    if (full1) { p4[1] = *(const float4*)(p + i1); g4[1] = *(const float4*)(g + i1); m4[1] = *(const float4*)(m + i1); v4[1] = *(const float4*)(v + i1); }
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        float s = (partial[lane] + partial[64 + lane]) + (partial[128 + lane] + partial[192 + lane]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) s_norm = sqrtf(s);
    }
    __syncthreads();
    const float norm = s_norm;
    const bool finite = isfinite(norm), skip = !finite || norm == 0.f;
    float coef = max_norm / (norm + 1e-6f);                 // torch.nn.utils.clip_grad_norm_
    coef = coef < 1.f ? coef : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state[1] = norm;
        if (!finite) state[5] += 1.f;
        state[6] = skip ? 1.f : 0.f; state[2] = skip ? 1.f : coef;
        if (!skip) state[0] = state[7];
        if (result_out) { result_out[0] = loss_in ? loss_in[0] : 0.f; result_out[1] = norm; }
    }
    if (skip) return;
    const float bc1 = state[3], bc2s = sqrtf(state[4]);
    auto one = [&](float& pw, float& gw, float& mw, float& vw) {
        const float gi = gw * coef;
        gw = gi;                                            // grads are left clipped, as clip_grad_norm_ does
        const float pi = pw * decay;
        const float mi = beta1 * mw + omb1 * gi;
        const float vi = beta2 * vw + omb2 * gi * gi;
        mw = mi; vw = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        pw = pi - (lr / bc1) * (mi / denom);
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = h ? i1 : i0;
        if (h ? full1 : full0) {
            one(p4[h].x, g4[h].x, m4[h].x, v4[h].x); one(p4[h].y, g4[h].y, m4[h].y, v4[h].y);
            one(p4[h].z, g4[h].z, m4[h].z, v4[h].z); one(p4[h].w, g4[h].w, m4[h].w, v4[h].w);
            *(float4*)(g + i) = g4[h]; *(float4*)(m + i) = m4[h]; *(float4*)(v + i) = v4[h]; *(float4*)(p + i) = p4[h];
        } else {
            for (int k = i; k < min(i + 4, n); ++k) one(p[k], g[k], m[k], v[k]);      // (the last, partial group of four)
        }
    }
}

/* eventad_clip_adamw that also leaves the step's (loss, gradient norm) in result_out f32[2] (loss read from loss_in f32[1];
 * written by the update's own launch: a caller that keeps several steps' results needs no copy launch behind it). */
extern "C" int eventad_clip_adamw_out(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int n, double lr,
                                      double beta1, double beta2, double eps, double weight_decay, double max_norm,
                                      float* partial, float* state, const float* loss_in, float* result_out, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !partial || !state || n <= 0) return EVAD_ERR_ARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return EVAD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sqnorm_partial, dim3(NORM_BLOCKS), dim3(256), 0, st, grads, n, partial, beta1, beta2, state);
    hipLaunchKernelGGL(k_adamw, dim3(evad_blocks(((long long)n + 7) / 8, 256)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, (float)lr,
                       (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)(1.0 - lr * weight_decay), (float)max_norm, partial, state, loss_in, result_out);
    return evad_last_error();
}

extern "C" int eventad_clip_adamw(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int n, double lr,
                                  double beta1, double beta2, double eps, double weight_decay, double max_norm,
                                  float* partial, float* state, void* stream) {
    return eventad_clip_adamw_out(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, max_norm, partial,
                                  state, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// Dropout masks of a training step (nn.GRU(dropout=.3) between the GRU layers, nn.Dropout(.3) in the fusion MLP:
// models/EventAD.py:26,86) from a counter-based generator: mask[i] = keep ? 1/p : 0 with the random word
// Philox4x32-10(key = seed, counter = (i / 4, step)).  A step's masks depend on (seed, step, i) only -- not on a generator
// state that launches on other streams advance -- so every launch form of the step (eager, one hipGraph, several backbone
// graphs in flight, each reading its step number from its own device word) draws the same masks for the same step.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}

__global__ void k_dropout_masks(float* __restrict__ out, long long n, float keep, float scale, uint64_t seed, int64_t step,
                                const int64_t* __restrict__ step_dev) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // 4 values per thread
    if (q * 4 >= n) return;
    const uint64_t s = (uint64_t)(step_dev ? *step_dev : step);
    uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)s, (uint32_t)(s >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ((c[j] >> 8) * (1.0f / 16777216.0f)) < keep ? scale : 0.f;   // 24-bit uniform in [0,1)
    if (q * 4 + 3 < n) *(float4*)(out + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
    else for (int j = 0; q * 4 + j < n; ++j) out[q * 4 + j] = v[j];
}

extern "C" int eventad_dropout_masks(float* out, int64_t n, double keep_prob, int64_t seed, int64_t step,
                                     const int64_t* step_dev, void* stream) {
    if (!out || n <= 0 || !(keep_prob > 0.0 && keep_prob <= 1.0) || ((uintptr_t)out & 15)) return EVAD_ERR_ARG;
    hipLaunchKernelGGL(k_dropout_masks, dim3(evad_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, (long long)n,
                       (float)keep_prob, (float)(1.0 / keep_prob), (uint64_t)seed, step, step_dev);
    return evad_last_error();
}

// ------------------------------------------------------------------------------------------------
// Streams restricted to a set of CUs (default priority): the training step can give the head's chain of small launches a
// few CUs of its own, so that its workgroups do not queue behind the chip-wide grids of the event-level kernels.
// mask: bit i set = CU i may be used; words = number of 32-bit words.
extern "C" int eventad_stream_create_cu_mask(const uint32_t* mask, int words, void** stream_out) {
    if (!mask || words < 1 || !stream_out) return EVAD_ERR_ARG;
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) { (void)hipGetLastError(); return -2; }
    *stream_out = (void*)s;
    return EVAD_OK;
}
extern "C" int eventad_stream_destroy(void* stream) {
    if (!stream) return EVAD_ERR_ARG;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? EVAD_OK : -2;
}
