// eventad_amd/csrc/headstage.hip -- one launch per clip stage of the EventAD head's forward recurrence.
//
// Reference: the per-clip, per-track loop of EventADModel.forward (models/EventAD.py:211-323): GRU(64->256, 2 layers)
// and GRU(4->32) cells on the tracks of a clip, then SpatialAttention over the clip's valid tracks, whose output is the
// hidden state handed to the next clip (:320-323).
//
// The chain of B clips is pure dependent latency.  As separate launches a stage is a grouped GEMM (state . Whh^T, many
// workgroups) followed by a one-workgroup cell + attention kernel (the attention is a reduction over whole rows), i.e.
// two dependent launches per clip.  Here a stage is ONE launch of 32 workgroups: workgroup g owns hidden columns
// 8g .. 8g+7 of the two 256-wide GRU layers and column g of the 32-wide coordinate GRU, and
//   A. recomputes the attention of the PREVIOUS clip from that clip's cell outputs (every workgroup does the whole
//      32 x 256 reduction redundantly: 8 tanh per thread) and so obtains the state entering this clip, as the A operand
//      of its products, in LDS;
//   B. multiplies it with its 24 gate rows of Whh (fp32 MFMA 32x32x2, K split over 4 waves, partial sums added in wave
//      order: reproducible) -- for layer 1 also the input product with the dropout'd layer-0 output;
//   C. applies the GRU cell to its columns and writes gates, new hidden columns and the state columns it owns.
// The wavefront of head_native.py is kept: a stage runs layer 0 and the coordinate GRU on clip b and layer 1 on clip b-1.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NT 32                   // track slots per clip
#define HS_LD 260               // LDS row stride (floats) of a 32 x 256 operand tile: 16-byte aligned rows
#define HS_WGS 32               // workgroups per stage = 256 / 8 hidden columns = the 32 columns of the coordinate GRU

// same forms as head.hip (v_exp_f32 based; saturate correctly)
__device__ __forceinline__ float hs_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float hs_tanh(float x) { return 1.f - 2.f / (__expf(2.f * x) + 1.f); }

static int hs_allow_lds(const void* kernel, int bytes, int) { return evad_allow_lds(kernel, (size_t)bytes); }

struct HsSlot {
    const float* hn_prev; const float* s_prev; const int* valid_prev; const float* w_att; float* alpha_prev; float* s_cur;
    const float* whh; const float* bhh; const float* gi; const float* xin; const float* wih; const float* bih;
    float* gh_out; float* gates; float* hn; const float* mask; float* xdrop; float* ghr_out; int present;
};
struct HsGroup { HsSlot s[3]; };     // [0] 256-wide, input product precomputed; [1] 256-wide, input product xin . Wih^T here; [2] 32-wide

#define HS_SMEM_FLOATS (3 * NT * HS_LD + NT * 36 + 9 * NT + 13 * NT * 24)

__global__ __launch_bounds__(1024) void k_head_stage_fwd(HsGroup G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*sA)[NT][HS_LD] = (float (*)[NT][HS_LD])smem;                  // [0] state of slot 0, [1] state of slot 1, [2] xin of slot 1
    float (*sC)[36] = (float (*)[36])(smem + 3 * NT * HS_LD);             // state of slot 2 (32 x 32)
    float* s_part = smem + 3 * NT * HS_LD + NT * 36;                      // [3][32] attention scores
    float* s_al = s_part + 3 * NT;                                        // [3][32] attention weights
    int* s_v = (int*)(s_al + 3 * NT);                                     // [3][32] validity of the previous clip's tracks
    float (*Red)[NT][24] = (float (*)[NT][24])(s_al + 6 * NT);            // [13][32][24] partial products of the 13 MFMA waves
    const int tid = threadIdx.x, k = tid >> 5, c = tid & 31, lane = tid & 63, g = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // (a scalar: what is selected by it is read with scalar loads)
    const HsSlot& S0 = G.s[0]; const HsSlot& S1 = G.s[1]; const HsSlot& S2 = G.s[2];
    const bool a0 = S0.present && S0.hn_prev, a1 = S1.present && S1.hn_prev, a2 = S2.present && S2.hn_prev;

    // ---- every global operand of the stage is requested up front (one round trip instead of one per phase): the previous
    //      clip's cell outputs, the states kept for invalid tracks, layer 1's input rows, and each MFMA wave's weight rows
    const int nn = lane & 31, h = lane >> 5;
    float4 bw[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bw[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool mf_live = wave < 4 ? S0.present != 0 : (wave < 12 ? S1.present != 0 : (wave == 12 && S2.present != 0));
    if (mf_live && wave < 12 && nn < 24) {
        const float* Wm = wave < 4 ? S0.whh : (wave < 8 ? S1.whh : S1.wih);
        const int n = (nn >> 3) * 256 + 8 * g + (nn & 7);                // gate row (r | z | n blocks of 256) of lane nn < 24
#pragma unroll
        for (int t = 0; t < 8; ++t) bw[t] = *(const float4*)(Wm + (size_t)n * 256 + (wave & 3) * 64 + 8 * t + 4 * h);
    } else if (mf_live && wave == 12 && nn < 3) {                          // coordinate GRU: K = 32, gate rows g, 32 + g, 64 + g
#pragma unroll
        for (int t = 0; t < 4; ++t) bw[t] = *(const float4*)(S2.whh + (size_t)(nn * 32 + g) * 32 + 8 * t + 4 * h);
    }
    float hn0[8], hn1[8], kp0[8], kp1[8], xin[8], hnc = 0.f, kpc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { hn0[i] = 0.f; hn1[i] = 0.f; kp0[i] = 0.f; kp1[i] = 0.f; xin[i] = 0.f; }
    if (S0.present) {
        const float* keep = a0 ? S0.s_prev : S0.s_cur;       // no previous clip: the initial state, stored in this clip's rows
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int j = c + 32 * i; if (a0) hn0[i] = S0.hn_prev[k * 256 + j]; kp0[i] = keep[k * 256 + j]; }
    }
    if (S1.present) {
        const float* keep = a1 ? S1.s_prev : S1.s_cur;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = c + 32 * i;
            if (a1) hn1[i] = S1.hn_prev[k * 256 + j];
            kp1[i] = keep[k * 256 + j]; xin[i] = S1.xin[k * 256 + j];
        }
    }
    if (S2.present) { if (a2) hnc = S2.hn_prev[k * 32 + c]; kpc = (a2 ? S2.s_prev : S2.s_cur)[k * 32 + c]; }
    // cell operands of phase C
    // (the slot of a thread of the cell phase is the same for its whole wave: as a scalar, the slot's descriptor is read with
    // scalar loads from the kernel arguments; indexed per lane it becomes a vector load per field and a round trip of its own
    // in front of every operand)
    const int c_slot = min(__builtin_amdgcn_readfirstlane(tid >> 8), 2), c_kk = (tid & 255) >> 3, c_j = 8 * g + (tid & 7);
    float c_gi[3] = {0.f, 0.f, 0.f}, c_bh[3] = {0.f, 0.f, 0.f}, c_mask = 1.f;
    if (tid < 512 && G.s[c_slot].present) {
        const HsSlot& S = G.s[c_slot];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            c_bh[q] = S.bhh[q * 256 + c_j];
            c_gi[q] = c_slot == 0 ? S.gi[c_kk * 768 + q * 256 + c_j] : S.bih[q * 256 + c_j];
        }
        if (S.xdrop && S.mask) c_mask = S.mask[c_kk * 256 + c_j];
    } else if (tid >= 512 && tid < 512 + NT && S2.present) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { c_bh[q] = S2.bhh[q * 32 + g]; c_gi[q] = S2.gi[(tid - 512) * 96 + q * 32 + g]; }
    }

    // ---- A. attention of the previous clip (SpatialAttention, EventAD.py:100-138): s_k = tanh(Hn_k).w, alpha = softmax
    //         over the valid tracks, state_k = valid ? Hn_k * alpha_k : state kept
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (a0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p0 += hs_tanh(hn0[i]) * S0.w_att[c + 32 * i];
    }
    if (a1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p1 += hs_tanh(hn1[i]) * S1.w_att[c + 32 * i];
    }
    if (a2) p2 = hs_tanh(hnc) * S2.w_att[c];
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) { p0 += __shfl_xor(p0, d, 64); p1 += __shfl_xor(p1, d, 64); p2 += __shfl_xor(p2, d, 64); }
    if (c == 0) { s_part[k] = p0; s_part[NT + k] = p1; s_part[2 * NT + k] = p2; }
    if (wave < 3 && lane < NT) {                             // one wave per slot: the slot's descriptor is read with scalar loads
        const HsSlot& S = G.s[wave];
        s_v[wave * NT + lane] = (S.present && S.hn_prev) ? S.valid_prev[lane] : 0;
    }
    __syncthreads();
    if (wave < 3) {                                          // one wave per slot; both half-waves compute the same 32 values
        const int q = wave * NT + (lane & 31);
        const bool vq = s_v[q] != 0;
        float m = vq ? s_part[q] : -3.4e38f;
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        const float e = vq ? expf(s_part[q] - m) : 0.f;
        float sum = e;
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if (lane < 32) s_al[q] = vq ? e / sum : 0.f;
    }
    __syncthreads();
    if (S0.present) {
        const bool v = a0 && s_v[k] != 0; const float al = s_al[k];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = c + 32 * i;
            const float val = v ? hn0[i] * al : kp0[i];
            sA[0][k][j] = val;
            if (a0 && (j >> 3) == g) S0.s_cur[k * 256 + j] = val;
        }
        if (a0 && g == 0 && c == 0) S0.alpha_prev[k] = al;
    }
    if (S1.present) {
        const bool v = a1 && s_v[NT + k] != 0; const float al = s_al[NT + k];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = c + 32 * i;
            const float val = v ? hn1[i] * al : kp1[i];
            sA[1][k][j] = val;
            if (a1 && (j >> 3) == g) S1.s_cur[k * 256 + j] = val;
            sA[2][k][j] = xin[i];
        }
        if (a1 && g == 0 && c == 0) S1.alpha_prev[k] = al;
    }
    if (S2.present) {
        const bool v = a2 && s_v[2 * NT + k] != 0; const float al = s_al[2 * NT + k];
        const float val = v ? hnc * al : kpc;
        sC[k][c] = val;
        if (a2 && c == g) S2.s_cur[k * 32 + c] = val;
        if (a2 && g == 0 && c == 0) S2.alpha_prev[k] = al;
    }
    __syncthreads();

    // ---- B. products with this workgroup's gate rows.  MFMA 32x32x2: lane l supplies A[l & 31][k'] and B[k'][l & 31] for
    //         k' = l >> 5; a lane loads 4 consecutive k as one float4 (k = 8t + 4h + i for half-wave h), so step (t, i) pairs
    //         k = 8t + i with k = 8t + 4 + i -- any pairing that A and B share is a valid order of the sum over k.
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mf_live && wave < 12) {
        const float (*At)[HS_LD] = wave < 4 ? sA[0] : (wave < 8 ? sA[1] : sA[2]);
        const int kc = (wave & 3) * 64;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 a = *(const float4*)&At[nn][kc + 8 * t + 4 * h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[t].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[t].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[t].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[t].w, acc, 0, 0, 0);
        }
    } else if (mf_live && wave == 12) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 a = *(const float4*)&sC[nn][8 * t + 4 * h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[t].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[t].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[t].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[t].w, acc, 0, 0, 0);
        }
    }
    // C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (wave < 13 && nn < 24) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][nn] = acc[r];
    }
    __syncthreads();

    // ---- C. GRU cell (PyTorch gate order r, z, n) on this workgroup's columns
    if (tid < 512) {
        const int slot = c_slot, kk = c_kk, jj = tid & 7, j = c_j;
        const HsSlot& S = G.s[slot];
        if (S.present) {
            float gh[3], gi[3], raw[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int w0 = 4 * slot, col = q * 8 + jj;
                raw[q] = ((Red[w0][kk][col] + Red[w0 + 1][kk][col]) + Red[w0 + 2][kk][col]) + Red[w0 + 3][kk][col];
                gh[q] = raw[q] + c_bh[q];
                if (slot == 0) gi[q] = c_gi[q];
                else gi[q] = (((Red[8][kk][col] + Red[9][kk][col]) + Red[10][kk][col]) + Red[11][kk][col]) + c_gi[q];
            }
            const float r = hs_sigmoid(gi[0] + gh[0]), z = hs_sigmoid(gi[1] + gh[1]), n = hs_tanh(gi[2] + r * gh[2]);
            const float hp = sA[slot][kk][j];
            const float hn = (1.f - z) * n + z * hp;
            float* gt = S.gates + kk * 768; float* go = S.gh_out + kk * 768;
            gt[j] = r; gt[256 + j] = z; gt[512 + j] = n;
            go[j] = gh[0]; go[256 + j] = gh[1]; go[512 + j] = gh[2];
            if (S.ghr_out) { float* gr = S.ghr_out + kk * 768; gr[j] = raw[0]; gr[256 + j] = raw[1]; gr[512 + j] = raw[2]; }
            S.hn[kk * 256 + j] = hn;
            if (S.xdrop) S.xdrop[kk * 256 + j] = S.mask ? hn * c_mask : hn;
        }
    } else if (tid < 512 + NT && S2.present) {
        const int kk = tid - 512;
        float gh[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) gh[q] = Red[12][kk][q] + c_bh[q];
        const float r = hs_sigmoid(c_gi[0] + gh[0]), z = hs_sigmoid(c_gi[1] + gh[1]), n = hs_tanh(c_gi[2] + r * gh[2]);
        const float hp = sC[kk][g];
        const float hn = (1.f - z) * n + z * hp;
        float* gt = S2.gates + kk * 96; float* go = S2.gh_out + kk * 96;
        gt[g] = r; gt[32 + g] = z; gt[64 + g] = n;
        go[g] = gh[0]; go[32 + g] = gh[1]; go[64 + g] = gh[2];
        if (S2.ghr_out) { float* gr = S2.ghr_out + kk * 96; gr[g] = Red[12][kk][0]; gr[32 + g] = Red[12][kk][1]; gr[64 + g] = Red[12][kk][2]; }
        S2.hn[kk * 32 + g] = hn;
    }
}

/* desc: HOST array, 3 slots x 19 int64: slot 0 = 256-wide GRU layer whose input product gi (incl. bias) is precomputed,
 * slot 1 = 256-wide layer whose input product xin . wih^T + bih is computed here, slot 2 = 32-wide layer with precomputed gi.
 * Per slot: [0] present (0 = skip the slot) [1] hn_prev f32[32,H] cell output of the previous clip (0 = there is none: the
 * state is read from [6] as it is) [2] s_prev f32[32,H] state that entered the previous clip (kept for its invalid
 * tracks) [3] valid_prev i32[32] [4] attention w f32[H] [5] alpha_prev out f32[32] [6] s_cur f32[32,H]: state entering
 * this clip (out, unless [1] is 0) [7] whh f32[3H,H] [8] bhh [9] gi f32[32,3H] [10] xin f32[32,H] [11] wih f32[3H,H]
 * [12] bih [13] gh out f32[32,3H] (state . whh^T + bhh) [14] gates out f32[32,3H] (r|z|n) [15] hn out f32[32,H]
 * [16] dropout mask (0 = none) [17] xdrop out = hn * mask (0 = none) [18] ghr out f32[32,3H] = state . whh^T without the
 * bias (0 = none; read by eventad_head_stage_bwd). */
extern "C" int eventad_head_stage_fwd(const int64_t* d, void* stream) {
    if (!d) return EVAD_ERR_ARG;
    HsGroup G;
    bool any = false;
    for (int i = 0; i < 3; ++i) {
        const int64_t* q = d + 19 * i; HsSlot& S = G.s[i];
        S.present = q[0] != 0;
        S.hn_prev = (const float*)q[1]; S.s_prev = (const float*)q[2]; S.valid_prev = (const int*)q[3]; S.w_att = (const float*)q[4];
        S.alpha_prev = (float*)q[5]; S.s_cur = (float*)q[6]; S.whh = (const float*)q[7]; S.bhh = (const float*)q[8];
        S.gi = (const float*)q[9]; S.xin = (const float*)q[10]; S.wih = (const float*)q[11]; S.bih = (const float*)q[12];
        S.gh_out = (float*)q[13]; S.gates = (float*)q[14]; S.hn = (float*)q[15]; S.mask = (const float*)q[16]; S.xdrop = (float*)q[17]; S.ghr_out = (float*)q[18];
        if (!S.present) continue;
        any = true;
        if (!S.s_cur || !S.whh || !S.bhh || !S.gh_out || !S.gates || !S.hn) return EVAD_ERR_ARG;
        if (S.hn_prev && (!S.s_prev || !S.valid_prev || !S.w_att || !S.alpha_prev)) return EVAD_ERR_ARG;
        if (i == 1 ? (!S.xin || !S.wih || !S.bih) : !S.gi) return EVAD_ERR_ARG;
        if (((uintptr_t)S.whh & 15) || (i == 1 && ((uintptr_t)S.wih & 15))) return EVAD_ERR_ARG;
    }
    if (!any) return EVAD_OK;
    const int lds = (int)(HS_SMEM_FLOATS * sizeof(float));
    if (hs_allow_lds((const void*)k_head_stage_fwd, lds, 0)) return -2;
    hipLaunchKernelGGL(k_head_stage_fwd, dim3(HS_WGS), dim3(1024), lds, (hipStream_t)stream, G);
    return evad_last_error();
}

// =============================================================================================
// Backward of one clip stage in ONE launch (replaces cell + attention backward, the grouped products dG . W that carry
// the gradient to the previous clip's state, and the launch that summed their split-K slabs).
//
// Workgroup g again owns hidden columns 8g .. 8g+7 (column g of the coordinate GRU).  What makes the backward stage
// awkward is the attention: d alpha needs, per track, the dot product over the WHOLE row  da_k = sum_j dS[k,j] Hn[k,j],
// where dS = (element-wise part) + dG_next . Whh is itself a product that no workgroup holds completely.  Here the
// PRODUCING stage (the one that computed dG_next) already writes, per workgroup, its share of that dot product:
//     sum_j (dG_next . Whh)[k,j] Hn[k,j] = sum_n dG_next[k,n] (Whh Hn_k)[n],   (Whh Hn_k)[n] = ghr_next[k,n] / alpha_k
// -- ghr_next is the forward's raw recurrent product of the next clip, whose state is alpha_k Hn_k for a valid track --
// so a stage starts by adding 32 partial dots per track instead of waiting for a reduction over rows.  Then
//   1. the products dG_next . W for this workgroup's 8 columns: A rows straight from memory (float4 along k), the 8-column
//      slices of W staged in LDS, fp32 MFMA 16x16x4, K split over 4 waves per product, partial sums added in wave order;
//   2. attention backward + GRU cell backward on its columns (same arithmetic as k_step_bwd, head.hip);
//   3. its columns of dgi / dgh / dS, its entries of d(attention w), and its share of the previous clip's row dots.
// No float atomics; every sum has a fixed order.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HbSlot {
    const float* hn; const float* w_att; const float* alpha; const int* valid; const float* dhn_out; const float* mask;
    const float* gates; const float* gh; const float* hprev;
    float* dgi; float* dgh; float* dsprev; float* dw;
    const float* dsprev_next; const float* dgh_next; const float* whh; const float* pd_in;
    const float* dgi_x; const float* wih_x;
    const float* ghr; const float* hn_prev; const float* alpha_prev; const int* valid_prev; float* pd_out;
    int present, has_next, has_dx, has_prev;
};
struct HbGroup { HbSlot s[3]; };     // [0] layer 1 (+ dhn_out), [1] layer 0 (+ mask, dx through layer 1's input product), [2] coordinate GRU

#define HB_BLD 10                    // LDS row stride of a staged 8-column weight slice: the 4 k-groups of a wave hit disjoint banks
#define HB_SMEM_FLOATS (3 * 768 * HB_BLD + 96 + 13 * NT * 8 + 9 * NT + 2 * NT * 8 + NT)

__global__ __launch_bounds__(1024) void k_head_stage_bwd(HbGroup G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*Bs)[768][HB_BLD] = (float (*)[768][HB_BLD])smem;              // [0] whh of slot 0, [1] whh of slot 1, [2] wih_x of slot 1
    float* BsC = smem + 3 * 768 * HB_BLD;                                 // column g of the coordinate GRU's whh
    float (*Red)[NT][8] = (float (*)[NT][8])(BsC + 96);                   // [13][32][8]
    float* s_da = (float*)(Red + 13);                                     // [3][32]
    float* s_al = s_da + 3 * NT;                                          // [3][32]
    int* s_v = (int*)(s_al + 3 * NT);                                     // [3][32]
    float (*s_dw)[NT][8] = (float (*)[NT][8])(s_al + 6 * NT);             // [2][32][8]  ds * tanh(hn) of slots 0, 1
    float* s_dwc = (float*)(s_dw + 2);                                    // [32]        ... of slot 2
    const int tid = threadIdx.x, lane = tid & 63, g = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // (a scalar: what is selected by it is read with scalar loads)
    const HbSlot& S0 = G.s[0]; const HbSlot& S1 = G.s[1]; const HbSlot& S2 = G.s[2];

    // ---- the first half of each MFMA wave's A rows is requested before anything else (it arrives while the weight slices
    //      are staged); the second half is requested under the first half's MFMAs
    const int mi = lane & 15, mq = lane >> 4;
    const int mpr = wave >> 2;
    const bool mlive = wave < 12 && (mpr == 0 ? (S0.present && S0.has_next) : (mpr == 1 ? (S1.present && S1.has_next) : (S1.present && S1.has_dx)));
    const float* a0p = nullptr; const float* a1p = nullptr;
    float4 a0[12], a1[12];
    if (mlive) {
        const float* A = mpr == 0 ? S0.dgh_next : (mpr == 1 ? S1.dgh_next : S1.dgi_x);
        a0p = A + (size_t)mi * 768 + (wave & 3) * 192 + 4 * mq;
        a1p = a0p + 16 * 768;
#pragma unroll
        for (int t = 0; t < 6; ++t) { a0[t] = *(const float4*)(a0p + 16 * t); a1[t] = *(const float4*)(a1p + 16 * t); }
    }
    // ---- row dots and attention weights of the three slots, one wave per slot (the slot's descriptor is then read with
    //      scalar loads): validity, attention weight and all 32 partial dots of a track are requested here, in front of the
    //      weight slices, and added up behind them (8 partial dots at a time, after the slices, were five round trips in
    //      front of the stage's first barrier)
    const bool rowdots = wave < 3 && lane < NT;
    float pv[HS_WGS], rd_al = 0.f; int rd_v = 0; bool rd_next = false;
#pragma unroll
    for (int q = 0; q < HS_WGS; ++q) pv[q] = 0.f;
    if (rowdots) {
        const HbSlot& S = G.s[wave];
        if (S.present) {
            rd_v = S.valid[lane]; rd_al = S.alpha[lane];
            if (S.has_next) {
                rd_next = true;
#pragma unroll
                for (int q = 0; q < HS_WGS; ++q) pv[q] = S.pd_in[q * NT + lane];
            }
        }
    }
    // ---- stage this workgroup's weight columns.  All pieces of the
    //      three slices (two per thread and slice, the second for the lower half of the threads) are requested before the
    //      first is stored: written as a copy loop per slice, each trip became load -> wait for everything outstanding (the
    //      A rows above included) -> store, six round trips one after the other.  A slice that is not live reads the first 16 bytes
    //      of an operand that exists and stores nothing.
    {
        const float* Wp[3] = {S0.whh, S1.whh, S1.wih_x};
        const bool live[3] = {S0.present && S0.has_next, S1.present && S1.has_next, S1.present && S1.has_dx};
        // (hn: the one operand every present slot has -- the weight pointers of a slice that is not live may be null -- and
        // at least one slot is present)
        const float* safe = S0.present ? S0.hn : (S1.present ? S1.hn : S2.hn);
        const int n0 = tid >> 1, n1 = min((tid + 1024) >> 1, 767), half = tid & 1;
        const bool second = tid < 512;
        float4 v00, v01, v10, v11, v20, v21;
#define HB_LD(P, N, ON) (*(const float4*)((ON) ? Wp[P] + (size_t)(N) * 256 + 8 * g + 4 * half : safe))
        v00 = HB_LD(0, n0, live[0]); v01 = HB_LD(0, n1, live[0] && second);
        v10 = HB_LD(1, n0, live[1]); v11 = HB_LD(1, n1, live[1] && second);
        v20 = HB_LD(2, n0, live[2]); v21 = HB_LD(2, n1, live[2] && second);
#undef HB_LD
        float vc = 0.f;
        const bool lc = S2.present && S2.has_next && tid < 96;
        if (lc) vc = S2.whh[tid * 32 + g];
        // (every value above is needed here: no load can be moved behind a store)
        asm volatile("" : "+v"(v00.x), "+v"(v00.y), "+v"(v00.z), "+v"(v00.w), "+v"(v01.x), "+v"(v01.y), "+v"(v01.z), "+v"(v01.w),
                          "+v"(v10.x), "+v"(v10.y), "+v"(v10.z), "+v"(v10.w), "+v"(v11.x), "+v"(v11.y), "+v"(v11.z), "+v"(v11.w),
                          "+v"(v20.x), "+v"(v20.y), "+v"(v20.z), "+v"(v20.w), "+v"(v21.x), "+v"(v21.y), "+v"(v21.z), "+v"(v21.w), "+v"(vc));
#define HB_ST(P, N, V, ON) if (ON) { float* d = &Bs[P][N][4 * half]; d[0] = V.x; d[1] = V.y; d[2] = V.z; d[3] = V.w; }
        HB_ST(0, n0, v00, live[0]) HB_ST(0, n1, v01, live[0] && second)
        HB_ST(1, n0, v10, live[1]) HB_ST(1, n1, v11, live[1] && second)
        HB_ST(2, n0, v20, live[2]) HB_ST(2, n1, v21, live[2] && second)
#undef HB_ST
        if (lc) BsC[tid] = vc;
    }
    if (rowdots) {
        float da = 0.f;
        asm volatile("" : "+v"(da));                         // (keeps the first addition, and its wait, behind the weight slices)
        if (rd_next) {
#pragma unroll
            for (int q = 0; q < HS_WGS; ++q) da += pv[q];    // workgroup order, as the producing stage numbers them
        }
        s_da[wave * NT + lane] = rd_v ? da : 0.f; s_al[wave * NT + lane] = rd_al; s_v[wave * NT + lane] = rd_v;
    }
    __syncthreads();

    // ---- operands of the element-wise phase: requested now, consumed after the products
    // (a thread's slot is the same for its whole wave: as a scalar, the slot's descriptor is read with scalar loads from the
    // kernel arguments; indexed per lane every field is a vector load and a round trip of its own in front of the operand)
    const int slot = __builtin_amdgcn_readfirstlane(tid < 512 ? tid >> 8 : 2);
    const bool el = tid < 512 ? G.s[slot].present != 0 : (tid < 512 + NT && S2.present);
    const int H = slot == 2 ? 32 : 256;
    const int k = slot == 2 ? tid - 512 : (tid & 255) >> 3, jj = slot == 2 ? 0 : tid & 7, j = slot == 2 ? g : 8 * g + jj;
    float e_hn = 0.f, e_w = 0.f, e_dho = 0.f, e_mask = 1.f, e_r = 0.f, e_z = 0.f, e_n = 0.f, e_ghn = 0.f, e_hp = 0.f, e_dsn = 0.f;
    float e_gr[3] = {0.f, 0.f, 0.f}, e_hnp = 0.f, e_ap = 0.f; int e_vp = 0;
    if (el) {
        const HbSlot& S = G.s[slot];
        const size_t t = (size_t)k * H + j;
        e_hn = S.hn[t]; e_w = S.w_att[j];
        if (S.dhn_out) e_dho = S.dhn_out[t];
        if (S.has_dx && S.mask) e_mask = S.mask[t];
        const float* gt = S.gates + (size_t)k * 3 * H;
        e_r = gt[j]; e_z = gt[H + j]; e_n = gt[2 * H + j];
        e_ghn = S.gh[(size_t)k * 3 * H + 2 * H + j];
        e_hp = S.hprev[t];
        if (S.has_next) e_dsn = S.dsprev_next[t];
        if (S.has_prev) {
            const float* gr = S.ghr + (size_t)k * 3 * H;
            e_gr[0] = gr[j]; e_gr[1] = gr[H + j]; e_gr[2] = gr[2 * H + j];
            e_hnp = S.hn_prev[t]; e_ap = S.alpha_prev[k]; e_vp = S.valid_prev[k];
        }
    }

    // ---- 1. products for this workgroup's columns.  MFMA 16x16x4: lane l supplies A[l & 15][k'] and B[k'][l & 15] for
    //         k' = l >> 4, D[4 (l >> 4) + r][l & 15] comes back in register r.  A lane loads 4 consecutive k as one float4
    //         (k = 16t + 4q + c for its k-group q), so step (t, c) covers k = 16t + c, 16t + 4 + c, 16t + 8 + c, 16t + 12 + c.
    {
        const int i = lane & 15, q = lane >> 4;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (wave < 12) {
            const int pr = mpr;
            if (mlive) {
                const int kb = (wave & 3) * 192;
#pragma unroll
                for (int t = 0; t < 12; ++t) {
                    if (t + 6 < 12) { a0[t + 6] = *(const float4*)(a0p + 16 * (t + 6)); a1[t + 6] = *(const float4*)(a1p + 16 * (t + 6)); }
                    const float* bp = &Bs[pr][kb + 16 * t + 4 * q][i & 7];
                    const float b0 = i < 8 ? bp[0] : 0.f, b1 = i < 8 ? bp[HB_BLD] : 0.f, b2 = i < 8 ? bp[2 * HB_BLD] : 0.f, b3 = i < 8 ? bp[3 * HB_BLD] : 0.f;
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t].x, b0, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t].x, b0, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t].y, b1, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t].y, b1, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t].z, b2, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t].z, b2, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t].w, b3, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t].w, b3, acc1, 0, 0, 0);
                }
            }
        } else if (wave == 12 && S2.present && S2.has_next) {   // coordinate GRU: K = 96, one column
            const float* a0p = S2.dgh_next + (size_t)i * 96 + 4 * q;
            const float* a1p = a0p + 16 * 96;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const float4 x0 = *(const float4*)(a0p + 16 * t), x1 = *(const float4*)(a1p + 16 * t);
                const float* bp = &BsC[16 * t + 4 * q];
                const float b0 = i == 0 ? bp[0] : 0.f, b1 = i == 0 ? bp[1] : 0.f, b2 = i == 0 ? bp[2] : 0.f, b3 = i == 0 ? bp[3] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, b0, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, b0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, b1, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, b1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, b2, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, b2, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, b3, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, b3, acc1, 0, 0, 0);
            }
        }
        if (wave < 13 && i < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { Red[wave][4 * q + r][i] = acc0[r]; Red[wave][16 + 4 * q + r][i] = acc1[r]; }
        }
    }
    __syncthreads();

    // ---- 2. attention backward + GRU cell backward on this workgroup's columns (arithmetic of k_step_bwd, head.hip)
    float pd = 0.f;
    if (el) {
        const HbSlot& S = G.s[slot];
        const size_t t = (size_t)k * H + j;
        float gs = 0.f;
        if (S.has_next) {
            const int w0 = slot == 2 ? 12 : 4 * slot;
            const float gsum = slot == 2 ? Red[12][k][0] : ((Red[w0][k][jj] + Red[w0 + 1][k][jj]) + Red[w0 + 2][k][jj]) + Red[w0 + 3][k][jj];
            gs = gsum + e_dsn;
        }
        const float* da = s_da + slot * NT; const float* alv = s_al + slot * NT; const int* vv = s_v + slot * NT;
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < NT; ++q) if (vv[q]) dot += alv[q] * da[q];
        const bool v = vv[k] != 0;
        const float al = alv[k], ds = v ? al * (da[k] - dot) : 0.f;
        const float hn = e_hn, th = hs_tanh(hn);
        float dh = 0.f, dspass = gs;                         // state kept: the gradient passes through
        if (v) {
            dh = gs * al + ds * e_w * (1.f - th * th);
            dh += e_dho;
            dspass = 0.f;
        }
        if (slot == 2) s_dwc[k] = ds * th; else s_dw[slot][k][jj] = ds * th;
        if (S.has_dx) {
            const float dx = ((Red[8][k][jj] + Red[9][k][jj]) + Red[10][k][jj]) + Red[11][k][jj];
            dh += dx * e_mask;
        }
        const float r = e_r, z = e_z, n = e_n, ghn = e_ghn, h = e_hp;
        const float dn = dh * (1.f - z), dz = dh * (h - n);
        const float dan = dn * (1.f - n * n), daz = dz * z * (1.f - z), dar = dan * ghn * r * (1.f - r);
        float* a = S.dgi + (size_t)k * 3 * H; float* cg = S.dgh + (size_t)k * 3 * H;
        a[j] = dar; a[H + j] = daz; a[2 * H + j] = dan;
        cg[j] = dar; cg[H + j] = daz; cg[2 * H + j] = dan * r;
        const float dsp = dspass + dh * z;
        S.dsprev[t] = dsp;
        if (S.has_prev) {                                    // this workgroup's share of the previous clip's row dots
            const float inva = (e_vp != 0 && e_ap != 0.f) ? 1.f / e_ap : 0.f;
            pd = dsp * e_hnp + inva * ((dar * e_gr[0] + daz * e_gr[1]) + dan * r * e_gr[2]);
        }
    }
    if (tid < 512) {                                         // the 8 columns of a track sit in 8 neighbouring lanes
        pd += __shfl_xor(pd, 1, 64); pd += __shfl_xor(pd, 2, 64); pd += __shfl_xor(pd, 4, 64);
    }
    if (el && jj == 0 && G.s[slot].has_prev) G.s[slot].pd_out[g * NT + k] = pd;
    __syncthreads();
    // d(attention w) of this workgroup's columns: sum over the 32 tracks in track order
    if (wave < 2 && lane < 8) {                              // (one wave per slot, as above)
        const int sl = wave, c = lane;
        if (G.s[sl].present) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < NT; ++q) acc += s_dw[sl][q][c];
            G.s[sl].dw[8 * g + c] += acc;
        }
    } else if (tid == 16 && S2.present) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < NT; ++q) acc += s_dwc[q];
        S2.dw[g] += acc;
    }
}

/* desc: HOST array, 3 slots x 28 int64: slot 0 = GRU layer 1 (256 wide), slot 1 = GRU layer 0 (256 wide; receives the
 * gradient of its dropout'd output through layer 1's input product), slot 2 = coordinate GRU (32 wide).  Per slot:
 * [0] present [1] hn [2] attention w [3] alpha [4] valid [5] dhn_out (0 = none) [6] dropout mask (0 = none) [7] gates
 * [8] gh [9] hprev [10] dgi out [11] dgh out [12] dsprev out f32[32,H] (element-wise part of the gradient of the state
 * entering this clip) [13] dw accumulator f32[H] [14] has_next: the clip is not the last one, then [15] dsprev_next
 * (the [12] of the next clip) [16] dgh_next f32[32,3H] [17] whh f32[3H,H] [18] pd_in f32[32,32] partial row dots written
 * by the next clip's stage  [19] has_dx (slot 1), then [20] dgi_x f32[32,768] layer 1's dgi of this clip [21] wih_x
 * [22] has_prev: the clip is not the first one, then [23] ghr f32[32,3H] (raw recurrent product of this clip, from
 * eventad_head_stage_fwd) [24] hn_prev [25] alpha_prev [26] valid_prev [27] pd_out f32[32,32]. */
extern "C" int eventad_head_stage_bwd(const int64_t* d, void* stream) {
    if (!d) return EVAD_ERR_ARG;
    HbGroup G;
    bool any = false;
    for (int i = 0; i < 3; ++i) {
        const int64_t* q = d + 28 * i; HbSlot& S = G.s[i];
        S.present = q[0] != 0;
        S.hn = (const float*)q[1]; S.w_att = (const float*)q[2]; S.alpha = (const float*)q[3]; S.valid = (const int*)q[4];
        S.dhn_out = (const float*)q[5]; S.mask = (const float*)q[6]; S.gates = (const float*)q[7]; S.gh = (const float*)q[8];
        S.hprev = (const float*)q[9]; S.dgi = (float*)q[10]; S.dgh = (float*)q[11]; S.dsprev = (float*)q[12]; S.dw = (float*)q[13];
        S.has_next = q[14] != 0; S.dsprev_next = (const float*)q[15]; S.dgh_next = (const float*)q[16]; S.whh = (const float*)q[17];
        S.pd_in = (const float*)q[18]; S.has_dx = q[19] != 0; S.dgi_x = (const float*)q[20]; S.wih_x = (const float*)q[21];
        S.has_prev = q[22] != 0; S.ghr = (const float*)q[23]; S.hn_prev = (const float*)q[24]; S.alpha_prev = (const float*)q[25];
        S.valid_prev = (const int*)q[26]; S.pd_out = (float*)q[27];
        if (!S.present) { S.has_next = S.has_dx = S.has_prev = 0; continue; }
        any = true;
        if (!S.hn || !S.w_att || !S.alpha || !S.valid || !S.gates || !S.gh || !S.hprev || !S.dgi || !S.dgh || !S.dsprev || !S.dw) return EVAD_ERR_ARG;
        if (S.has_next && (!S.dsprev_next || !S.dgh_next || !S.whh || !S.pd_in || ((uintptr_t)S.whh & 15) || ((uintptr_t)S.dgh_next & 15))) return EVAD_ERR_ARG;
        if (S.has_dx && (i != 1 || !S.dgi_x || !S.wih_x || ((uintptr_t)S.wih_x & 15) || ((uintptr_t)S.dgi_x & 15))) return EVAD_ERR_ARG;
        if (S.has_prev && (!S.ghr || !S.hn_prev || !S.alpha_prev || !S.valid_prev || !S.pd_out)) return EVAD_ERR_ARG;
    }
    if (!any) return EVAD_OK;
    const int lds = (int)(HB_SMEM_FLOATS * sizeof(float));
    if (hs_allow_lds((const void*)k_head_stage_bwd, lds, 1)) return -2;
    hipLaunchKernelGGL(k_head_stage_bwd, dim3(HS_WGS), dim3(1024), lds, (hipStream_t)stream, G);
    return evad_last_error();
}
