// eventad_amd/csrc/common.h -- shared helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EVAD_OK 0
#define EVAD_ERR_ARG (-1)
#define EVENTAD_ERR_TOO_LARGE (-3)
#define EVAD_SCAN_TILE 1024         // slots per workgroup of the tiled slot scans (pool.hip)

// error-flag bits written by kernels into a caller-owned u32 word (checked lazily by the host)
#define EVAD_FLAG_T_NOT_MONOTONE 1u   // timestamps not non-decreasing in event order inside a clip
#define EVAD_FLAG_POS_OUT_OF_FOV 2u   // an event fell outside [0,W)x[0,H) or batch outside [0,B)
#define EVAD_FLAG_POOL_WINDOW    4u   // a pooled edge fell outside the 5x5x3 neighbour window
#define EVAD_FLAG_T_OUT_OF_RANGE 8u   // normalised t outside [0,1]
#define EVAD_FLAG_INCR_OVERFLOW 16u   // incremental update: more appended events in one pixel than the edge search collects

static inline int evad_last_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? EVAD_OK : (int)e;
}

// More than 64 KB of dynamic LDS has to be allowed per kernel and per device before the launch (not a stream operation).
// Remembered per (kernel, device) under a lock: entry points may be called from several host threads.
#include <mutex>
static inline int evad_allow_lds(const void* kernel, size_t bytes) {
    struct Entry { const void* kernel; int dev; size_t bytes; };
    static std::mutex mu;
    static Entry tab[256];
    static int n = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -2; }
    std::lock_guard<std::mutex> lock(mu);
    int at = -1;
    for (int i = 0; i < n; ++i) if (tab[i].kernel == kernel && tab[i].dev == dev) { at = i; break; }
    if (at >= 0 && tab[at].bytes >= bytes) return 0;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { (void)hipGetLastError(); return -2; }
    if (at < 0 && n < 256) at = n++;
    if (at >= 0) tab[at] = Entry{kernel, dev, bytes};
    return 0;
}

static inline unsigned evad_blocks(long long n, int threads) {
    long long b = (n + threads - 1) / threads;
    return (unsigned)(b < 1 ? 1 : b);
}

// Clear device memory with a fill KERNEL rather than a memset node: every entry point of this library may be recorded
// into a hipGraph by its caller, and memset nodes recorded on this platform have replayed wrongly after the first replay.
// p must be 4-byte aligned and bytes a multiple of 4.
static __global__ void k_evad_zero(uint32_t* __restrict__ p, size_t words) {
    const size_t head = min(words, (size_t)((16 - ((uintptr_t)p & 15)) & 15) >> 2);      // words before 16-byte alignment
    const size_t vec = (words - head) >> 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4* q = (uint4*)(p + head);
    for (size_t k = i; k < vec; k += (size_t)gridDim.x * blockDim.x) q[k] = make_uint4(0u, 0u, 0u, 0u);
    if (i < head) p[i] = 0u;
    const size_t tail0 = head + (vec << 2);
    if (i < words - tail0) p[tail0 + i] = 0u;
}
static inline void evad_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return;
    const size_t words = bytes >> 2;
    size_t blocks = (words / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_evad_zero, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, words);
}

// order-preserving float <-> uint map so float max can use integer atomicMax (0 < every real)
__device__ __forceinline__ unsigned evad_f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float evad_ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Bilinear sampling set-up of F.grid_sample(align_corners=True) as the reference calls it (net.py:211-228): normalised
// event position -> top-left map pixel + fractions, every step rounded like the reference's separate torch ops.
__device__ __forceinline__ void evad_bilinear_setup(float px_norm, float py_norm, float Wf, float Hf, int w_in, int h_in,
                                               int* x0, int* y0, float* fx, float* fy) {
    float x = __fmul_rn(px_norm, Wf), y = __fmul_rn(py_norm, Hf);
    float xn = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, x), Wf - 1.f), 1.f);
    float yn = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, y), Hf - 1.f), 1.f);
    float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xn, 1.f), 2.f), (float)(w_in - 1));
    float iy = __fmul_rn(__fdiv_rn(__fadd_rn(yn, 1.f), 2.f), (float)(h_in - 1));
    float fx0 = floorf(ix), fy0 = floorf(iy);
    *x0 = (int)fx0; *y0 = (int)fy0; *fx = ix - fx0; *fy = iy - fy0;
}

// torch_spline_conv degree-1 basis for one pseudo-coordinate: v = attr*(kernel_size-1);
// index k_mod in {0,1}: ((int64)v + k_mod) % 5,  weight 1 - frac - k_mod + 2*frac*k_mod.
__device__ __forceinline__ void evad_basis1(float attr, int* i0, int* i1, float* b0, float* b1) {
    float v = __fmul_rn(attr, 4.f);
    int t = (int)v;                                    // truncation, as the (int64_t) cast
    float fr = __fsub_rn(v, floorf(v));
    int a = t % 5, b = (t + 1) % 5;
    *i0 = a < 0 ? a + 5 : a; *i1 = b < 0 ? b + 5 : b;  // advanced indexing wraps negative indices
    *b0 = __fsub_rn(1.f, fr);
    *b1 = __fadd_rn(__fsub_rn(__fsub_rn(1.f, fr), 1.f), __fmul_rn(2.f, fr));
}


// One output channel c (= lane) of one destination node of a pooled-level spline convolution (tail.hip): the 4-tap blend of
// the sources' Z rows over the node's in-edges (ascending edge order = the reference's order inside a destination), the
// root term, folded BatchNorm, the skip branch, ReLU.  All 64 lanes of a wave must call it for the same node.
__device__ __forceinline__ float evad_tail_blend_row(const float* __restrict__ Z, int ldz, int root_col, const float* __restrict__ Zs,
                                                     int ldzs, int skip_col, const unsigned* __restrict__ rowptr,
                                                     const int* __restrict__ csr_src, const int* __restrict__ csr_dxy, int node, int c,
                                                     float div_x, float div_y, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float* __restrict__ sscale,
                                                     const float* __restrict__ sshift) {
    float acc = 0.f;
    const unsigned e0 = rowptr[node], e1 = rowptr[node + 1];
    // Up to 64 edges at a time: lane l fetches edge l's source and offset (ONE coalesced load each instead of a
    // dependent pair per edge) and works out that edge's 4 taps; the taps are then broadcast edge by edge, so the
    // Z gathers of consecutive edges are independent loads in flight together.  Messages are still added in
    // ascending edge order (= the reference's order inside a destination).
    for (unsigned eb = e0; eb < e1; eb += 64) {
        const int ne = min(64u, e1 - eb);
        int s = 0, o00 = 0, o10 = 0, o01 = 0, o11 = 0;
        float w00 = 0.f, w10 = 0.f, w01 = 0.f, w11 = 0.f;
        if (c < ne) {
            s = csr_src[eb + c];
            const int pk = csr_dxy[eb + c];
            const int dx = (int)(short)(pk & 0xffff), dy = pk >> 16;
            // LUT grid attribute (spline_conv.py:27-30): d / (2*M*size) + 0.5
            const float ax = __fadd_rn(__fdiv_rn((float)dx, div_x), 0.5f), ay = __fadd_rn(__fdiv_rn((float)dy, div_y), 0.5f);
            int ix0, ix1, iy0, iy1; float bx0, bx1, by0, by1;
            evad_basis1(ax, &ix0, &ix1, &bx0, &bx1);
            evad_basis1(ay, &iy0, &iy1, &by0, &by1);
            o00 = (ix0 + 5 * iy0) * 64; o10 = (ix1 + 5 * iy0) * 64; o01 = (ix0 + 5 * iy1) * 64; o11 = (ix1 + 5 * iy1) * 64;
            w00 = __fmul_rn(bx0, by0); w10 = __fmul_rn(bx1, by0); w01 = __fmul_rn(bx0, by1); w11 = __fmul_rn(bx1, by1);
        }
        for (int j0 = 0; j0 < ne; j0 += 4) {                 // 4 edges = 16 independent gathers per trip
            float z[4][4], ww[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + u, ne - 1);
                const float* zr = Z + (size_t)__shfl(s, j, 64) * ldz + c;
                z[u][0] = zr[__shfl(o00, j, 64)]; z[u][1] = zr[__shfl(o10, j, 64)];
                z[u][2] = zr[__shfl(o01, j, 64)]; z[u][3] = zr[__shfl(o11, j, 64)];
                ww[u][0] = __shfl(w00, j, 64); ww[u][1] = __shfl(w10, j, 64); ww[u][2] = __shfl(w01, j, 64); ww[u][3] = __shfl(w11, j, 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u < ne) {
                    float m = __fmul_rn(ww[u][0], z[u][0]);
                    m = fmaf(ww[u][1], z[u][1], m);
                    m = fmaf(ww[u][2], z[u][2], m);
                    m = fmaf(ww[u][3], z[u][3], m);
                    acc += m;
                }
            }
        }
    }
    acc += Z[(size_t)node * ldz + root_col * 64 + c];
    float y = fmaf(acc, scale[c], shift[c]);
    if (Zs) y += fmaf(Zs[(size_t)node * ldzs + skip_col * 64 + c], sscale[c], sshift[c]);
    y = fmaxf(y, 0.f);
    return y;
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

// The same for the <= 15 edges of one ELL row at once: all bitmap words are fetched first (independent loads -- the
// volatile test-then-set above serialises one round trip per edge), then only the missing bits are OR-ed in.
// A stale read costs at most a redundant atomicOr.
template <int MAXNB>
__device__ __forceinline__ void evad_mark_row_packed(const int (&vs)[MAXNB], int nn, int vd, int slot_d, unsigned* __restrict__ bitmap,
                                                     unsigned* flags) {
    unsigned* w[MAXNB]; unsigned m[MAXNB], cur[MAXNB];
    bool bad = false;
#pragma unroll
    for (int e = 1; e < MAXNB; ++e) {
        w[e] = nullptr; m[e] = 0; cur[e] = 0;
        if (e < nn && vs[e] != vd) {
            int dx = (vs[e] & 255) - (vd & 255), dy = ((vs[e] >> 8) & 255) - ((vd >> 8) & 255), dp = (vs[e] >> 16) - (vd >> 16);
            if (dx < -2 || dx > 2 || dy < -2 || dy > 2 || dp < -1 || dp > 1) bad = true;
            else {
                int bit = ((dp + 1) * 5 + (dy + 2)) * 5 + (dx + 2);
                w[e] = bitmap + (size_t)slot_d * 3 + (bit >> 5); m[e] = 1u << (bit & 31);
                cur[e] = __builtin_nontemporal_load(w[e]);
            }
        }
    }
    if (bad) atomicOr(flags, EVAD_FLAG_POOL_WINDOW);
    unsigned set0 = 0, set1 = 0, set2 = 0;                   // several edges usually hit the same bit: OR each word once
    unsigned* base = bitmap + (size_t)slot_d * 3;
#pragma unroll
    for (int e = 1; e < MAXNB; ++e) if (w[e] && !(cur[e] & m[e])) { int wi = (int)(w[e] - base); if (wi == 0) set0 |= m[e]; else if (wi == 1) set1 |= m[e]; else set2 |= m[e]; }
    // events are pixel-sorted, so the lanes of a wave fall into a few runs with the same destination voxel: OR the
    // masks down each run and let its first lane issue the atomics (3 words per run instead of per lane).
    // Must be reached by every lane of the wave (callers pass nn = 0 for idle lanes).
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int os = __shfl_down(slot_d, d, 64);
        const unsigned o0 = __shfl_down(set0, d, 64), o1 = __shfl_down(set1, d, 64), o2 = __shfl_down(set2, d, 64);
        if (lane + d < 64 && os == slot_d) { set0 |= o0; set1 |= o1; set2 |= o2; }
    }
    const int prev = __shfl_up(slot_d, 1, 64);
    if (lane == 0 || prev != slot_d) {
        if (set0) atomicOr(base, set0);
        if (set1) atomicOr(base + 1, set1);
        if (set2) atomicOr(base + 2, set2);
    }
}
