// eventad_amd/csrc/layer1.hip -- the two event-level spline convolutions of DAGR's first Layer.
//
// Reference: Layer/ConvBlock/ConvBlockWithSkip (src/dagr/model/layers/conv.py:10-72) around
// MySplineConv with the cached LUT message (src/dagr/model/layers/spline_conv.py:39-78):
//     out_i = sum_{j->i} LUT[dx_ji, dy_ji]^T x_j  +  W_root x_i ;  y = ReLU(BN(out))            (conv 1)
//     y2    = ReLU( BN(out2) + BN_skip(W_skip x_i) )                                            (conv 2)
// The reference gathers LUT[dx,dy] into an [E,Cin,Cout] temporary (36.6 GB in its own comment,
// spline_conv.py:43-44) and scatter-adds.  Here every destination owns <=16 contiguous ELL slots
// (graph.hip), the LUT (one Cin x 16 matrix per spiral step, 49 steps at r=3) lives in LDS (at larger radii: its inner
// rings, as many as fit into 152 KiB shared by the 16 waves of a CU), and the 4 lanes of a
// destination split the INPUT channels: no atomics, no temporary.
//
// Weights arrive pre-multiplied by the folded eval-mode BatchNorm scale of their output column (host side:
// eventad_amd/backbone.py), the shifts pre-summed; LUT row `steps` is the matrix of the self edge that opens every ELL
// row (LUT[0] + W_root), row `steps + 1` is all zeros (what slots past a row's degree multiply by).  So a node's output is ONE accumulation over its edges (+ the skip product) + shift, ReLU.
// The rows of every weight matrix come in the order [i][j] = i-th channel of lane j (row i*4 + j holds channel 4j+i for
// i < 4 and channel 16+j for i = 4), see eventad_amd/spline.py::quad_row_order.
//
// Mapping (64-lane waves, 16 destinations per pass, no workgroup barrier after the LUT is staged):
//  * 64 consecutive destinations are ranked by degree class (ceil(degree / 2)) with ballots and split into 4 passes of 16
//    destinations of (nearly) equal degree -- a wave pays for the longest row of a pass, in blocks of 4 slots or half a block;
//    a wave takes work items of 1 or 2 such passes from a per-XCD
//    work counter (dynamic: items differ several times in edge count, and a wave's last item is the kernel's tail);
//  * lane j of a destination's quad owns input channels {4j..4j+3} (+ channel 16+j when Cin = 20): a neighbour row
//    costs one 16-byte (+ one 4-byte) load per lane, each byte of the row fetched once per quad, and only 4-5 registers;
//    the lane keeps partial sums for all 16 outputs, reduced over the quad once per NODE with DPP adds;
//  * lane j also holds ELL slots 4j..4j+3 of the row (one int4 + one uint2); slot e reaches the quad by a DPP quad
//    broadcast (the row is rotated one lane per block of 4 slots, so every lane/component choice is static);
//  * with the [i][j][16] row order the quad reads 64 consecutive floats of the LDS copy of a LUT matrix;
//    quad a of a 16-lane group reads its four 16-byte chunks in the order (s + a) & 3, which makes the 16 lanes of a
//    ds_read_b128 pass hit 16 different bank groups whatever their cells are (cell stride = Cin*16 floats = 0 mod 64);
//  * neighbour rows are prefetched 4 slots ahead (across pass boundaries), the next pass's ELL row and the next group's id
//    one step ahead.
// Messages are added in ELL slot order (self edge, then spiral order) = the reference's edge order.
#include "common.h"

#define L1_THREADS 512          // 8 waves share one LDS copy of the LUT: 2 workgroups/CU = 16 waves/CU (a table that does not fit 80 KiB: 16 waves, one copy of 152 KiB)
#define L1_GROUP 64             // destinations ranked together (4 passes of 16)
#define L1_MAX_WG 512

// Per-lane addresses are formed as a 32-bit byte offset from a wave-uniform base pointer (arrays here are < 4 GiB), so
// the loads take the scalar-base + 32-bit-offset form instead of keeping a 64-bit address per lane in VGPRs.
template <typename T> __device__ __forceinline__ T ld_off(const void* base, unsigned byte_off) {
    return *(const T*)((const char*)base + byte_off);
}
// (x << SH) + y in one full-rate instruction.  Written as asm because the compiler turns shift-and-add chains back into
// a 32-bit multiply (v_mul_lo_u32 / v_mad_u64_u32), which issues at a quarter of the rate.
template <int SH> __device__ __forceinline__ unsigned lshl_add(unsigned x, unsigned y) {
    unsigned r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(SH), "v"(y));
    return r;
}
// j16 = 16 * j, je = 64 - 12 * j (byte offsets of the lane's 16-byte piece and, from there, of its channel 16 + j)
template <int CINP>
__device__ __forceinline__ void l1_load_row(const float* __restrict__ xin, int src, unsigned j16, unsigned je, float4& v, float& e) {
    // row bytes: 64 (16 channels) or 80 = 64 + 16 (20 channels)
    const unsigned o = CINP == 20 ? lshl_add<6>((unsigned)src, lshl_add<4>((unsigned)src, j16)) : lshl_add<6>((unsigned)src, j16);
    v = ld_off<float4>(xin, o);
    e = 0.f;
    if (CINP == 20) e = ld_off<float>(xin, o + je);
}

__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    return v;
}

// acc[s] += x * W[chunk (s+a)&3] for the KL channels of this lane, the matrix read from global memory: base = a uniform
// pointer, ad[s] = byte offset of the lane's chunk s in the matrix's first row for this lane (rows are stored [i][j], 16
// floats each, so channel i is 256 bytes further).
template <int KL>
__device__ __forceinline__ void l1_fma_glb(float4 (&acc)[4], const float4& xv, float xe, const char* __restrict__ base,
                                           const unsigned (&ad)[4]) {
    const float xs[5] = {xv.x, xv.y, xv.z, xv.w, xe};
    // One channel's four chunks are requested while the previous channel's products run: two channels' weights (8 x 16 bytes =
    // 32 registers) in flight, and no more -- unbounded hoisting of the 20 reads of an edge, and of the next edge's, runs the
    // kernel out of registers, and a spill's scratch traffic shares vmcnt with the prefetched rows.  The channel's 256 bytes
    // go to the uniform base, so that they end up in the load's immediate offset.
    float4 w[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) w[0][s] = ld_off<float4>(base, ad[s]);
#pragma unroll
    for (int i = 0; i < KL; ++i) {
        if (i + 1 < KL) {
#pragma unroll
            for (int s = 0; s < 4; ++s) w[(i + 1) & 1][s] = ld_off<float4>(base + (i + 1) * 256, ad[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 v = w[i & 1][s];
            acc[s].x = fmaf(xs[i], v.x, acc[s].x); acc[s].y = fmaf(xs[i], v.y, acc[s].y);
            acc[s].z = fmaf(xs[i], v.z, acc[s].z); acc[s].w = fmaf(xs[i], v.w, acc[s].w);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same for a matrix in LDS, addressed in bytes: ad[s] = the lane's chunk (s + a) & 3 of the matrix's first row for this
// lane (one multiply-add per chunk and edge, see l1_edge); channel i is 256 bytes further (an immediate offset).
template <int KL, bool PIPE>
__device__ __forceinline__ void l1_fma_lds(float4 (&acc)[4], const float4& xv, float xe, const float* __restrict__ smem0,
                                           const unsigned (&ad)[4]) {
    const float xs[5] = {xv.x, xv.y, xv.z, xv.w, xe};
    if (PIPE) {
        // one channel's four chunks are requested while the previous channel's products run (two channels = 32 registers in
        // flight); the scheduling barriers keep that order
        float4 w[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) w[0][s] = *(const float4*)((const char*)smem0 + ad[s]);
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            if (i + 1 < KL) {
#pragma unroll
                for (int s = 0; s < 4; ++s) w[(i + 1) & 1][s] = *(const float4*)((const char*)smem0 + ad[s] + (i + 1) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 v = w[i & 1][s];
                acc[s].x = fmaf(xs[i], v.x, acc[s].x); acc[s].y = fmaf(xs[i], v.y, acc[s].y);
                acc[s].z = fmaf(xs[i], v.z, acc[s].z); acc[s].w = fmaf(xs[i], v.w, acc[s].w);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // (the instantiations that also carry the global-memory branch have no registers for two channels in flight)
#pragma unroll
        for (int i = 0; i < KL; ++i) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 v = *(const float4*)((const char*)smem0 + ad[s] + i * 256);
                acc[s].x = fmaf(xs[i], v.x, acc[s].x); acc[s].y = fmaf(xs[i], v.y, acc[s].y);
                acc[s].z = fmaf(xs[i], v.z, acc[s].z); acc[s].w = fmaf(xs[i], v.w, acc[s].w);
            }
            if (i & 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int K> __device__ __forceinline__ int quad_bcast(int v) {      // value of lane K of the quad, in all four lanes
    return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xf, 0xf, true);
}
__device__ __forceinline__ int quad_rotate(int v) {                       // lane j takes the value of lane (j + 1) & 3
    return __builtin_amdgcn_update_dpp(0, v, 0x39, 0xf, 0xf, true);       // quad_perm [1,2,3,0]
}
template <int S> __device__ __forceinline__ int comp4(const int4& v) { return S == 0 ? v.x : S == 1 ? v.y : S == 2 ? v.z : v.w; }

// One ELL slot (slot 4b + S of the current row; the quad keeps the current block's slots in lane 0 and the next block's in
// lane 1 by rotating the row one lane per block): consume ring buffer S, refill it with the row that will be consumed 4
// slots later -- slot 4(b+1) + S of this row, or, in the row's last block, slot S of the NEXT pass's row, so the first
// rows of the next pass are in flight while this pass reduces and stores.
// Branch-free on purpose: a load inside a divergent branch lands in a temporary that the compiler can only copy into
// the ring buffer after waiting for ALL outstanding loads, which serialises the prefetch.  Slots past the row's degree
// multiply a cached row (the destination's own) by the all-zero matrix LUT[steps + 1] instead.
template <int CINP, int KL, int S, bool ALL_IN_LDS>
__device__ __forceinline__ void l1_edge(float4 (&acc)[4], float4 (&xv)[4], float (&xe)[4], const int4& sv, const uint2& cv,
                                        const int4& sv2, int b, bool last, bool half, int d, int d2, unsigned j16, unsigned je,
                                        int safe, int safe2, const float* __restrict__ xin, const float* __restrict__ smem0,
                                        const unsigned (&lbase)[4], const char* __restrict__ lutg,
                                        int nlds, int steps) {
    constexpr int LS = CINP * 16;
    // byte address of chunk s of matrix `c` in the LDS table (and, from `lutg`, in the table in global memory): a 24-bit
    // multiply (full rate; cells and LS * 4 are far below 2^24) and one add per chunk
    auto mat_addr = [&](int c, unsigned (&ad)[4]) {
        const unsigned m = __umul24((unsigned)c, (unsigned)(LS * 4));
#pragma unroll
        for (int s = 0; s < 4; ++s) ad[s] = m + lbase[s];
    };
    const int e = 4 * b + S;
    const int pair = quad_bcast<0>((int)(S & 2 ? cv.y : cv.x));
    int cell = (S & 1) ? (int)((unsigned)pair >> 16) : (pair & 0xffff);
    if (S == 0) cell = b == 0 ? steps : cell;                            // the self edge reads the folded root matrix
    cell = e < d ? cell : steps + 1;
    int nsrc = quad_bcast<1>(comp4<S>(sv));
    nsrc = e + 4 < d ? nsrc : safe;
    int nnext = quad_bcast<0>(comp4<S>(sv2));
    nnext = S < d2 ? nnext : safe2;
    const int src = last ? nnext : nsrc;
    // `half` (uniform): the row's last block holds at most two live slots -- slots 2 and 3 only refill their buffers
    if (S < 2 || !half) {
        unsigned ad[4];
        if (ALL_IN_LDS) { mat_addr(cell, ad); l1_fma_lds<KL, true>(acc, xv[S], xe[S], smem0, ad); }
        else {
            // Large search radius: only the first `nlds` spiral steps (the inner rings, where the <= 16 neighbours of a dense
            // window lie) + the two special matrices are in LDS.  A slot in which ANY lane of the wave needs an outer ring
            // reads its matrices from global memory (L2) for the whole wave -- rare, and wave-uniform, so no lane diverges.
            const bool outer = cell >= nlds && cell < steps;
            if (__any(outer)) { mat_addr(cell, ad); l1_fma_glb<KL>(acc, xv[S], xe[S], lutg, ad); }
            else { mat_addr(cell < nlds ? cell : nlds + cell - steps, ad); l1_fma_lds<KL, true>(acc, xv[S], xe[S], smem0, ad); }
        }
    }
    // the refill comes after the products that read the buffer, so the old row needs no copy
    l1_load_row<CINP>(xin, src, j16, je, xv[S], xe[S]);
}

template <int CINP, bool SKIP, bool ALL_IN_LDS, int THREADS, bool BY_TAIL>
__global__ __launch_bounds__(THREADS, 4) void k_l1_conv(const float* __restrict__ xin, const int* __restrict__ ell_src,
                                                 const uint16_t* __restrict__ ell_step, const int* __restrict__ deg,
                                                 const float* __restrict__ lut, const float* __restrict__ shift,
                                                 const float* __restrict__ xskip, const float* __restrict__ wskip,
                                                 float* __restrict__ out, unsigned* __restrict__ work, int n_first, int N, int steps, int nlds) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KL = CINP == 20 ? 5 : 4;                // input channels per lane
    constexpr int LS = CINP * 16;                         // floats per LUT matrix
    static_assert(CINP == 20 || CINP == 16, "lane/channel map is written for 20 or 16 input channels");
    float* s_skip = smem;                                 // [5][4][16] (SKIP only)
    float* s_lut = s_skip + (SKIP ? 20 * 16 : 0);         // [nlds+2][KL][4][16]: spiral steps 0..nlds-1, self-edge matrix, zeros
    const int tid = threadIdx.x;
    const int lane = tid & 63, j = lane & 3, q = lane >> 2, a = q & 3;
    // ---- once per workgroup: LUT and skip weights (already in the [i][j][16] row order) into LDS.  The table goes there
    //      directly (global_load_lds_dwordx4: a wave-instruction writes 64 x 16 bytes to a wave-uniform LDS address, no
    //      registers in between), every piece requested at once -- a load -> wait -> ds_write loop paid one L2 round trip per
    //      trip, 8 to 10 of them.
    const int xcd = blockIdx.x & 7;
    {
        const float4* g4 = (const float4*)lut;
        const int n4 = nlds * LS / 4, m4 = 2 * LS / 4, skip4 = (steps - nlds) * LS / 4, tot = n4 + m4;
        constexpr int TRIPS = 10, WAVES = THREADS / 64;   // the host checks that this covers the table: 80 KiB / (512 x 16 B), 152 KiB / (1024 x 16 B)
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int u = 0; u < TRIPS; ++u) {
            const int piece = u * WAVES + wv, t = piece * 64 + lane;         // 1 KiB pieces, dealt round-robin to the waves
            if (t < tot)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g4 + (t < n4 ? t : t + skip4)),
                                                 (__attribute__((address_space(3))) void*)((char*)s_lut + piece * 1024), 16, 0, 0);
        }
    }
    if (SKIP) { if (tid < 80) ((float4*)s_skip)[tid] = ((const float4*)wskip)[tid]; }
    const float4 sh = *(const float4*)(shift + 4 * j);
    // per-lane byte offsets, kept in registers (opaque to the compiler, which would otherwise rebuild them from j and a with
    // shifts at every use): the lane's piece of a row, its channel 16 + j from there, and lbase[s] = where the lane's chunk
    // (s + a) & 3 of a matrix's first row for this lane lies in the LDS table's first matrix
    constexpr unsigned LUT_OFF = SKIP ? 20 * 16 * 4 : 0;  // the table's byte offset in LDS
    unsigned j16 = 16u * (unsigned)j, je = 64u - 12u * (unsigned)j;
    unsigned lbase[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) lbase[s] = LUT_OFF + 64u * (unsigned)j + 16u * (unsigned)((s + a) & 3);
    asm volatile("" : "+v"(j16), "+v"(je), "+v"(lbase[0]), "+v"(lbase[1]), "+v"(lbase[2]), "+v"(lbase[3]));
    const char* lutg = (const char*)lut - LUT_OFF;        // the same offsets address the table in global memory from here
    constexpr bool by_tail = BY_TAIL;                     // rank the rows of a group by their outer tail, see below (the host's choice)
    static_assert(!(BY_TAIL && ALL_IN_LDS), "a table that is resident as a whole has no outer tail");
    // Work items: passes of groups of 64 consecutive destinations, handed out by one counter per XCD (blockIdx & 7: workgroups
    // are dealt round-robin to the 8 XCDs).  The waves of one XCD walk ONE contiguous eighth of the pixel-sorted node array, so
    // the neighbour rows they gather -- a few sensor rows up or down -- are mostly in that XCD's own L2.
    // (Handing a wave a fixed share of the items up front instead wins where items are short -- 256x256: 64 / 58 us with two
    // thirds of the items fixed against 73 / 66 -- and loses where they are long: the waves drift apart in the array and out
    // of each other's cache lines, 1280x720: +10 % at half the items fixed, 640x480: +3 %.  What the fixed items saved at
    // 256x256 was mostly the counter's round trip in front of every item's loads, see below.  Several counters per XCD, each
    // with a share of its groups and its workgroups: 2 / 4 / 8 of them 65 / 65 / 66 us against 68 at 256x256, +1 / 3 / 5 % at
    // 1280x720; requests for several items at a time: no gain.)
    // Passes per work item.  A wave's last item is the kernel's tail, and items differ several times in edge count, so small
    // items balance better -- but every item of a group repeats the group's ranking.  Measured: one pass per item wins where
    // passes are uneven or long (tail-ranked 1280x720: conv1 3.04 -> 2.61 ms, conv2 1.95 -> 1.81; conv1 at 640x480: 8.6 ->
    // 8.15 ms), two passes for the 16-channel convolution otherwise (640x480: 6.8 against 6.9 ms; 256x256: 79 against 81 us).
    constexpr int L1_ITEM_PASSES = (BY_TAIL || !SKIP) ? 1 : 2;
    constexpr int IPG = 4 / L1_ITEM_PASSES;               // work items per group
    // destinations [n_first, N): the whole node array, or the rows appended by an incremental update
    const int ngroups = (N - n_first + L1_GROUP - 1) / L1_GROUP, g8 = (ngroups + 7) >> 3;
    const int gbeg = xcd * g8, nitems = (min(gbeg + g8, ngroups) - gbeg) * IPG;
    unsigned* ctr = work + 32 * xcd;                      // one 128-byte line per XCD: counters sharing a line serialise like one word
    // A wave's first item is its own number among the XCD's waves (the grid is a multiple of 8 workgroups): no round trip
    // before the first row, and no burst of one request per wave on the eight counters when the kernel starts (requests to
    // one address are served one after the other, at the memory side of the chip); the counter hands out the items after those.
    const int waves_xcd = (int)(gridDim.x >> 3) * (THREADS / 64);
    int item = (int)(blockIdx.x >> 3) * (THREADS / 64) + __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned next_raw = 0;
    // The only workgroup barrier -- table complete -- is passed right before a wave's first read of the table, i.e. after its
    // first item's dependent loads (degrees -> ELL row -> neighbour rows) are under way; a wave without items passes it at
    // the end.  Every wave executes exactly one of the two (the conditions are wave-uniform); from there on it is on its own.
    bool staged = false;
    while (item < nitems) {
        // Every item requests the next one.  Loads return in order: a request issued at the top of the item would sit in
        // front of the item's own first loads (degrees, ELL row, neighbour rows), and every wait for one of those would
        // wait for the counter's answer as well (256x256: 74 / 64 us, against 69 / 59 us this way).  It is issued when the
        // item's last pass has its loads under way (or after the passes, when they end early) and is waited for at the
        // item's end.
        bool request = true;
        auto ask = [&]() {
            next_raw = 0;
            if (lane == 0) next_raw = atomicAdd(ctr, 1u);
            request = false;
        };
        const int g = gbeg + item / IPG, p0 = (item % IPG) * L1_ITEM_PASSES;
        const int n0 = n_first + g * L1_GROUP;
        // rank the 64 destinations by degree class (descending; ties by lane) with 9 ballots, then invert the ranking.  A pass
        // costs whole blocks of 4 slots, or half a block for 1-2 slots, so degrees 2c - 1 and 2c cost the same and need not be
        // told apart: the passes' costs are those of an exact ranking, at half the ballots.
        const int d_l = n0 + lane < N ? min(ld_off<int>(deg, 4u * (unsigned)(n0 + lane)), 16) : 0;
        const int c_l = (d_l + 1) >> 1;
        int rank = 0, base = 0;
#pragma unroll
        for (int v = 8; v >= 0; --v) {
            const unsigned long long m = __ballot(c_l == v);
            // base + number of lower lanes in the class (v_mbcnt_lo / v_mbcnt_hi: two instructions)
            const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
            if (c_l == v) rank = r;
            base += __popcll(m);
        }
        if (by_tail) {
            // Large radius: a slot reads its matrices from global memory for the whole wave as soon as ONE of the pass's 16
            // rows has a cell outside the LDS-resident rings there.  Cells ascend along a row (spiral order), so a row's
            // outer cells are its tail: ranking the rows by the slot where that tail begins (rows without one first), and
            // by degree only among equals, gathers the rows that need the slow branch from early on into the same passes.
            // (Two stable rankings, least significant key first: the degree ranking above, then this one over its order.)
            // Only where most of the table is NOT resident: at 640x480 (117 of 225 cells resident, 2 % of the rows with a tail)
            // the second ranking costs more than it saves (C5 conv1 8.5 -> 9.1 ms), at 1280x720 (117 of 729) it takes C4's two
            // convolutions from 4.0 + 3.1 to 3.1 + 2.0 ms.
            const uint4* srow = (const uint4*)((const char*)ell_step + (size_t)min(n0 + lane, N - 1) * 32u);
            const uint4 s0 = srow[0], s1 = srow[1];
            const unsigned w[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            unsigned outer = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                outer |= ((int)(w[k] & 0xffffu) >= nlds ? 1u : 0u) << (2 * k) | ((int)(w[k] >> 16) >= nlds ? 1u : 0u) << (2 * k + 1);
            outer &= (1u << d_l) - 1u & ~1u;                                 // live slots only; slot 0 is the self edge
            const int f_l = outer ? __ffs((int)outer) - 1 : 16;
            const int f_r = __builtin_amdgcn_ds_permute(rank << 2, f_l);     // position r of the degree order <- its row's f
            int rank2 = 0, base2 = 0;
#pragma unroll
            for (int v = 16; v >= 0; --v) {
                const unsigned long long m = __ballot(f_r == v);
                const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base2));
                if (f_r == v) rank2 = r;
                base2 += __popcll(m);
            }
            rank = __builtin_amdgcn_ds_bpermute(rank << 2, rank2);           // this row's place in the final order
        }
        const int perm = __builtin_amdgcn_ds_permute(rank << 2, lane);      // lane r <- the lane whose rank is r
        const int dperm = __builtin_amdgcn_ds_permute(rank << 2, d_l);
        int nl = __shfl(perm, 16 * p0 + q, 64), d = __shfl(dperm, 16 * p0 + q, 64);
        int safe = min(n0 + nl, N - 1);                   // a row that exists: lanes past the end / past their degree load it
        int4 sv = ld_off<int4>(ell_src, (unsigned)safe * 64u + 16u * (unsigned)j);
        uint2 cv = ld_off<uint2>(ell_step, (unsigned)safe * 32u + 8u * (unsigned)j);
        float4 xv[4]; float xe[4];
        {
            const int s0 = quad_bcast<0>(sv.x), s1 = quad_bcast<0>(sv.y), s2 = quad_bcast<0>(sv.z), s3 = quad_bcast<0>(sv.w);
            const int first[4] = {s0, s1, s2, s3};
#pragma unroll
            for (int u = 0; u < 4; ++u) l1_load_row<CINP>(xin, u < d ? first[u] : safe, j16, je, xv[u], xe[u]);
        }
#pragma unroll 1
        for (int p = p0; p < p0 + L1_ITEM_PASSES; ++p) {
            int dmax;
            if (!by_tail) {
                // ranked by degree class: quad 0 of the pass is in the pass's highest class, whose upper end bounds every row
                dmax = (__builtin_amdgcn_readfirstlane(d) + 1) & ~1;
                if (dmax == 0) break;                                      // ... and the passes after an empty one are empty
            } else {                                                       // ranked by the outer tail first: any quad may
                int m = d;
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) m = max(m, __shfl_xor(m, o, 64));
                dmax = __builtin_amdgcn_readfirstlane(m);
            }
            const int node = n0 + nl;
            // the next pass's row (ELL slots + degree) is fetched now
            const bool more = p + 1 < p0 + L1_ITEM_PASSES;
            const int r2 = 16 * (more ? p + 1 : p) + q;
            const int nl2 = __shfl(perm, r2, 64), d2 = more ? __shfl(dperm, r2, 64) : 0;
            const int safe2 = min(n0 + nl2, N - 1);
            const int4 sv2 = ld_off<int4>(ell_src, (unsigned)safe2 * 64u + 16u * (unsigned)j);
            const uint2 cv2 = ld_off<uint2>(ell_step, (unsigned)safe2 * 32u + 8u * (unsigned)j);
            if (by_tail && dmax == 0 && more) {
                // an empty pass in front of one that is not (possible only in this ranking): no block runs, so the next pass's
                // first rows, which a pass's last block requests, are requested here
                const int t0 = quad_bcast<0>(sv2.x), t1 = quad_bcast<0>(sv2.y), t2 = quad_bcast<0>(sv2.z), t3 = quad_bcast<0>(sv2.w);
                const int nxt[4] = {t0, t1, t2, t3};
#pragma unroll
                for (int u = 0; u < 4; ++u) l1_load_row<CINP>(xin, u < d2 ? nxt[u] : safe2, j16, je, xv[u], xe[u]);
            }
            // the destination's own input row for the skip branch: requested ahead of the edge loop, except in the one
            // instantiation where holding it across the loop costs spills (there it is loaded when it is needed)
            constexpr bool LATE_SKIP = SKIP && !ALL_IN_LDS && !BY_TAIL;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f); float ke = 0.f;
            if (SKIP && !LATE_SKIP) l1_load_row<20>(xskip, safe, j16, je, kv, ke);
            float4 acc[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!staged) { __syncthreads(); staged = true; }
            if (request && !more) ask();
#pragma unroll 1
            for (int b = 0; 4 * b < dmax; ++b) {
                const bool last = 4 * (b + 1) >= dmax, half = dmax - 4 * b <= 2;
#define L1_E(S) l1_edge<CINP, KL, S, ALL_IN_LDS>(acc, xv, xe, sv, cv, sv2, b, last, half, d, d2, j16, je, safe, safe2, xin, smem, lbase, lutg, nlds, steps)
                L1_E(0); L1_E(1); L1_E(2); L1_E(3);
#undef L1_E
                sv.x = quad_rotate(sv.x); sv.y = quad_rotate(sv.y); sv.z = quad_rotate(sv.z); sv.w = quad_rotate(sv.w);
                cv.x = (unsigned)quad_rotate((int)cv.x); cv.y = (unsigned)quad_rotate((int)cv.y);
            }
            // skip branch on the layer's input row (conv.py:50).  Behind a condition the compiler cannot see through (always
            // true): as straight-line code its 20 weight reads are scheduled into the end of the edge loop's last trip and the
            // reduction below, which costs the kernel 9 to 14 spilled registers; a block of its own keeps them out.  Not in
            // the tail-ranked instantiation (1280x720): its passes are short, and there the same overlap is worth 13 % of
            // the kernel's time (conv2 on 4 M events: 1.94 ms with, 2.20 ms without); with one pass per work item it has
            // the registers for it.
            int skip_on = 1;
            asm volatile("" : "+s"(skip_on));
            if (SKIP && (BY_TAIL || skip_on)) {
                if (LATE_SKIP) l1_load_row<20>(xskip, safe, j16, je, kv, ke);
                unsigned sk = 0u - LUT_OFF;               // (the skip matrix lies at the start of LDS, in the table's row order)
                asm volatile("" : "+v"(sk));              // opaque: or the 20 loop-invariant weight reads are hoisted out of
                const unsigned ad[4] = {lbase[0] + sk, lbase[1] + sk, lbase[2] + sk, lbase[3] + sk};   // every loop and held in
                l1_fma_lds<5, false>(acc, kv, ke, smem, ad);                                           // 80 registers for the kernel's lifetime
            }
            // reduce the partial sums over the quad; lane j keeps output chunk j, which sits in accumulator (j - a) & 3
            float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const int mine = (j - a) & 3;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 t = make_float4(quad_sum(acc[s].x), quad_sum(acc[s].y), quad_sum(acc[s].z), quad_sum(acc[s].w));
                if (s == mine) r4 = t;
            }
            if (d > 0) {
                float4 y = make_float4(fmaxf(r4.x + sh.x, 0.f), fmaxf(r4.y + sh.y, 0.f), fmaxf(r4.z + sh.z, 0.f), fmaxf(r4.w + sh.w, 0.f));
                *(float4*)((char*)out + ((unsigned)node * 64u + 16u * (unsigned)j)) = y;
            }
            safe = safe2; nl = nl2; d = d2; sv = sv2; cv = cv2;
        }
        if (request) ask();                               // (the passes ended early)
        item = waves_xcd + (int)__builtin_amdgcn_readfirstlane(next_raw);
    }
    if (!staged) __syncthreads();
}

template <int CINP, bool SKIP>
static int launch_l1(const float* xin, const int* ell_src, const uint16_t* ell_step, const int* deg, const float* lut,
                     const float* shift, const float* xskip, const float* wskip, float* out, unsigned* work, int n_first, int N,
                     int steps, int max_nb, hipStream_t st) {
    constexpr int LS = CINP * 16;
    if (max_nb != 16) return EVAD_ERR_ARG;              // an ELL row is read as 4 x int4 + 4 x uint2
    if ((unsigned long long)N * 80ull > 0xffffffffull) return EVAD_ERR_ARG;     // 32-bit byte offsets into the row arrays (53 M events)
    // LDS: the whole table when two 8-wave workgroups per CU can each hold it (80 KiB; radius <= 3); else ONE 16-wave
    // workgroup per CU with 152 KiB -- twice the inner rings in LDS for the same 16 waves per CU
    const size_t fixed = sizeof(float) * (SKIP ? 320 : 0);
    const bool whole = fixed + sizeof(float) * (size_t)(steps + 2) * LS <= 80 * 1024;
    const size_t budget = whole ? 80 * 1024 : 152 * 1024;
    int nlds = (int)((budget - fixed) / (sizeof(float) * LS)) - 2;               // spiral steps whose matrices fit beside the 2 special ones
    if (nlds > steps) nlds = steps;
    const size_t lds = fixed + sizeof(float) * (size_t)(nlds + 2) * LS;
    const int ngroups = (int)evad_blocks(N - n_first, L1_GROUP);
    const int threads = whole ? L1_THREADS : 2 * L1_THREADS, waves = threads / 64;
    if ((size_t)(nlds + 2) * LS / 4 > (size_t)10 * threads) return EVAD_ERR_ARG;    // the kernel stages the table in 10 trips of 16 bytes per lane
    int wgs = (ngroups + waves - 1) / waves;            // at least one workgroup per `waves` groups ...
    const int max_wg = whole ? L1_MAX_WG : L1_MAX_WG / 2;
    if (wgs > max_wg) wgs = max_wg;                     // ... at most 16 waves per CU: the waves pull groups until none is left
    wgs = (wgs + 7) / 8 * 8;                            // every XCD gets the same number of workgroups
    dim3 grid((unsigned)wgs), block((unsigned)threads);
    if (nlds == steps && whole) {
        auto kern = k_l1_conv<CINP, SKIP, true, L1_THREADS, false>;
        if (evad_allow_lds((const void*)kern, lds)) return -2;
        hipLaunchKernelGGL(kern, grid, block, lds, st, xin, ell_src, ell_step, deg, lut, shift, xskip, wskip, out, work, n_first, N, steps, nlds);
    } else if (3 * nlds < steps) {      // large radius, most of the table not resident: rows ranked by their outer tail (see the kernel)
        auto kern = k_l1_conv<CINP, SKIP, false, 2 * L1_THREADS, true>;
        if (evad_allow_lds((const void*)kern, lds)) return -2;
        hipLaunchKernelGGL(kern, grid, block, lds, st, xin, ell_src, ell_step, deg, lut, shift, xskip, wskip, out, work, n_first, N, steps, nlds);
    } else {                            // large radius: inner rings in LDS, outer rings read from global memory / L2
        auto kern = k_l1_conv<CINP, SKIP, false, 2 * L1_THREADS, false>;
        if (evad_allow_lds((const void*)kern, lds)) return -2;
        hipLaunchKernelGGL(kern, grid, block, lds, st, xin, ell_src, ell_step, deg, lut, shift, xskip, wskip, out, work, n_first, N, steps, nlds);
    }
    return evad_last_error();
}

// conv 1: x0s[N,20] (19 channels + pad) -> y1[N,16].  lut f32[steps+2][20][16], pre-scaled, row `steps` = self-edge matrix, row `steps+1` = zeros.
extern "C" int eventad_l1_conv1(const float* x0s, const int32_t* ell_src, const uint16_t* ell_step, const int32_t* deg,
                                const float* lut, const float* shift, float* y1, uint32_t* work, int N, int steps,
                                int max_num_neighbors, void* stream) {
    if (!x0s || !ell_src || !ell_step || !deg || !lut || !shift || !y1 || !work || N < 0 || steps < 1) return EVAD_ERR_ARG;
    if (N == 0) return EVAD_OK;
    return launch_l1<20, false>(x0s, ell_src, ell_step, deg, lut, shift, nullptr, nullptr, y1, work, 0, N, steps, max_num_neighbors,
                                (hipStream_t)stream);
}

// conv 2 + skip branch: y1[N,16], x0s[N,20] -> y2[N,16].  lut f32[steps+2][16][16] and wskip f32[20][16] pre-scaled,
// shift = folded BN shift of the conv + folded BN shift of the skip branch.
extern "C" int eventad_l1_conv2(const float* y1, const float* x0s, const int32_t* ell_src, const uint16_t* ell_step,
                                const int32_t* deg, const float* lut, const float* wskip, const float* shift, float* y2,
                                uint32_t* work, int N, int steps, int max_num_neighbors, void* stream) {
    if (!y1 || !x0s || !ell_src || !ell_step || !deg || !lut || !shift || !wskip || !y2 || !work || N < 0 || steps < 1)
        return EVAD_ERR_ARG;
    if (N == 0) return EVAD_OK;
    return launch_l1<16, true>(y1, ell_src, ell_step, deg, lut, shift, x0s, wskip, y2, work, 0, N, steps, max_num_neighbors,
                               (hipStream_t)stream);
}

// The two convolutions on the rows [n_first, N) only: an incremental update appends nodes whose in-edges come from older
// nodes, so the rows of the existing nodes do not change (edges point from older to newer events).  work: 256 words each,
// cleared here.  Bit-identical to the rows a pass over all N nodes writes.
extern "C" int eventad_l1_conv1_rows(const float* x0s, const int32_t* ell_src, const uint16_t* ell_step, const int32_t* deg,
                                     const float* lut, const float* shift, float* y1, uint32_t* work, int n_first, int N,
                                     int steps, int max_num_neighbors, void* stream) {
    if (!x0s || !ell_src || !ell_step || !deg || !lut || !shift || !y1 || !work || n_first < 0 || N < n_first || steps < 1) return EVAD_ERR_ARG;
    if (N == n_first) return EVAD_OK;
    evad_zero_async(work, 256 * sizeof(uint32_t), (hipStream_t)stream);
    return launch_l1<20, false>(x0s, ell_src, ell_step, deg, lut, shift, nullptr, nullptr, y1, work, n_first, N, steps,
                                max_num_neighbors, (hipStream_t)stream);
}

extern "C" int eventad_l1_conv2_rows(const float* y1, const float* x0s, const int32_t* ell_src, const uint16_t* ell_step,
                                     const int32_t* deg, const float* lut, const float* wskip, const float* shift, float* y2,
                                     uint32_t* work, int n_first, int N, int steps, int max_num_neighbors, void* stream) {
    if (!y1 || !x0s || !ell_src || !ell_step || !deg || !lut || !shift || !wskip || !y2 || !work || n_first < 0 || N < n_first || steps < 1)
        return EVAD_ERR_ARG;
    if (N == n_first) return EVAD_OK;
    evad_zero_async(work, 256 * sizeof(uint32_t), (hipStream_t)stream);
    return launch_l1<16, true>(y1, ell_src, ell_step, deg, lut, shift, x0s, wskip, y2, work, n_first, N, steps, max_num_neighbors,
                               (hipStream_t)stream);
}
