// eventad_amd/csrc/roi.hip -- per-object ROI pooling of the out4 node features.
//
// Reference: EventADModel.extract_features / _process_bbox (models/EventAD.py:335-403, 406-499): a
// Python loop over clips and boxes with ~10 tiny kernels and 6 .item() syncs per box.  Here: one
// wave per box row, lane = feature channel, no host round trip.
//   nodes of the clip inside the normalised box -> inverse-distance weights 1/(d+1e-6), normalised;
//   else mean of the min(5,n) nearest nodes; else (clip has no nodes) the clip mean = zeros.
// Box scalars follow the reference's Python-double arithmetic (x/width etc.), then the fp32 casts
// torch applies when a Python float meets an fp32 tensor.
#include "common.h"

#define ROI_MAX_BOXES 30

// One wave per box row.  Node tests run 64 nodes at a time (lane = node); everything whose floating-point result
// depends on order -- the weight sum, the 5-nearest insertion, the weighted feature sum -- is then replayed in
// ascending node order from ballot masks, so the results are those of the sequential loops of the reference.
__global__ __launch_bounds__(64) void k_roi_pool(const float* __restrict__ nx, const float* __restrict__ npos,
                                                 const int* __restrict__ nbatch, const int* __restrict__ n_dev, int n_max,
                                                 const float* __restrict__ bbox, const long long* __restrict__ bbox_batch,
                                                 int M, int frame, int n_frames, float Wimg, float Himg, int B,
                                                 float* __restrict__ feats /* [B, n_frames, 31, 64] */) {
    const int m = blockIdx.x, c = threadIdx.x;
    if (m >= M) return;
    const float* bb = bbox + (size_t)m * 6;
    const int b = (int)bbox_batch[m];
    const int track = (int)bb[5];
    if (track <= 0 || track > ROI_MAX_BOXES || b < 0 || b >= B) return;
    // the reference processes rows in order and overwrites: the LAST row of (clip, track) wins
    for (int base = m + 1; base < M; base += 64) {
        const int m2 = base + c;
        const bool later = m2 < M && (int)bbox_batch[m2] == b && (int)bbox[(size_t)m2 * 6 + 5] == track;
        if (__ballot(later)) return;
    }
    const double x = bb[0], y = bb[1], w = bb[2], h = bb[3];
    const double x1d = x / (double)Wimg, y1d = y / (double)Himg, x2d = (x + w) / (double)Wimg, y2d = (y + h) / (double)Himg;
    const float x1 = (float)x1d, y1 = (float)y1d, x2 = (float)x2d, y2 = (float)y2d;
    const float cx = (float)((x1d + x2d) / 2), cy = (float)((y1d + y2d) / 2);
    const int n = min(*n_dev, n_max);
    // pass 1: count nodes of the clip / inside the box, weight sum, 5 nearest
    int n_clip = 0, n_in = 0;
    float wsum = 0.f;
    float bd[5]; int bi[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { bd[k] = 3.4e38f; bi[k] = -1; }
    for (int base = 0; base < n; base += 64) {
        const int i = base + c;
        const bool mine = i < n && nbatch[i] == b;
        float px = 0.f, py = 0.f;
        if (mine) { px = npos[3 * i]; py = npos[3 * i + 1]; }
        const float dx = px - cx, dy = py - cy;
        const float d = sqrtf(dx * dx + dy * dy);
        const bool in = mine && px >= x1 && px <= x2 && py >= y1 && py <= y2;
        const float wl = 1.0f / (d + 1e-6f);
        const unsigned long long mclip = __ballot(mine);
        unsigned long long min_ = __ballot(in);
        n_clip += __popcll(mclip); n_in += __popcll(min_);
        while (min_) { const int l = __ffsll((long long)min_) - 1; min_ &= min_ - 1; wsum += __shfl(wl, l, 64); }
        unsigned long long mc = __ballot(mine && d < bd[4]);         // candidates of the 5 smallest (stable in index)
        while (mc) {
            const int l = __ffsll((long long)mc) - 1; mc &= mc - 1;
            const float cd = __shfl(d, l, 64);
            if (cd < bd[4]) {
                int k = 4;
                while (k > 0 && cd < bd[k - 1]) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; --k; }
                bd[k] = cd; bi[k] = base + l;
            }
        }
    }
    float* o = feats + (((size_t)b * n_frames + frame) * (ROI_MAX_BOXES + 1) + track) * 64;
    if (n_clip == 0) { o[c] = 0.f; return; }                     // global feature of an empty clip = zeros
    float acc = 0.f;
    if (n_in > 0) {
        for (int base = 0; base < n; base += 64) {
            const int i = base + c;
            const bool mine = i < n && nbatch[i] == b;
            float px = 0.f, py = 0.f;
            if (mine) { px = npos[3 * i]; py = npos[3 * i + 1]; }
            const bool in = mine && px >= x1 && px <= x2 && py >= y1 && py <= y2;
            const float dx = px - cx, dy = py - cy;
            const float wgt = (1.0f / (sqrtf(dx * dx + dy * dy) + 1e-6f)) / wsum;
            unsigned long long min_ = __ballot(in);
            while (min_) {
                const int l = __ffsll((long long)min_) - 1; min_ &= min_ - 1;
                acc += nx[(size_t)(base + l) * 64 + c] * __shfl(wgt, l, 64);
            }
        }
    } else {
        int k = min(5, n_clip);
        for (int j = 0; j < k; ++j) acc += nx[(size_t)bi[j] * 64 + c];
        acc = acc / (float)k;
    }
    o[c] = acc;
}

// feats must be zero-filled by the caller (absent objects stay zero = "absent", EventAD.py:229).
extern "C" int eventad_roi_pool(const float* node_x, const float* node_pos, const int32_t* node_batch,
                                const int32_t* n_dev, int n_max, const float* bbox, const int64_t* bbox_batch, int M,
                                int frame, int n_frames, int width, int height, int B, float* feats, void* stream) {
    if (!node_x || !node_pos || !node_batch || !n_dev || !feats || M < 0 || (M > 0 && (!bbox || !bbox_batch)) ||
        frame < 0 || frame >= n_frames || width <= 0 || height <= 0 || B <= 0)
        return EVAD_ERR_ARG;
    if (M == 0) return EVAD_OK;
    hipLaunchKernelGGL(k_roi_pool, dim3(M), dim3(64), 0, (hipStream_t)stream, node_x, node_pos, node_batch, n_dev, n_max,
                       bbox, (const long long*)bbox_batch, M, frame, n_frames, (float)width, (float)height, B, feats);
    return evad_last_error();
}
