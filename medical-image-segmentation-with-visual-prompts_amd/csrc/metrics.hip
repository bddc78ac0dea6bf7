// Evaluation metrics without host round trips: the reference's MeanIoU / DiceCoefficient
// (modules/utils.py:14-64) call .item() 2 x classes times per update; here one launch per batch adds the per-class counts
//   [c][0] = |pred == c & target == c|,  [c][1] = |pred == c|,  [c][2] = |target == c|
// of arg-max(logits) against the label volume into a device-resident int64 table (integer atomics: the result does not
// depend on the order), read once per volume.
#include "common.hpp"

namespace {
constexpr int MAXC = 16;

__global__ __launch_bounds__(256) void k_seg_counts(const float* __restrict__ logits, const float* __restrict__ target, long nvox,
                                                    int C, int channels_last, long vol, unsigned long long* __restrict__ counts) {
    __shared__ unsigned int sm[MAXC * 3];
    for (int i = threadIdx.x; i < MAXC * 3; i += 256) sm[i] = 0u;
    __syncthreads();
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (long)gridDim.x * 256) {
        // first maximum, like torch.argmax
        const long b = v / vol, s = v - b * vol;
        int best = 0;
        float bv = channels_last ? logits[v * C] : logits[(b * C) * vol + s];
        for (int c = 1; c < C; ++c) {
            const float x = channels_last ? logits[v * C + c] : logits[(b * C + c) * vol + s];
            if (x > bv) { bv = x; best = c; }
        }
        const float t = target[v];
        atomicAdd(&sm[best * 3 + 1], 1u);
        for (int c = 0; c < C; ++c)
            if (t == (float)c) { atomicAdd(&sm[c * 3 + 2], 1u); if (best == c) atomicAdd(&sm[c * 3 + 0], 1u); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 3; i += 256)
        if (sm[i]) atomicAdd(&counts[i], (unsigned long long)sm[i]);
}
}  // namespace

extern "C" int mivp_seg_counts(const float* logits, const float* target, int64_t nvox, int32_t C, int32_t channels_last,
                               int64_t vol, void* counts, mivp_stream_t stream) {
    MIVP_REQUIRE(logits && target && counts && nvox > 0 && C >= 1 && C <= MAXC && vol > 0 && nvox % vol == 0);
    const unsigned grid = (unsigned)((nvox + 255) / 256 > 2048 ? 2048 : (nvox + 255) / 256);
    hipLaunchKernelGGL(k_seg_counts, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, (long)nvox, (int)C,
                       (int)channels_last, (long)vol, (unsigned long long*)counts);
    return mivp_check_launch("seg_counts");
}
