// Point sampling of the students/teacher objective (losses/clustered_prototype_loss.py:162-204 ``sample_embedding``):
// the reference's affine_grid (identity) + grid_sample (bilinear, align_corners = False) is a separable linear interpolation
// at the cell centres of a reduced grid, optionally on a jitter-cropped sub-volume.  The latent is the model's channels-last
// bf16 tensor, the coordinate grid a channels-first f32 tensor; the sampled points are f32 [B][N][C].
//
// The per-axis interpolation (two taps per output point; for the backward pass: at most two output points per voxel,
// because consecutive sample positions are at least one voxel apart) is tabulated on the host (mivp_amd/losses.py), so the
// kernels are pure gathers: the backward is a GATHER over the voxels, not a scatter -- no atomics, fixed summation order.
#include "common.hpp"

namespace {

struct SampleAxes {             // device pointers: forward taps per output index, backward taps per voxel coordinate
    const int* lo[3];           // [od_a] absolute voxel coordinate of the lower tap
    const int* hi[3];           // [od_a]
    const float* w[3];          // [od_a] weight of the upper tap
};

template <bool BF16, bool CLAST>
__global__ __launch_bounds__(256) void k_sample_points_fwd(const void* __restrict__ vol, int B, int H, int W, int D, int C,
                                                           int o0, int o1, int o2, SampleAxes ax, float* __restrict__ out) {
    const long total = (long)B * o0 * o1 * o2 * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long p = e / C;
        const int k = (int)(p % o2); p /= o2;
        const int j = (int)(p % o1); p /= o1;
        const int i = (int)(p % o0);
        const long b = p / o0;
        const int h[2] = {ax.lo[0][i], ax.hi[0][i]}, w[2] = {ax.lo[1][j], ax.hi[1][j]}, dd[2] = {ax.lo[2][k], ax.hi[2][k]};
        const float wh = ax.w[0][i], ww = ax.w[1][j], wd = ax.w[2][k];
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const float wt = (a ? wh : 1.f - wh) * (bb ? ww : 1.f - ww) * (cc ? wd : 1.f - wd);
                    const long vox = ((long)h[a] * W + w[bb]) * D + dd[cc];
                    const long idx = CLAST ? (b * ((long)H * W * D) + vox) * C + c : (b * C + c) * ((long)H * W * D) + vox;
                    const float val = BF16 ? (float)reinterpret_cast<const bf16_t*>(vol)[idx] : reinterpret_cast<const float*>(vol)[idx];
                    acc += wt * val;
                }
        out[e] = acc;
    }
}

struct SampleAxesBwd {          // per voxel coordinate of axis a: up to two (output index, weight) pairs, index -1 = none
    const int* i1[3];
    const int* i2[3];
    const float* w1[3];
    const float* w2[3];
};

// gvol [B][H][W][D][C] bf16 (channels-last, the layout of the model's latent): 8 channels per thread
__global__ __launch_bounds__(256) void k_sample_points_bwd(const float* __restrict__ gout, int B, int H, int W, int D, int C,
                                                           int o0, int o1, int o2, SampleAxesBwd ax, bf16_t* __restrict__ gvol) {
    const int C8 = C / 8;
    const long total = (long)B * H * W * D * C8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        long p = e / C8;
        const int z = (int)(p % D); p /= D;
        const int y = (int)(p % W); p /= W;
        const int x = (int)(p % H);
        const long b = p / H;
        const int ih[2] = {ax.i1[0][x], ax.i2[0][x]}, iw[2] = {ax.i1[1][y], ax.i2[1][y]}, id[2] = {ax.i1[2][z], ax.i2[2][z]};
        const float fh[2] = {ax.w1[0][x], ax.w2[0][x]}, fw[2] = {ax.w1[1][y], ax.w2[1][y]}, fd[2] = {ax.w1[2][z], ax.w2[2][z]};
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    if (ih[a] < 0 || iw[bb] < 0 || id[cc] < 0) continue;
                    const float wt = fh[a] * fw[bb] * fd[cc];
                    const float* src = gout + ((b * o0 + ih[a]) * (long)o1 * o2 + (long)iw[bb] * o2 + id[cc]) * C + 8 * c8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc[i] += wt * v0[i]; acc[4 + i] += wt * v1[i]; }
                }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)acc[i];
        st8(gvol + e * 8, o);
    }
}

}  // namespace

extern "C" int mivp_sample_points_fwd(const void* vol, int32_t is_bf16, int32_t channels_last, int32_t B, int32_t H, int32_t W,
                                      int32_t D, int32_t C, const int32_t* out_dims, const int32_t* const* lo,
                                      const int32_t* const* hi, const float* const* w, float* out, mivp_stream_t stream) {
    MIVP_REQUIRE(vol && out_dims && lo && hi && w && out);
    MIVP_REQUIRE(B > 0 && H > 0 && W > 0 && D > 0 && C > 0);
    MIVP_REQUIRE(out_dims[0] > 0 && out_dims[1] > 0 && out_dims[2] > 0);
    SampleAxes ax;
    for (int a = 0; a < 3; ++a) { ax.lo[a] = lo[a]; ax.hi[a] = hi[a]; ax.w[a] = w[a]; MIVP_REQUIRE(lo[a] && hi[a] && w[a]); }
    const long total = (long)B * out_dims[0] * out_dims[1] * out_dims[2] * C;
    const unsigned grid = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
#define SP_FWD(BF, CL) hipLaunchKernelGGL((k_sample_points_fwd<BF, CL>), dim3(grid), dim3(256), 0, st, vol, (int)B, (int)H, (int)W, \
                                          (int)D, (int)C, (int)out_dims[0], (int)out_dims[1], (int)out_dims[2], ax, out)
    if (is_bf16) { if (channels_last) SP_FWD(true, true); else SP_FWD(true, false); }
    else { if (channels_last) SP_FWD(false, true); else SP_FWD(false, false); }
#undef SP_FWD
    return mivp_check_launch("sample_points_fwd");
}

extern "C" int mivp_sample_points_bwd(const float* gout, int32_t B, int32_t H, int32_t W, int32_t D, int32_t C,
                                      const int32_t* out_dims, const int32_t* const* i1, const int32_t* const* i2,
                                      const float* const* w1, const float* const* w2, void* gvol, mivp_stream_t stream) {
    MIVP_REQUIRE(gout && out_dims && i1 && i2 && w1 && w2 && gvol);
    MIVP_REQUIRE(B > 0 && H > 0 && W > 0 && D > 0 && C > 0 && C % 8 == 0);
    SampleAxesBwd ax;
    for (int a = 0; a < 3; ++a) { ax.i1[a] = i1[a]; ax.i2[a] = i2[a]; ax.w1[a] = w1[a]; ax.w2[a] = w2[a]; MIVP_REQUIRE(i1[a] && i2[a] && w1[a] && w2[a]); }
    const long total = (long)B * H * W * D * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 > 262144 ? 262144 : (total + 255) / 256);
    hipLaunchKernelGGL(k_sample_points_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, gout, (int)B, (int)H, (int)W, (int)D,
                       (int)C, (int)out_dims[0], (int)out_dims[1], (int)out_dims[2], ax, (bf16_t*)gvol);
    return mivp_check_launch("sample_points_bwd");
}

// ---------------------------------------------------------------------------------------------
// Multi-tensor optimizer kernels: AdamW over a list of parameter tensors with per-tensor hyper-parameters
// (the reference's two parameter groups: students_teacher.py:27-68) and the EMA teacher update
// (momentum_model.py:27-36), each ONE launch per step however many tensors there are.
// A chunk table (tensor id, element offset) maps workgroups to 1024-element pieces; per-tensor metadata sits in
// device memory and is refreshed only when a learning rate changes (the schedule multiplies lr on the host side).
// ---------------------------------------------------------------------------------------------
namespace {

struct AdamTensor {             // 40 bytes; static across steps (the gradient pointers change and travel separately)
    float* p; float* m; float* v;
    long n;
    int group; int pad;
};
struct AdamGroup { float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, pad; };   // bias corrections precomputed per step
struct AdamGroups { AdamGroup g[8]; };                          // by value in the kernel arguments: no per-step upload
constexpr int ADAM_GRADS_PER_LAUNCH = 384;
struct AdamGrads { const float* g[ADAM_GRADS_PER_LAUNCH]; };    // gradient pointers change every backward: kernel arguments too

constexpr int OPT_CHUNK = 1024;

// DEV: the per-group hyper-parameters come from device memory (``groups_dev``: [8] AdamGroup) instead of the kernel
// arguments, so that a launch recorded in a HIP graph sees the learning rate and bias corrections of the step it is
// REPLAYED in (train.GraphedStep refreshes the buffer before every replay).
template <bool DEV>
__global__ __launch_bounds__(256) void k_adamw_multi(const AdamTensor* __restrict__ tensors, AdamGrads grads, int grad_base,
                                                     AdamGroups groups, const AdamGroup* __restrict__ groups_dev,
                                                     const int2* __restrict__ chunks) {
    const int2 ck = chunks[blockIdx.x];
    const AdamTensor t = tensors[ck.x];
    const float* __restrict__ tg = grads.g[ck.x - grad_base];
    const AdamGroup h = DEV ? groups_dev[t.group] : groups.g[t.group];
    const long base = (long)ck.y * OPT_CHUNK;
    // torch.optim.AdamW (single-tensor formulas, same operation order, so the result is bit-equal on f32):
    //   p *= 1 - lr * wd ; m = m + (g - m) * (1 - b1)  [lerp] ; v = v * b2 + g * g * (1 - b2)
    //   denom = sqrt(v) / sqrt(bias_correction2) + eps ; p -= (lr / bias_correction1) * m / denom
    const float step_size = h.lr / h.bc1;
#pragma unroll
    for (int u = 0; u < OPT_CHUNK / 256; ++u) {
        const long i = base + threadIdx.x + 256 * u;
        if (i >= t.n) break;
        const float g = tg[i];
        float p = t.p[i], m = t.m[i], v = t.v[i];
        p = p * (1.f - h.lr * h.weight_decay);
        m = m + (g - m) * (1.f - h.beta1);
        v = v * h.beta2 + (g * g) * (1.f - h.beta2);
        const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
        p = p - step_size * (m / denom);
        t.p[i] = p; t.m[i] = m; t.v[i] = v;
    }
}

struct EmaTensor { float* teacher; const float* student; long n; };

__global__ __launch_bounds__(256) void k_ema_multi(const EmaTensor* __restrict__ tensors, const int2* __restrict__ chunks, float tau) {
    const int2 ck = chunks[blockIdx.x];
    const EmaTensor t = tensors[ck.x];
    const long base = (long)ck.y * OPT_CHUNK;
    const float om = 1.f - tau;
#pragma unroll
    for (int u = 0; u < OPT_CHUNK / 256; ++u) {
        const long i = base + threadIdx.x + 256 * u;
        if (i >= t.n) break;
        t.teacher[i] = tau * t.teacher[i] + om * t.student[i];          // two roundings, as the reference's tensor expression
    }
}

}  // namespace

static int adamw_launch(const void* tensors, const void* const* grads, int32_t n_tensors, const int32_t* chunk_begin,
                        const float* groups_host, const float* groups_dev, int32_t n_groups, const void* chunks,
                        mivp_stream_t stream) {
    AdamGroups gs;
    for (int i = 0; i < 8; ++i) {
        const float* src = groups_host ? groups_host + 8 * (i < n_groups ? i : 0) : nullptr;
        gs.g[i] = src ? AdamGroup{src[0], src[1], src[2], src[3], src[4], src[5], src[6], 0.f} : AdamGroup{};
    }
    // gradient pointers (and, in the plain form, the hyper-parameters) travel as kernel arguments: nothing is uploaded per step
    for (int base = 0; base < n_tensors; base += ADAM_GRADS_PER_LAUNCH) {
        const int cnt = n_tensors - base < ADAM_GRADS_PER_LAUNCH ? n_tensors - base : ADAM_GRADS_PER_LAUNCH;
        AdamGrads gp;
        for (int i = 0; i < ADAM_GRADS_PER_LAUNCH; ++i) gp.g[i] = (const float*)grads[base + (i < cnt ? i : 0)];
        const int c0 = chunk_begin[base], c1 = chunk_begin[base + cnt];
        if (c1 <= c0) continue;
        if (groups_dev)
            hipLaunchKernelGGL(k_adamw_multi<true>, dim3((unsigned)(c1 - c0)), dim3(256), 0, (hipStream_t)stream,
                               (const AdamTensor*)tensors, gp, base, gs, (const AdamGroup*)groups_dev, (const int2*)chunks + c0);
        else
            hipLaunchKernelGGL(k_adamw_multi<false>, dim3((unsigned)(c1 - c0)), dim3(256), 0, (hipStream_t)stream,
                               (const AdamTensor*)tensors, gp, base, gs, (const AdamGroup*)nullptr, (const int2*)chunks + c0);
    }
    return mivp_check_launch("adamw_multi");
}

extern "C" int mivp_adamw_multi(const void* tensors, const void* const* grads, int32_t n_tensors, const int32_t* chunk_begin,
                                const float* groups, int32_t n_groups, const void* chunks, mivp_stream_t stream) {
    MIVP_REQUIRE(tensors && grads && groups && chunks && chunk_begin && n_tensors > 0 && n_groups >= 1 && n_groups <= 8);
    return adamw_launch(tensors, grads, n_tensors, chunk_begin, groups, nullptr, n_groups, chunks, stream);
}

extern "C" int mivp_adamw_multi_dev(const void* tensors, const void* const* grads, int32_t n_tensors, const int32_t* chunk_begin,
                                    const float* groups_dev, int32_t n_groups, const void* chunks, mivp_stream_t stream) {
    MIVP_REQUIRE(tensors && grads && groups_dev && chunks && chunk_begin && n_tensors > 0 && n_groups >= 1 && n_groups <= 8);
    return adamw_launch(tensors, grads, n_tensors, chunk_begin, nullptr, groups_dev, n_groups, chunks, stream);
}

// up to 64 floats from the HOST into device memory as kernel arguments of a one-wave launch: stream-ordered, no pinned
// staging buffer whose reuse would have to be fenced (the per-replay refresh of a graph's scalars)
namespace {
struct FloatPack { float v[64]; };
__global__ void k_store_floats(float* __restrict__ dst, FloatPack vals, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}
}  // namespace

extern "C" int mivp_store_floats(float* dst, const float* host_values, int32_t n, mivp_stream_t stream) {
    MIVP_REQUIRE(dst && host_values && n > 0 && n <= 64);
    FloatPack pk;
    for (int i = 0; i < 64; ++i) pk.v[i] = i < n ? host_values[i] : 0.f;
    hipLaunchKernelGGL(k_store_floats, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, pk, (int)n);
    return mivp_check_launch("store_floats");
}

extern "C" int mivp_ema_multi(const void* tensors, const void* chunks, int32_t n_chunks, float tau, mivp_stream_t stream) {
    MIVP_REQUIRE(tensors && chunks && n_chunks > 0);
    hipLaunchKernelGGL(k_ema_multi, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (const EmaTensor*)tensors,
                       (const int2*)chunks, tau);
    return mivp_check_launch("ema_multi");
}

extern "C" int mivp_sizeof_opt(int which) { return which == 0 ? (int)sizeof(AdamTensor) : which == 1 ? (int)sizeof(AdamGroup) : which == 2 ? (int)sizeof(EmaTensor) : -1; }
