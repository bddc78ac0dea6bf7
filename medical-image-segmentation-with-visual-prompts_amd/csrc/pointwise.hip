// 1x1x1 convolution with few output channels (the last layer of the reconstruction head, swin_unetr.py:204-209:
// nn.Conv3d(C, input_channels, kernel_size=1)) and global average pooling (nn.AdaptiveAvgPool3d((1,1,1)) in front of
// the rotation / contrastive heads, swin_unetr.py:72-81).  Memory-bound streaming kernels on channels-last bf16.
#include "common.hpp"

namespace {
constexpr int PW_MAXCO = 4;

// y[v][co] (f32) = bias[co] + sum_c x[v][c] * w[co][c]          one thread per voxel
template <int COUT>
__global__ __launch_bounds__(256) void k_pointwise_fwd(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, long n_vox, int C, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);               // [COUT][C]
    for (int i = threadIdx.x; i < COUT * C; i += 256) wl[i] = w[i];
    __syncthreads();
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n_vox; v += (long)gridDim.x * 256) {
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
        for (int c = 0; c < C; c += 8) {
            const bf16x8 xv = ld8(x + v * C + c);
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[co] += (float)xv[i] * wl[co * C + c + i];
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) y[v * COUT + co] = acc[co];
    }
}

// dx[v][c] (bf16) = sum_co dy[v][co] * w[co][c];  dyb[v][0..3] (bf16, zero padded) = dy for the weight gradient GEMM
template <int COUT>
__global__ __launch_bounds__(256) void k_pointwise_bwd(const float* __restrict__ dy, const float* __restrict__ w, long n_vox, int C,
                                                       bf16_t* __restrict__ dx, bf16_t* __restrict__ dyb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < COUT * C; i += 256) wl[i] = w[i];
    __syncthreads();
    const int G = C / 8;
    const long items = n_vox * G;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long v = (unsigned)it / (unsigned)G;                 // host checks items < 2^31
        const int cg = (int)(it - v * G);
        float g[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) g[co] = dy[v * COUT + co];
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = 0.f;
#pragma unroll
            for (int co = 0; co < COUT; ++co) a += g[co] * wl[co * C + cg * 8 + i];
            o[i] = (bf16_t)a;
        }
        st8(dx + v * C + cg * 8, o);
        if (cg == 0 && dyb) {
            bf16x4 p = zero4();
#pragma unroll
            for (int co = 0; co < COUT; ++co) p[co] = (bf16_t)g[co];
            st4(dyb + v * 4, p);
        }
    }
}
}  // namespace

extern "C" int mivp_pointwise_fwd(const void* x, const float* w, const float* bias, int64_t n_vox, int32_t C, int32_t Cout,
                                  float* y, mivp_stream_t stream) {
    MIVP_REQUIRE(x && w && y && n_vox > 0 && C > 0 && C % 8 == 0 && Cout >= 1 && Cout <= PW_MAXCO);
    const unsigned grid = (unsigned)((n_vox + 255) / 256 > 8192 ? 8192 : (n_vox + 255) / 256);
    const size_t lds = (size_t)Cout * C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define PW_F(CO) hipLaunchKernelGGL((k_pointwise_fwd<CO>), dim3(grid), dim3(256), lds, st, (const bf16_t*)x, w, bias, (long)n_vox, (int)C, y)
    switch (Cout) { case 1: PW_F(1); break; case 2: PW_F(2); break; case 3: PW_F(3); break; default: PW_F(4); break; }
#undef PW_F
    return mivp_check_launch("pointwise_fwd");
}

extern "C" int mivp_pointwise_bwd(const float* dy, const float* w, int64_t n_vox, int32_t C, int32_t Cout, void* dx, void* dyb,
                                  mivp_stream_t stream) {
    MIVP_REQUIRE(dy && w && dx && n_vox > 0 && C > 0 && C % 8 == 0 && Cout >= 1 && Cout <= PW_MAXCO);
    const long items = n_vox * (C / 8);
    MIVP_REQUIRE(items < (1L << 31));                            // 32-bit decode in the kernel
    const unsigned grid = (unsigned)((items + 255) / 256 > 16384 ? 16384 : (items + 255) / 256);
    const size_t lds = (size_t)Cout * C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define PW_B(CO) hipLaunchKernelGGL((k_pointwise_bwd<CO>), dim3(grid), dim3(256), lds, st, dy, w, (long)n_vox, (int)C, (bf16_t*)dx, (bf16_t*)dyb)
    switch (Cout) { case 1: PW_B(1); break; case 2: PW_B(2); break; case 3: PW_B(3); break; default: PW_B(4); break; }
#undef PW_B
    return mivp_check_launch("pointwise_bwd");
}
