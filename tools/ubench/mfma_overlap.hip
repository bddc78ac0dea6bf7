// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?  Workgroups of 8 waves (wave w -> SIMD w % 4), 4 per CU:
// waves 0-3 of a workgroup run role A, waves 4-7 role B.  Roles: 0 idle, 1 MFMA 16x16x32 chain x6, 2 v_add x8, 3 v_exp x8,
// 4 MFMA 32x32x16 x3.   build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_overlap mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float role_mfma(int iters, int lane) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(lane * 0.001f + i); fb[i] = (__bf16)(1.f - i); }
    for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fa, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fb, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, a5, 0, 0, 0);
    }
    return a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0];
}
__device__ __forceinline__ float role_mfma32(int iters, int lane) {
    f32x16 a0, a1, a2;
    for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; }
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(lane * 0.001f + i); fb[i] = (__bf16)(1.f - i); }
    for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fa, a2, 0, 0, 0);
    }
    return a0[0] + a1[0] + a2[0];
}
template <bool EXP>
__device__ __forceinline__ float role_valu(int iters, int lane) {
    float a0 = lane * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    for (int it = 0; it < iters; ++it) {
        if (EXP)
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0f));
    }
    return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(512) void k(float* out, int iters, int roleA, int roleB) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int role = wave < 4 ? roleA : roleB;
    float r = 0.f;
    if (role == 1) r = role_mfma(iters, lane);
    else if (role == 2) r = role_valu<false>(iters, lane);
    else if (role == 3) r = role_valu<true>(iters, lane);
    else if (role == 4) r = role_mfma32(iters, lane);
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

static double run(int roleA, int roleB, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 4;                                   // 4 workgroups per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, iters, roleA, roleB);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, iters, roleA, roleB);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
    const int iters = 100000;
    const char* rn[] = {"idle", "6x MFMA 16x16x32", "8x v_add", "8x v_exp", "3x MFMA 32x32x16"};
    int pairs[][2] = {{1, 0}, {1, 1}, {2, 0}, {2, 2}, {3, 0}, {3, 3}, {4, 0}, {4, 4}, {1, 2}, {1, 3}, {4, 2}, {4, 3}};
    for (auto& p : pairs) {
        const double t = run(p[0], p[1], iters, out);
        printf("waves 0-3: %-18s waves 4-7: %-18s  %8.3f ns per iteration (%.1f cycles at 2.39 GHz)\n", rn[p[0]], rn[p[1]], t / iters * 1e9,
               t / iters * 2.39e9);
    }
    return 0;
}
