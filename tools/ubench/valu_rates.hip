// Issue-rate microbenchmarks for gfx950 (one SIMD's vector pipe under 1..8 resident waves):
//   v_add_f32, v_exp_f32, v_cvt_pk_bf16_f32, the attention step mix (2 MFMA + 8 exp + 4 cvt + 1 MFMA), MFMA alone.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, s0 = acc, s1 = acc, t0 = acc, t1 = acc, t2 = acc;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(a0 + i); fb[i] = (__bf16)(a1 - i); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 8 independent v_add_f32
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0f));
        } else if (MODE == 1) {   // 8 independent v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 2) {   // 8 v_cvt_pk_bf16_f32
            asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                         "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) {   // the attention step: 2 S MFMAs -> 8 exp -> 4 cvt -> PV MFMA (dependent, like the kernel)
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc, 0, 0, 0);
            float e0 = __builtin_amdgcn_exp2f(s0[0]), e1 = __builtin_amdgcn_exp2f(s0[1]), e2 = __builtin_amdgcn_exp2f(s0[2]), e3 = __builtin_amdgcn_exp2f(s0[3]);
            float e4 = __builtin_amdgcn_exp2f(s1[0]), e5 = __builtin_amdgcn_exp2f(s1[1]), e6 = __builtin_amdgcn_exp2f(s1[2]), e7 = __builtin_amdgcn_exp2f(s1[3]);
            bf16x8 p;
            p[0] = (__bf16)e0; p[1] = (__bf16)e1; p[2] = (__bf16)e2; p[3] = (__bf16)e3; p[4] = (__bf16)e4; p[5] = (__bf16)e5; p[6] = (__bf16)e6; p[7] = (__bf16)e7;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, p, acc, 0, 0, 0);
            acc[0] *= 1e-30f; acc[1] *= 1e-30f; acc[2] *= 1e-30f; acc[3] *= 1e-30f;   // keep values finite (4 extra VALU)
        } else if (MODE == 4) {   // 3 independent MFMAs
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, s1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fa, acc, 0, 0, 0);
        } else if (MODE == 5) {   // 8 exp + 8 add interleaved (does a plain VALU op hide under a transcendental?)
            asm volatile("v_exp_f32 %0, %0\n v_add_f32 %4, %4, %8\n v_exp_f32 %1, %1\n v_add_f32 %5, %5, %8\n"
                         "v_exp_f32 %2, %2\n v_add_f32 %6, %6, %8\n v_exp_f32 %3, %3\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0f));
        } else if (MODE == 7) {   // 3 MFMA + 8 independent v_add
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, s1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fa, acc, 0, 0, 0);
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0f));
        } else if (MODE == 8) {   // 3 MFMA + 8 independent v_exp
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, s1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fa, acc, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 9) {   // 3 independent 16x16x16 (bf16_1k) MFMAs
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            s16x4 ha = {1, 2, 3, 4}, hb = {4, 3, 2, 1};
            s0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha, hb, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hb, ha, s1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha, ha, acc, 0, 0, 0);
        } else if (MODE == 10) {  // 6 independent 16x16x32 MFMAs (more in flight per wave)
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, s1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fa, acc, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fb, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, t2, 0, 0, 0);
        } else if (MODE == 6) {   // 4 exp f16 packed input?  v_exp_f16 x8
            asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n"
                         "v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc[0] + acc[1] + acc[2] + acc[3] + s0[0] + s1[0] + t0[0] + t1[0] + t2[0];
}

static long long* g_clk;
static double g_mhz;
template <int MODE>
static double run(int waves_per_simd, int iters, float* out) {
    // 256 CUs x 4 SIMDs; one wave per workgroup, grid = SIMDs x waves per SIMD (the dispatcher spreads them evenly)
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, g_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, g_clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, g_clk, sizeof(h), hipMemcpyDeviceToHost);
    g_mhz = (double)h[0] / (double)h[1] * 100.0;             // shader-clock ticks per 100 MHz wall tick
    return ms * 1e-3;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
    hipMalloc(&g_clk, 2 * sizeof(long long));
    const int iters = 100000;
    const char* names[] = {"8 v_add_f32", "8 v_exp_f32", "8 v_cvt_pk_bf16", "attn step (3 MFMA, 8 exp, 4 cvt, 4 mul)", "3 MFMA 16x16x32", "4 exp + 4 add interleaved", "8 v_exp_f16", "3 MFMA + 8 v_add (independent)", "3 MFMA + 8 v_exp (independent)", "3 MFMA 16x16x16", "6 MFMA 16x16x32"};
    for (int w : {1, 2, 4, 8}) {
        double t[11], mhz[11];
        t[0] = run<0>(w, iters, out); mhz[0] = g_mhz; t[1] = run<1>(w, iters, out); mhz[1] = g_mhz; t[2] = run<2>(w, iters, out); mhz[2] = g_mhz;
        t[3] = run<3>(w, iters, out); mhz[3] = g_mhz; t[4] = run<4>(w, iters, out); mhz[4] = g_mhz; t[5] = run<5>(w, iters, out); mhz[5] = g_mhz;
        t[6] = run<6>(w, iters, out); mhz[6] = g_mhz; t[7] = run<7>(w, iters, out); mhz[7] = g_mhz; t[8] = run<8>(w, iters, out); mhz[8] = g_mhz;
        t[9] = run<9>(w, iters, out); mhz[9] = g_mhz; t[10] = run<10>(w, iters, out); mhz[10] = g_mhz;
        // ns per iteration per SIMD = t / iters / ... ; report per-SIMD time per iteration-of-one-wave: t / (iters * w)
        for (int m = 0; m < 11; ++m)
            printf("waves/SIMD %d  %-42s  %8.3f ns per iteration per wave-slot = %6.1f cycles at the measured %4.0f MHz (clock64 / wall_clock64)\n", w,
                   names[m], t[m] / iters / w * 1e9, t[m] / iters / w * mhz[m] * 1e6, mhz[m]);
    }
    return 0;
}
