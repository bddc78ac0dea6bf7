// LDS access patterns of the fused attention backward (csrc/swin_bwd_fused.hip): cycles per wave-instruction when four waves
// (one per SIMD) of a workgroup issue the same pattern back to back.  A conflict-free 8-byte-per-lane access moves 512 B.
// build: hipcc --offload-arch=gfx950 -O3 -w -o lds_patterns lds_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int kr_off(int row, int elem) { return row * 64 + 16 * ((elem >> 3) ^ ((0 - (row >> 2)) & 3)) + 2 * (elem & 7); }
__device__ __forceinline__ int oimg_off(int row, int byte) { return row * 32 + (byte ^ (((row >> 3) & 1) << 4)); }
__device__ __forceinline__ int exch_off(int tile, int key, int quad) { return tile * 512 + 128 * quad + 8 * (key ^ ((quad >> 1) << 3)); }

// kind: 0 tr-read b64, 1 plain read b64, 2 plain read b128, 3 write b64, 4 write b128
template <int KIND>
__global__ __launch_bounds__(256) void k(int pattern, int iters, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    int off = 0;
    switch (pattern) {
        case 0: off = (4 * g + (r >> 2)) * 32 + 8 * (r & 3); break;                  // plain 32-byte rows, transposing read (fwd V, bwd K rows)
        case 1: off = oimg_off(4 * g + (r >> 2), 8 * (r & 3)); break;                // o_tr
        case 2: off = oimg_off(r, 8 * g); break;                                      // o_rd (g < 2 meaningful; here all)
        case 3: off = kr_off(4 * g + (r >> 2), 4 * (r & 3)); break;                  // q_tr
        case 4: off = kr_off(r, 8 * g); break;                                        // q_rd (b128)
        case 5: off = exch_off(wave, r, g); break;                                    // ex_wr
        case 6: off = exch_off(wave, 4 * g + (r >> 2), r & 3); break;                // ex_tr
        case 7: off = wave * 1024 + r * 64 + 16 * g; break;                           // dqp write / read (b128)
        case 8: off = r * 32 + 8 * g; break;                                          // plain 32-byte rows, row read b64 (no swizzle)
        case 9: off = (4 * g + (r >> 2)) * 64 + 8 * (r & 3); break;                  // plain 64-byte rows, transposing read
        case 10: off = 16 * g; break;                                                  // lse / delta rows: f32x4 per g, all r the same
        case 11: off = wave * 1024 + r * 64 + 16 * g + 4096 * 0; break;
    }
    const char* p = smem + off;
    float acc = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(p));                                // (keeps the loads in the loop)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char* q = p + 2048 * (u & 3);
            if (KIND == 0) { bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q); acc += (float)v[0]; }
            else if (KIND == 1) { bf16x4 v = *reinterpret_cast<const bf16x4*>(q); acc += (float)v[0]; }
            else if (KIND == 2) { bf16x8 v = *reinterpret_cast<const bf16x8*>(q); acc += (float)v[0]; }
            else if (KIND == 3) { bf16x4 v; v[0] = (__bf16)acc; v[1] = v[0]; v[2] = v[0]; v[3] = v[0]; *reinterpret_cast<bf16x4*>(const_cast<char*>(q)) = v; }
            else { f32x4 v = {acc, acc, acc, acc}; *reinterpret_cast<f32x4*>(const_cast<char*>(q)) = v; }
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
static double run(int pattern) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 65536, 0, pattern, iters, out, cyc);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    hipFree(out); hipFree(cyc);
    return s / 256 / (iters * 8.0);      // clock64 ticks per wave-instruction (four waves share the CU's LDS pipe)
}

int main() {
    printf("clock64 ticks per wave-instruction (4 waves per CU issuing together; lower = fewer bank conflicts)\n");
    printf("tr-read  plain 32B rows (fwd V / bwd K rows) %6.2f\n", run<0>(0));
    printf("tr-read  Oimg half-swap (o_tr)               %6.2f\n", run<0>(1));
    printf("b64 read Oimg half-swap (o_rd)               %6.2f\n", run<1>(2));
    printf("b64 read plain 32B rows (row r, 8g)          %6.2f\n", run<1>(8));
    printf("tr-read  Qimg swizzled 64B rows (q_tr)       %6.2f\n", run<0>(3));
    printf("tr-read  plain 64B rows                      %6.2f\n", run<0>(9));
    printf("b128 read Qimg swizzled (q_rd)               %6.2f\n", run<2>(4));
    printf("b64 write exch slot (ex_wr)                  %6.2f\n", run<3>(5));
    printf("tr-read  exch slot (ex_tr)                   %6.2f\n", run<0>(6));
    printf("b128 write dqp (wave*1024 + r*64 + 16g)      %6.2f\n", run<4>(7));
    printf("b128 read  dqp                               %6.2f\n", run<2>(7));
    printf("b128 read  lse / delta (broadcast rows)      %6.2f\n", run<2>(10));
    return 0;
}
