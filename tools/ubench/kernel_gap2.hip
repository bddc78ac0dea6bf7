// Second probe of the 5.8 us gaps that follow four kernels of the cfg1 step: copy kernels with the read / write
// footprints of those kernels (reads R MB from one buffer, writes W MB to another, 8-byte or 16-byte pieces), each followed by a tiny kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int R, int W, int PIECE>
__global__ __launch_bounds__(256) void mover(const f32x4* __restrict__ src, char* __restrict__ dst, long nr, long nw) {
    const long stride = (long)gridDim.x * 256, t = (long)blockIdx.x * 256 + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = t; i < nr; i += stride) acc += src[i];
    if (PIECE == 16) { for (long i = t; i < nw; i += stride) reinterpret_cast<f32x4*>(dst)[i] = acc + (float)i; }
    else { for (long i = t; i < 2 * nw; i += stride) { const f32x2 v = {acc[0] + (float)i, acc[1]}; reinterpret_cast<f32x2*>(dst)[i] = v; } }
}
template <int R, int W, int PIECE>
__global__ void tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
template <int R, int W, int PIECE>
void run(const f32x4* a, char* b, float* small, hipStream_t st, int grid) {
    for (int it = 0; it < 10; ++it) {
        hipLaunchKernelGGL((mover<R, W, PIECE>), dim3(grid), dim3(256), 0, st, a, b, (long)R * 65536, (long)W * 65536);
        hipLaunchKernelGGL((tiny<R, W, PIECE>), dim3(1), dim3(64), 0, st, small);
    }
}
int main() {
    f32x4* a; char* b; float* small;
    hipMalloc(&a, 256L << 20); hipMalloc(&b, 256L << 20); hipMalloc(&small, 4096);
    hipMemset(a, 0, 256L << 20); hipMemset(small, 0, 4096);
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 2; ++rep) {
        run<42, 139, 8>(a, b, small, st, 3773);    // stage-0 QKV: reads x, writes q | k | v in 8-byte pieces
        run<42, 139, 16>(a, b, small, st, 3773);
        run<139, 46, 8>(a, b, small, st, 5488);    // stage-0 attention
        run<92, 42, 16>(a, b, small, st, 3773);    // proj + MLP (no gap in the step)
        run<47, 127, 16>(a, b, small, st, 9216);   // upsample + concat of the last decoder stage
        run<200, 42, 16>(a, b, small, st, 768);    // last-stage conv
        run<16, 16, 16>(a, b, small, st, 1024);
    }
    hipStreamSynchronize(st);
    printf("done\n");
    return 0;
}
