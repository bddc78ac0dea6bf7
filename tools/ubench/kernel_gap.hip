// Gap between a kernel that leaves dirty lines in the L2s and the next kernel on the same stream (rocprofv3 --kernel-trace):
//   writer<MB, NT> writes MB megabytes (NT: nontemporal stores), then tiny<MB, NT> runs.  tools/ubench/kernel_gap_summary.py
//   prints tiny.start - writer.end per (MB, NT).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MB, int NT>
__global__ __launch_bounds__(256) void writer(f32x4* __restrict__ p, long n, float v) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const f32x4 val = {v, v + 1.f, v + 2.f, (float)i};
        if (NT) __builtin_nontemporal_store(val, p + i);
        else p[i] = val;
    }
}
template <int MB, int NT>
__global__ void tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
template <int MB, int NT>
void run(f32x4* buf, float* small, hipStream_t st) {
    const long n = (long)MB * 1024 * 1024 / 16;
    for (int it = 0; it < 12; ++it) {
        hipLaunchKernelGGL((writer<MB, NT>), dim3(4096), dim3(256), 0, st, buf, n, (float)it);
        hipLaunchKernelGGL((tiny<MB, NT>), dim3(1), dim3(64), 0, st, small);
    }
}
int main() {
    f32x4* buf; float* small;
    hipMalloc(&buf, 512L << 20); hipMalloc(&small, 4096); hipMemset(small, 0, 4096);
    hipStream_t st; hipStreamCreate(&st);
    // captured into a graph: the bench replays graphs, and eager launches add host gaps of their own
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    run<4, 0>(buf, small, st); run<16, 0>(buf, small, st); run<32, 0>(buf, small, st); run<48, 0>(buf, small, st);
    run<64, 0>(buf, small, st); run<128, 0>(buf, small, st); run<256, 0>(buf, small, st);
    run<32, 1>(buf, small, st); run<64, 1>(buf, small, st); run<128, 1>(buf, small, st); run<256, 1>(buf, small, st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    printf("done\n");
    return 0;
}
