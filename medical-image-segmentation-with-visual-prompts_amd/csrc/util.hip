// Error plumbing, tiny utilities and the MFMA lane-map self test.
#include "common.hpp"
#include <string.h>

static thread_local char g_err[256] = "ok";

void mivp_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "unknown", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int mivp_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        mivp_set_error(buf);
        return MIVP_ELAUNCH;
    }
    return MIVP_OK;
}

extern "C" int mivp_abi_version(void) { return 12; }
extern "C" const char* mivp_last_error(void) { return g_err; }

__global__ void k_cast_f32_bf16(const float* __restrict__ in, long n, bf16_t* __restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (bf16_t)in[i];
}

extern "C" int mivp_cast_f32_bf16(const float* in, int64_t n, void* out, mivp_stream_t stream) {
    MIVP_REQUIRE(in && out && n >= 0);
    if (n == 0) return MIVP_OK;
    const unsigned grid = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(k_cast_f32_bf16, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (long)n, (bf16_t*)out);
    return mivp_check_launch("cast_f32_bf16");
}

__global__ void k_add_bf16(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, long n8, bf16_t* __restrict__ y) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) {
        bf16x8 va = ld8(a + 8 * i), vb = ld8(b + 8 * i), vo;
#pragma unroll
        for (int j = 0; j < 8; ++j) vo[j] = (bf16_t)((float)va[j] + (float)vb[j]);
        st8(y + 8 * i, vo);
    }
}

extern "C" int mivp_add_bf16(const void* a, const void* b, int64_t n, void* y, mivp_stream_t stream) {
    MIVP_REQUIRE(a && b && y && n >= 0 && n % 8 == 0);
    if (n == 0) return MIVP_OK;
    const long n8 = n / 8;
    const unsigned grid = (unsigned)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
    hipLaunchKernelGGL(k_add_bf16, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, n8,
                       (bf16_t*)y);
    return mivp_check_launch("add_bf16");
}

// out[r] = sum_i in[i*rows + r].  Block = 32 columns x 32 row-slices (1024 threads): a slice walks rows
// slice, slice+32, ... in a fixed order, then the 32 slice sums are combined by a fixed-shape tree in LDS:
// deterministic, and rows/32 blocks keep enough CUs busy for the [1372 x 3072] prompt-gradient partials.
__global__ __launch_bounds__(1024) void k_reduce_rows(const float* __restrict__ in, long n, long rows, float* __restrict__ out) {
    __shared__ float part[32][33];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const long rr = (long)blockIdx.x * 32 + col;
    float a0 = 0.f, a1 = 0.f;
    if (rr < rows) {
        long i = slice;
        // eight independent loads per round trip, added in the same order as the two-at-a-time tail loop (same bits)
        for (; i + 224 < n; i += 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[(i + 32 * u) * rows + rr];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 += v[u]; a1 += v[u + 1]; }
        }
        for (; i + 32 < n; i += 64) { a0 += in[i * rows + rr]; a1 += in[(i + 32) * rows + rr]; }
        if (i < n) a0 += in[i * rows + rr];
    }
    part[slice][col] = a0 + a1;
    __syncthreads();
    for (int h = 16; h > 0; h >>= 1) {
        if (slice < h) part[slice][col] += part[slice + h][col];
        __syncthreads();
    }
    if (slice == 0 && rr < rows) out[rr] = part[0][col];
}

// prompt-token bias scores (relative_positional_encoding.py:128-135): ts[h][t] = scale * <W[h], E[t]> and its gradients.
// heads <= 64, Np <= 64 tokens, embed dim e: one workgroup (the torch form was ~8 tiny launches per prompted block and step)
// operands are staged in LDS first: a thread walking global memory through a 64-long dependent chain costs ~50 us
__global__ __launch_bounds__(256) void k_token_scores_fwd(const float* __restrict__ W, const float* __restrict__ E, int heads,
                                                          int np, int e, float scale, float* __restrict__ ts) {
    extern __shared__ __attribute__((aligned(16))) char smem_ts[];
    float* Ws = reinterpret_cast<float*>(smem_ts);             // [heads][e + 1]
    float* Es = Ws + heads * (e + 1);                          // [np][e + 1]
    for (int i = threadIdx.x; i < heads * e; i += 256) Ws[(i / e) * (e + 1) + i % e] = W[i];
    for (int i = threadIdx.x; i < np * e; i += 256) Es[(i / e) * (e + 1) + i % e] = E[i];
    __syncthreads();
    for (int i = threadIdx.x; i < heads * np; i += 256) {
        const int h = i / np, t = i - h * np;
        float acc = 0.f;
        for (int k = 0; k < e; ++k) acc += Ws[h * (e + 1) + k] * Es[t * (e + 1) + k];
        ts[i] = acc * scale;
    }
}
__global__ __launch_bounds__(256) void k_token_scores_bwd(const float* __restrict__ dts, const float* __restrict__ W,
                                                          const float* __restrict__ E, int heads, int np, int e, float scale,
                                                          float* __restrict__ dW, float* __restrict__ dE) {
    extern __shared__ __attribute__((aligned(16))) char smem_ts[];
    float* Ws = reinterpret_cast<float*>(smem_ts);             // [heads][e]
    float* Es = Ws + heads * e;                                // [np][e]
    float* Ds = Es + np * e;                                   // [heads][np]
    for (int i = threadIdx.x; i < heads * e; i += 256) Ws[i] = W[i];
    for (int i = threadIdx.x; i < np * e; i += 256) Es[i] = E[i];
    for (int i = threadIdx.x; i < heads * np; i += 256) Ds[i] = dts[i];
    __syncthreads();
    for (int i = threadIdx.x; i < heads * e; i += 256) {       // dW[h][k] = scale * sum_t dts[h][t] E[t][k]
        const int h = i / e, k = i - h * e;
        float acc = 0.f;
        for (int t = 0; t < np; ++t) acc += Ds[h * np + t] * Es[t * e + k];
        dW[i] = acc * scale;
    }
    for (int i = threadIdx.x; i < np * e; i += 256) {          // dE[t][k] = scale * sum_h dts[h][t] W[h][k]
        const int t = i / e, k = i - t * e;
        float acc = 0.f;
        for (int h = 0; h < heads; ++h) acc += Ds[h * np + t] * Ws[h * e + k];
        dE[i] = acc * scale;
    }
}

// The same for several blocks in ONE launch (blockIdx.x = block): these kernels last 5-10 us each whatever they compute, and
// a prompt-tuning step runs one per prompted block and direction.
struct TokenScoreJob { const float* W; const float* E; const float* dts; float* ts; float* dW; float* dE; int heads, np, e; float scale; };
struct TokenScoreJobs { TokenScoreJob job[16]; };
__global__ __launch_bounds__(256) void k_token_scores_fwd_multi(TokenScoreJobs jobs) {
    const TokenScoreJob& jb = jobs.job[blockIdx.x];
    const float* __restrict__ W = jb.W;
    const float* __restrict__ E = jb.E;
    const int heads = jb.heads, np = jb.np, e = jb.e;
    extern __shared__ __attribute__((aligned(16))) char smem_ts[];
    float* Ws = reinterpret_cast<float*>(smem_ts);             // [heads][e + 1]
    float* Es = Ws + heads * (e + 1);                          // [np][e + 1]
    for (int i = threadIdx.x; i < heads * e; i += 256) Ws[(i / e) * (e + 1) + i % e] = W[i];
    for (int i = threadIdx.x; i < np * e; i += 256) Es[(i / e) * (e + 1) + i % e] = E[i];
    __syncthreads();
    for (int i = threadIdx.x; i < heads * np; i += 256) {
        const int h = i / np, t = i - h * np;
        float acc = 0.f;
        for (int k = 0; k < e; ++k) acc += Ws[h * (e + 1) + k] * Es[t * (e + 1) + k];
        jb.ts[i] = acc * jb.scale;
    }
}
__global__ __launch_bounds__(256) void k_token_scores_bwd_multi(TokenScoreJobs jobs) {
    const TokenScoreJob& jb = jobs.job[blockIdx.x];
    const int heads = jb.heads, np = jb.np, e = jb.e;
    const float scale = jb.scale;
    extern __shared__ __attribute__((aligned(16))) char smem_ts[];
    float* Ws = reinterpret_cast<float*>(smem_ts);             // [heads][e]
    float* Es = Ws + heads * e;                                // [np][e]
    float* Ds = Es + np * e;                                   // [heads][np]
    for (int i = threadIdx.x; i < heads * e; i += 256) Ws[i] = jb.W[i];
    for (int i = threadIdx.x; i < np * e; i += 256) Es[i] = jb.E[i];
    for (int i = threadIdx.x; i < heads * np; i += 256) Ds[i] = jb.dts ? jb.dts[i] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < heads * e; i += 256) {       // dW[h][k] = scale * sum_t dts[h][t] E[t][k]
        const int h = i / e, k = i - h * e;
        float acc = 0.f;
        for (int t = 0; t < np; ++t) acc += Ds[h * np + t] * Es[t * e + k];
        jb.dW[i] = acc * scale;
    }
    for (int i = threadIdx.x; i < np * e; i += 256) {          // dE[t][k] = scale * sum_h dts[h][t] W[h][k]
        const int t = i / e, k = i - t * e;
        float acc = 0.f;
        for (int h = 0; h < heads; ++h) acc += Ds[h * np + t] * Ws[h * e + k];
        jb.dE[i] = acc * scale;
    }
}

/* n <= 16 blocks; arrays of n entries each (host memory).  bwd: dts[i] may be NULL (no gradient reached that block). */
extern "C" int mivp_token_scores_fwd_multi(int32_t n, const float* const* W, const float* const* E, const int32_t* heads,
                                           const int32_t* np, int32_t e, const float* scale, float* const* ts,
                                           mivp_stream_t stream) {
    MIVP_REQUIRE(n > 0 && n <= 16 && W && E && heads && np && scale && ts && e > 0);
    TokenScoreJobs jobs;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        MIVP_REQUIRE(W[i] && E[i] && ts[i] && heads[i] > 0 && np[i] > 0);
        jobs.job[i] = TokenScoreJob{W[i], E[i], nullptr, ts[i], nullptr, nullptr, heads[i], np[i], e, scale[i]};
        const size_t need = (size_t)(heads[i] + np[i]) * (e + 1) * sizeof(float);
        lds = need > lds ? need : lds;
    }
    MIVP_REQUIRE(lds <= 64 * 1024);
    hipLaunchKernelGGL(k_token_scores_fwd_multi, dim3(n), dim3(256), lds, (hipStream_t)stream, jobs);
    return mivp_check_launch("token_scores_fwd_multi");
}
extern "C" int mivp_token_scores_bwd_multi(int32_t n, const float* const* dts, const float* const* W, const float* const* E,
                                           const int32_t* heads, const int32_t* np, int32_t e, const float* scale,
                                           float* const* dW, float* const* dE, mivp_stream_t stream) {
    MIVP_REQUIRE(n > 0 && n <= 16 && dts && W && E && heads && np && scale && dW && dE && e > 0);
    TokenScoreJobs jobs;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        MIVP_REQUIRE(W[i] && E[i] && dW[i] && dE[i] && heads[i] > 0 && np[i] > 0);
        jobs.job[i] = TokenScoreJob{W[i], E[i], dts[i], nullptr, dW[i], dE[i], heads[i], np[i], e, scale[i]};
        const size_t need = ((size_t)(heads[i] + np[i]) * e + (size_t)heads[i] * np[i]) * sizeof(float);
        lds = need > lds ? need : lds;
    }
    MIVP_REQUIRE(lds <= 64 * 1024);
    hipLaunchKernelGGL(k_token_scores_bwd_multi, dim3(n), dim3(256), lds, (hipStream_t)stream, jobs);
    return mivp_check_launch("token_scores_bwd_multi");
}

extern "C" int mivp_token_scores_fwd(const float* W, const float* E, int32_t heads, int32_t np, int32_t e, float scale, float* ts,
                                     mivp_stream_t stream) {
    MIVP_REQUIRE(W && E && ts && heads > 0 && np > 0 && e > 0);
    const size_t lds = (size_t)(heads + np) * (e + 1) * sizeof(float);
    MIVP_REQUIRE(lds <= 64 * 1024);
    hipLaunchKernelGGL(k_token_scores_fwd, dim3(1), dim3(256), lds, (hipStream_t)stream, W, E, heads, np, e, scale, ts);
    return mivp_check_launch("token_scores_fwd");
}
extern "C" int mivp_token_scores_bwd(const float* dts, const float* W, const float* E, int32_t heads, int32_t np, int32_t e,
                                     float scale, float* dW, float* dE, mivp_stream_t stream) {
    MIVP_REQUIRE(dts && W && E && dW && dE && heads > 0 && np > 0 && e > 0);
    const size_t lds = ((size_t)(heads + np) * e + (size_t)heads * np) * sizeof(float);
    MIVP_REQUIRE(lds <= 64 * 1024);
    hipLaunchKernelGGL(k_token_scores_bwd, dim3(1), dim3(256), lds, (hipStream_t)stream, dts, W, E, heads, np, e, scale, dW, dE);
    return mivp_check_launch("token_scores_bwd");
}

// several [n][rows_i] partial arrays in one launch (the prompt-gradient partials of one block: dKp, dVp, token scores)
struct ReduceSegs {
    const float* in[4];
    float* out[4];
    long rows[4];
    int first_block[5];                                       // block range of segment i = [first_block[i], first_block[i+1])
};
__global__ __launch_bounds__(1024) void k_reduce_rows_multi(ReduceSegs sg, int nseg, long n) {
    __shared__ float part[32][33];
    int seg = 0;
    while (seg + 1 < nseg && (int)blockIdx.x >= sg.first_block[seg + 1]) ++seg;
    const float* __restrict__ in = sg.in[seg];
    const long rows = sg.rows[seg];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const long rr = (long)(blockIdx.x - sg.first_block[seg]) * 32 + col;
    float a0 = 0.f, a1 = 0.f;
    if (rr < rows) {
        long i = slice;
        // eight independent loads per round trip, added in the same order as the two-at-a-time tail loop (same bits)
        for (; i + 224 < n; i += 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[(i + 32 * u) * rows + rr];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 += v[u]; a1 += v[u + 1]; }
        }
        for (; i + 32 < n; i += 64) { a0 += in[i * rows + rr]; a1 += in[(i + 32) * rows + rr]; }
        if (i < n) a0 += in[i * rows + rr];
    }
    part[slice][col] = a0 + a1;
    __syncthreads();
    for (int h = 16; h > 0; h >>= 1) {
        if (slice < h) part[slice][col] += part[slice + h][col];
        __syncthreads();
    }
    if (slice == 0 && rr < rows) sg.out[seg][rr] = part[0][col];
}

extern "C" int mivp_reduce_rows_multi(int32_t nseg, const float* const* in, const int64_t* rows, float* const* out, int64_t n,
                                      mivp_stream_t stream) {
    MIVP_REQUIRE(nseg >= 1 && nseg <= 4 && in && rows && out && n >= 0);
    ReduceSegs sg;
    int blocks = 0;
    for (int i = 0; i < 4; ++i) {
        sg.first_block[i] = blocks;
        if (i < nseg) {
            MIVP_REQUIRE(in[i] && out[i] && rows[i] > 0);
            sg.in[i] = in[i]; sg.out[i] = out[i]; sg.rows[i] = (long)rows[i];
            blocks += (int)((rows[i] + 31) / 32);
        } else {
            sg.in[i] = nullptr; sg.out[i] = nullptr; sg.rows[i] = 0;
        }
    }
    sg.first_block[4] = blocks;
    for (int i = nseg; i < 4; ++i) sg.first_block[i + 1] = blocks;
    hipLaunchKernelGGL(k_reduce_rows_multi, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, sg, (int)nseg, (long)n);
    return mivp_check_launch("reduce_rows_multi");
}

extern "C" int mivp_reduce_rows(const float* in, int64_t n, int64_t rows, float* out, mivp_stream_t stream) {
    MIVP_REQUIRE(in && out && n >= 0 && rows > 0);
    hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)((rows + 31) / 32)), dim3(1024), 0, (hipStream_t)stream, in, (long)n,
                       (long)rows, out);
    return mivp_check_launch("reduce_rows");
}

// c[i][j] = sum_k a[i][k] * b[j][k]  (a, b: [16][32] bf16 row-major) through ONE MFMA with the lane
// maps every other kernel in this library assumes.
__global__ void k_selftest_mfma(const bf16_t* a, const bf16_t* b, float* c) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    bf16x8 fa = ld8(a + r * 32 + 8 * g);       // A[row r][k = 8g..]
    bf16x8 fb = ld8(b + r * 32 + 8 * g);       // B[k = 8g..][col r] = b[r][k]
    f32x4 acc = mfma16(fa, fb, fzero4());
    for (int j = 0; j < 4; ++j) c[(4 * g + j) * 16 + r] = acc[j];
}

extern "C" int mivp_selftest_mfma(const void* a, const void* b, float* c, mivp_stream_t stream) {
    MIVP_REQUIRE(a && b && c);
    hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, c);
    return mivp_check_launch("selftest_mfma");
}
