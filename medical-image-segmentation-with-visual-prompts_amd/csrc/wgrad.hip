// Weight gradients: dW[m][n] = alpha * sum_t A[t][m] * B[t][n]  ("TN" GEMM, K = tokens / voxels).
//
// Every trainable matrix of the path (nn.Linear of the Swin blocks and of PatchMerging, the 3x3x3
// convolutions) has a gradient of this shape: A = gradient w.r.t. the layer output, B = the layer input,
// both bf16 and token-major in HBM (one row per token), i.e. with the SUMMED index outermost.  The MFMA
// wants the summed index innermost in both operands, so tiles are staged token-major in LDS (coalesced
// 8-byte pieces) and read back with gfx950's transposing LDS load ds_read_b64_tr_b16: a 16-lane group
// reads a 4-token x 16-channel block and each lane receives 4 tokens of ONE channel -- exactly half of a
// 16x16x32 operand fragment.
//
// Work split: blockIdx.y/z pick a 64x64 block of dW, blockIdx.x a contiguous range of 128-token chunks.
// Inside a WG each of the 4 waves takes 32 tokens of the chunk (one MFMA k-step) against all 16 output
// tiles, so a wave issues 16 transposed reads per 16 MFMAs.  Per-WG results go to an fp32 workspace
// [splits][M][N]; k_gemm_tn_reduce adds the splits in a fixed order (deterministic), applies alpha,
// optionally accumulates into / permutes the destination.
//
// Operand addressing (MivpOperandDesc.mode):
//   0  rows:        (t, c) -> t*ld + c
//   1  head-split:  [T/rows][C/hd][rows][hd]   (q/k/v gradients of the attention kernels)
//   2  conv tap:    column c = tap*Cin + ci reads voxel t displaced by tap (kh-1, kw-1, kd-1), zero outside
//                   the volume (mivp_conv3d_fwd's im2col order, so dW comes out as [Cout][27][Cin])
#include "common.hpp"

namespace {

constexpr int TN_TOK = 128;      // tokens per chunk (4 waves x 32)
constexpr int TN_BLK = 64;       // output block edge

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

MIVP_DEV bf16x4 tr_read(const char* smem, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(smem + byte_off));
}

// LDS image of one operand chunk: blocks of 4 tokens x 16 channels (128 B, token-major inside the block);
// block (t4, cb) lives at slot t4*4 + (cb ^ ((t4 >> 1) & 1)): the two 16-lane groups of a 32-lane half read
// blocks two t4 apart in the same cb and the XOR puts those in opposite halves of the 256-byte bank row.
MIVP_DEV int blk_off(int t4, int cb) { return (t4 * 4 + (cb ^ ((t4 >> 1) & 1))) * 128; }

struct PieceAddr {               // per-thread constant part of one operand's column piece
    long col_off;                // element offset contributed by the column
    int dh, dw, dd;              // mode 2: tap displacement
    bool col_ok;
};

MIVP_DEV PieceAddr piece_setup(const MivpOperandDesc& o, int c, int C) {
    PieceAddr p;
    p.col_ok = c < C;
    p.dh = p.dw = p.dd = 0;
    p.col_off = c;
    if (o.mode == 1) {
        const int h = c / o.hd;
        p.col_off = (long)h * o.rows * o.hd + (c - h * o.hd);
    } else if (o.mode == 2) {
        const int tap = c / o.cin;
        p.col_off = c - tap * o.cin;
        p.dh = tap / 9 - 1;
        p.dw = (tap / 3) % 3 - 1;
        p.dd = tap % 3 - 1;
    }
    return p;
}

MIVP_DEV bf16x4 piece_load(const MivpOperandDesc& o, const PieceAddr& p, const bf16_t* __restrict__ base, long t, long T,
                           int C) {
    if (!p.col_ok || t >= T) return zero4();
    if (o.mode == 0) return ld4(base + t * o.ld + p.col_off);
    if (o.mode == 1) {
        const long w = t / o.rows;
        const int n = (int)(t - w * o.rows);
        return ld4(base + (w * (C / o.hd)) * o.rows * o.hd + (long)n * o.hd + p.col_off);
    }
    const int D = o.dims[2], W = o.dims[1], H = o.dims[0];
    long r = t;
    const int d = (int)(r % D); r /= D;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int hh = h + p.dh, ww = w + p.dw, dd = d + p.dd;
    if ((unsigned)hh >= (unsigned)H || (unsigned)ww >= (unsigned)W || (unsigned)dd >= (unsigned)D) return zero4();
    return ld4(base + (t + ((long)p.dh * W + p.dw) * D + p.dd) * o.ld + p.col_off);
}

__global__ __launch_bounds__(256) void k_gemm_tn(MivpGemmTnDesc d, const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                 float* __restrict__ part, int chunks_per_split) {
    __shared__ __attribute__((aligned(16))) char smem[2 * TN_TOK * TN_BLK * 2];   // A image, B image: 16 KB each
    char* As = smem;
    char* Bs = smem + TN_TOK * TN_BLK * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * TN_BLK, n0 = blockIdx.z * TN_BLK;
    const int cp = tid & 15, r0 = tid >> 4;                    // column piece (4 channels), first row
    const PieceAddr pa = piece_setup(d.a, m0 + 4 * cp, d.M);
    const PieceAddr pb = piece_setup(d.b, n0 + 4 * cp, d.N);

    const long nchunks = (d.T + TN_TOK - 1) / TN_TOK;
    const long c_lo = (long)blockIdx.x * chunks_per_split;
    long c_hi = c_lo + chunks_per_split;
    if (c_hi > nchunks) c_hi = nchunks;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fzero4();

    bf16x4 ra[8], rb[8];
    auto fetch = [&](long chunk) {
        const long t0 = chunk * TN_TOK + r0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ra[i] = piece_load(d.a, pa, a, t0 + 16 * i, d.T, d.M);
            rb[i] = piece_load(d.b, pb, b, t0 + 16 * i, d.T, d.N);
        }
    };
    // where this thread's pieces go: token row r0 + 16 i, channels 4cp .. 4cp+3
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = r0 + 16 * i;
            const int off = blk_off(row >> 2, cp >> 2) + (row & 3) * 32 + (cp & 3) * 8;
            *reinterpret_cast<bf16x4*>(As + off) = ra[i];
            *reinterpret_cast<bf16x4*>(Bs + off) = rb[i];
        }
    };
    // transposed fragment of output tile `ct` for this wave's 32 tokens: lane (r, g) gets tokens 8g..8g+7 of
    // channel 16ct + r.  Lane 4q+p of a 16-lane group supplies the address of block row q, channels 4p..4p+3.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto frag = [&](const char* img, int ct) -> bf16x8 {
        const int t4 = wave * 8 + 2 * g;
        const bf16x4 lo = tr_read(img, blk_off(t4, ct) + q * 32 + pp * 8);
        const bf16x4 hi = tr_read(img, blk_off(t4 + 1, ct) + q * 32 + pp * 8);
        return cat44(lo, hi);
    };

    if (c_lo < c_hi) fetch(c_lo);
    for (long c = c_lo; c < c_hi; ++c) {
        __syncthreads();                       // previous chunk's reads are done
        stage();
        __syncthreads();
        if (c + 1 < c_hi) fetch(c + 1);
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i] = frag(As, i); fb[i] = frag(Bs, i); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fa[i], fb[j], acc[i][j]);
    }

    // add the four waves in a fixed order through LDS (16 KB fp32 [64][64] image over the operand space)
    float* red = reinterpret_cast<float*>(smem);
    const int r = lane & 15;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* p = red + (16 * i + 4 * g + e) * TN_BLK + 16 * j + r;
                        *p = (w == 0 ? 0.f : *p) + acc[i][j][e];
                    }
        }
    }
    __syncthreads();
    float* dst = part + (long)blockIdx.x * d.M * d.N;
    for (int e = tid; e < TN_BLK * TN_BLK; e += 256) {
        const int m = m0 + (e >> 6), n = n0 + (e & 63);
        if (m < d.M && n < d.N) dst[(long)m * d.N + n] = red[e];
    }
}

// out[m][perm(n)] = (accumulate ? out : 0) + alpha * sum_s part[s][m][n];  perm_cin > 0 turns the im2col column
// order tap*Cin + ci into nn.Conv3d's ci*27 + tap.
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const float* __restrict__ part, int splits, int M, int N, float alpha,
                                                        int accumulate, int perm_cin, float* __restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)M * N) return;
    float s = 0.f;
    for (int i = 0; i < splits; ++i) s += part[(long)i * M * N + e];
    s *= alpha;
    long o = e;
    if (perm_cin > 0) {
        const int m = (int)(e / N), n = (int)(e - (long)m * N);
        const int tap = n / perm_cin, ci = n - tap * perm_cin;
        o = (long)m * N + (long)ci * 27 + tap;
    }
    out[o] = accumulate ? out[o] + s : s;
}

int tn_splits(const MivpGemmTnDesc* d, int* chunks_per_split) {
    const long nchunks = (d->T + TN_TOK - 1) / TN_TOK;
    const long blocks = (long)((d->M + TN_BLK - 1) / TN_BLK) * ((d->N + TN_BLK - 1) / TN_BLK);
    long want = (1024 + blocks - 1) / blocks;                 // ~4 WGs per CU over the whole launch
    if (want > nchunks) want = nchunks;
    if (want < 1) want = 1;
    const long cps = (nchunks + want - 1) / want;
    *chunks_per_split = (int)(cps < 1 ? 1 : cps);
    return (int)((nchunks + *chunks_per_split - 1) / *chunks_per_split);
}

int check_operand(const MivpOperandDesc& o, int C) {
    if (o.mode == 0) return o.ld >= C && o.ld % 4 == 0;
    if (o.mode == 1) return o.hd > 0 && o.hd % 4 == 0 && C % o.hd == 0 && o.rows > 0;
    if (o.mode == 2) return o.cin > 0 && o.cin % 4 == 0 && C == 27 * o.cin && o.ld >= o.cin && o.ld % 4 == 0 &&
                            o.dims[0] > 0 && o.dims[1] > 0 && o.dims[2] > 0;
    return 0;
}

}  // namespace

extern "C" size_t mivp_gemm_tn_ws(const MivpGemmTnDesc* d) {
    int cps;
    const int splits = tn_splits(d, &cps);
    return (size_t)splits * d->M * d->N * sizeof(float);
}

extern "C" int mivp_gemm_tn(const MivpGemmTnDesc* d, const void* a, const void* b, void* workspace, size_t ws_bytes,
                            float* out, mivp_stream_t stream) {
    MIVP_REQUIRE(d && a && b && workspace && out);
    MIVP_REQUIRE(d->T > 0 && d->M > 0 && d->N > 0);
    MIVP_REQUIRE(check_operand(d->a, d->M) && check_operand(d->b, d->N));
    MIVP_REQUIRE(d->perm_cin == 0 || (d->b.mode == 2 && d->perm_cin == d->b.cin));
    int cps;
    const int splits = tn_splits(d, &cps);
    MIVP_REQUIRE(ws_bytes >= (size_t)splits * d->M * d->N * sizeof(float));
    const dim3 grid(splits, (d->M + TN_BLK - 1) / TN_BLK, (d->N + TN_BLK - 1) / TN_BLK);
    hipLaunchKernelGGL(k_gemm_tn, grid, dim3(256), 0, (hipStream_t)stream, *d, (const bf16_t*)a, (const bf16_t*)b,
                       (float*)workspace, cps);
    const long total = (long)d->M * d->N;
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, splits, d->M, d->N, d->alpha, d->accumulate, d->perm_cin, out);
    return mivp_check_launch("mivp_gemm_tn");
}
