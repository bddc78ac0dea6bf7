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

// Where token t sits in each operand's token space, kept as 32-bit counters that advance with the token index: decoding
// t with 64-bit divisions for every 8-byte piece (five per conv-tap piece) was ~10x the MFMA time of a chunk.
struct TokPos {
    long t;                      // token index
    int h, w, d;                 // mode 2: voxel coordinates (the batch index is not needed: bounds are per volume)
    long win; int n;             // mode 1: window-head row block, row inside it
};

MIVP_DEV void tokpos_init(TokPos& s, long t, const MivpOperandDesc& oa, const MivpOperandDesc& ob) {
    s.t = t;
    s.h = s.w = s.d = 0; s.win = 0; s.n = 0;
    const MivpOperandDesc& o2 = oa.mode == 2 ? oa : ob;
    if (o2.mode == 2) {
        long r = t;
        s.d = (int)(r % o2.dims[2]); r /= o2.dims[2];
        s.w = (int)(r % o2.dims[1]); r /= o2.dims[1];
        s.h = (int)(r % o2.dims[0]);
    }
    const MivpOperandDesc& o1 = oa.mode == 1 ? oa : ob;
    if (o1.mode == 1) { s.win = t / o1.rows; s.n = (int)(t - s.win * o1.rows); }
}

MIVP_DEV void tokpos_advance(TokPos& s, int step, const MivpOperandDesc& oa, const MivpOperandDesc& ob) {
    s.t += step;
    const MivpOperandDesc& o2 = oa.mode == 2 ? oa : ob;
    if (o2.mode == 2) {
        s.d += step;
        while (s.d >= o2.dims[2]) { s.d -= o2.dims[2]; if (++s.w >= o2.dims[1]) { s.w = 0; if (++s.h >= o2.dims[0]) s.h = 0; } }
    }
    const MivpOperandDesc& o1 = oa.mode == 1 ? oa : ob;
    if (o1.mode == 1) {
        s.n += step;
        while (s.n >= o1.rows) { s.n -= o1.rows; ++s.win; }
    }
}

// a column piece is 4 channels (8-byte load) or, when every column offset and row stride involved is a multiple of 8
// (P8, decided on the host), 8 channels through one 16-byte load: half the loads, address updates and LDS stores
template <bool P8> struct Piece { typedef bf16x4 type; };
template <> struct Piece<true> { typedef bf16x8 type; };
template <bool P8> MIVP_DEV typename Piece<P8>::type piece_zero();
template <> MIVP_DEV bf16x4 piece_zero<false>() { return zero4(); }
template <> MIVP_DEV bf16x8 piece_zero<true>() { return zero8(); }
template <bool P8> MIVP_DEV typename Piece<P8>::type piece_ld(const bf16_t* p);
template <> MIVP_DEV bf16x4 piece_ld<false>(const bf16_t* p) { return ld4(p); }
template <> MIVP_DEV bf16x8 piece_ld<true>(const bf16_t* p) { return ld8(p); }

template <bool P8>
MIVP_DEV typename Piece<P8>::type piece_load(const MivpOperandDesc& o, const PieceAddr& p, const bf16_t* __restrict__ base,
                                             const TokPos& s, long T, int C) {
    const long t = s.t;
    if (!p.col_ok || t >= T) return piece_zero<P8>();
    if (o.mode == 0) return piece_ld<P8>(base + t * o.ld + p.col_off);
    if (o.mode == 1) return piece_ld<P8>(base + (s.win * (C / o.hd)) * o.rows * o.hd + (long)s.n * o.hd + p.col_off);
    const int D = o.dims[2], W = o.dims[1], H = o.dims[0];
    const int hh = s.h + p.dh, ww = s.w + p.dw, dd = s.d + p.dd;
    if ((unsigned)hh >= (unsigned)H || (unsigned)ww >= (unsigned)W || (unsigned)dd >= (unsigned)D) return piece_zero<P8>();
    return piece_ld<P8>(base + (t + ((long)p.dh * W + p.dw) * D + p.dd) * o.ld + p.col_off);
}


template <bool P8>
__global__ __launch_bounds__(256) void k_gemm_tn(MivpGemmTnDesc d, const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                 float* __restrict__ part, int chunks_per_split) {
    constexpr int PW = P8 ? 8 : 4;                            // channels per piece
    constexpr int PPR = TN_BLK / PW;                          // pieces per 64-channel row: 16 or 8
    constexpr int RPP = 256 / PPR;                            // token rows covered per pass: 16 or 32
    constexpr int NP = TN_TOK / RPP;                          // passes per 128-token chunk: 8 or 4
    __shared__ __attribute__((aligned(16))) char smem[2 * TN_TOK * TN_BLK * 2];   // A image, B image: 16 KB each
    char* As = smem;
    char* Bs = smem + TN_TOK * TN_BLK * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid = (N blocks, M blocks, token splits): the blocks that share a token range are neighbours in dispatch order and
    // stream it in step, so the operands come out of L2 (with the split index fastest every 64-column block of a conv
    // weight gradient re-read dy and its tap-displaced x from HBM: 4.0 ms for the 144->48 conv at 48^3, 41 TFLOP/s)
    const int m0 = blockIdx.y * TN_BLK, n0 = blockIdx.x * TN_BLK;
    const int cp = tid % PPR, r0 = tid / PPR;                  // column piece (PW channels), first row
    const PieceAddr pa = piece_setup(d.a, m0 + PW * cp, d.M);
    const PieceAddr pb = piece_setup(d.b, n0 + PW * cp, d.N);

    const long nchunks = (d.T + TN_TOK - 1) / TN_TOK;
    const long c_lo = (long)blockIdx.z * chunks_per_split;
    long c_hi = c_lo + chunks_per_split;
    if (c_hi > nchunks) c_hi = nchunks;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fzero4();

    typedef typename Piece<P8>::type piece_t;
    piece_t ra[NP], rb[NP];
    TokPos pos;                                               // token c_lo * 128 + r0, then +RPP per piece, chunk after chunk
    tokpos_init(pos, c_lo * TN_TOK + r0, d.a, d.b);
    auto fetch = [&](long) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            ra[i] = piece_load<P8>(d.a, pa, a, pos, d.T, d.M);
            rb[i] = piece_load<P8>(d.b, pb, b, pos, d.T, d.N);
            tokpos_advance(pos, RPP, d.a, d.b);
        }
    };
    // where this thread's pieces go: token row r0 + RPP i, channels PW cp .. PW cp + PW - 1 (16-channel block PW cp / 16)
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = r0 + RPP * i;
            const int off = blk_off(row >> 2, (PW * cp) >> 4) + (row & 3) * 32 + ((PW * cp) & 15) * 2;
            *reinterpret_cast<piece_t*>(As + off) = ra[i];
            *reinterpret_cast<piece_t*>(Bs + off) = rb[i];
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
    float* dst = part + (long)blockIdx.z * d.M * d.N;
    for (int e = tid; e < TN_BLK * TN_BLK; e += 256) {
        const int m = m0 + (e >> 6), n = n0 + (e & 63);
        if (m < d.M && n < d.N) dst[(long)m * d.N + n] = red[e];
    }
}

// out[m][perm(n)] = (accumulate ? out : 0) + alpha * sum_s part[s][m][n];  perm_cin > 0 turns the im2col column
// order tap*Cin + ci into nn.Conv3d's ci*27 + tap.  Block = 32 elements x 32 split-slices (the shape of
// k_reduce_rows): a slice adds splits slice, slice+32, ... in order, the 32 slice sums meet in a fixed LDS tree.
__global__ __launch_bounds__(1024) void k_gemm_tn_reduce(const float* __restrict__ part, int splits, int M, int N, float alpha,
                                                         int accumulate, int perm_cin, float* __restrict__ out) {
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5, S = blockDim.x >> 5;   // S slices: a power of two <= 32
    const long total = (long)M * N;
    const long e = (long)blockIdx.x * 32 + col;
    float s = 0.f;
    if (e < total)
        for (int i = slice; i < splits; i += S) s += part[(long)i * total + e];
    red[slice][col] = s;
    __syncthreads();
    for (int h = S >> 1; h > 0; h >>= 1) {
        if (slice < h) red[slice][col] += red[slice + h][col];
        __syncthreads();
    }
    if (slice != 0 || e >= total) return;
    s = red[0][col] * alpha;
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
    long want = (512 + blocks - 1) / blocks;                  // ~2 WGs per CU over the whole launch
    if (want > nchunks) want = nchunks;
    if (want < 1) want = 1;
    const long cps = (nchunks + want - 1) / want;
    *chunks_per_split = (int)(cps < 1 ? 1 : cps);
    return (int)((nchunks + *chunks_per_split - 1) / *chunks_per_split);
}

// every column offset and row stride of the operand is a multiple of 8 elements: 16-byte pieces are legal
int operand_p8(const MivpOperandDesc& o, int C, const void* base) {
    if (C % 8 || ((uintptr_t)base & 15)) return 0;
    if (o.mode == 0) return o.ld % 8 == 0;
    if (o.mode == 1) return o.hd % 8 == 0;
    return o.cin % 8 == 0 && o.ld % 8 == 0;
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
    const dim3 grid((d->N + TN_BLK - 1) / TN_BLK, (d->M + TN_BLK - 1) / TN_BLK, splits);
    if (operand_p8(d->a, d->M, a) && operand_p8(d->b, d->N, b))
        hipLaunchKernelGGL(k_gemm_tn<true>, grid, dim3(256), 0, (hipStream_t)stream, *d, (const bf16_t*)a, (const bf16_t*)b,
                           (float*)workspace, cps);
    else
        hipLaunchKernelGGL(k_gemm_tn<false>, grid, dim3(256), 0, (hipStream_t)stream, *d, (const bf16_t*)a, (const bf16_t*)b,
                           (float*)workspace, cps);
    const long total = (long)d->M * d->N;
    int slices = 1;
    while (slices < splits && slices < 32) slices *= 2;
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3((unsigned)((total + 31) / 32)), dim3(32 * slices), 0, (hipStream_t)stream,
                       (const float*)workspace, splits, d->M, d->N, d->alpha, d->accumulate, d->perm_cin, out);
    return mivp_check_launch("mivp_gemm_tn");
}

// ---------------------------------------------------------------------------------------------
// LayerNorm parameter gradients + the normalised rows the Linear weight gradients need.
//   rows x[t] (direct, or gathered through the block's tok_src table like mivp_swin_qkv_fwd does),
//   dn[t] = gradient w.r.t. the LayerNorm OUTPUT (before gamma is applied backwards), emitted by the
//   fused backward kernels.
//   pass 1  k_ln_rowstats : (mean, rstd) per row; rstd = -1 marks padding slots (tok_src == -2)
//   pass 2  k_ln_wgrad    : n[t] = xhat*gamma + beta (bf16, the B operand of mivp_gemm_tn; zero on padding
//                           slots), per-block partials of dgamma = sum dn*xhat and dbeta = sum dn.  Every
//                           thread keeps one 8-channel group, blocks reduce in a fixed order.
// ---------------------------------------------------------------------------------------------
namespace {

struct LnRow { const bf16_t* p; int cls; };     // cls: 0 data row, 1 zero row (window zero-pad), 2 padding slot

MIVP_DEV LnRow ln_row(const bf16_t* __restrict__ x, const int* __restrict__ tok_src, long t, int C, int Nqp, int P, long vol) {
    LnRow r;
    r.cls = 0;
    r.p = x + t * C;
    if (tok_src) {
        const unsigned bp = (unsigned)t / (unsigned)Nqp;          // callers check T < 2^31
        const int slot = (int)((unsigned)t - bp * (unsigned)Nqp);
        const long b = bp / (unsigned)P;
        const int pw = (int)(bp - (unsigned)b * (unsigned)P);
        const int src = tok_src[pw * Nqp + slot];
        r.cls = src >= 0 ? 0 : (src == -1 ? 1 : 2);
        r.p = x + (b * vol + (src >= 0 ? src : 0)) * C;
    }
    return r;
}

__global__ __launch_bounds__(256) void k_ln_rowstats(const bf16_t* __restrict__ x, const int* __restrict__ tok_src, long T, int C,
                                                     int Nqp, int P, long vol, float eps, float* __restrict__ stats) {
    const int l16 = threadIdx.x & 15;
    const long t = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = t < T;
    LnRow row = ln_row(x, tok_src, ok ? t : 0, C, Nqp, P, vol);
    float s = 0.f;
    if (ok && row.cls == 0)
        for (int c4 = l16; c4 < C / 4; c4 += 16) { const bf16x4 v = ld4(row.p + 4 * c4); s += (float)v[0] + (float)v[1] + (float)v[2] + (float)v[3]; }
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
    if (ok && row.cls == 0)
        for (int c4 = l16; c4 < C / 4; c4 += 16) {
            const bf16x4 v = ld4(row.p + 4 * c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dlt = (float)v[j] - mean; q += dlt * dlt; }
        }
    for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (ok && l16 == 0) {
        stats[2 * t] = mean;
        stats[2 * t + 1] = row.cls == 2 ? -1.f : rsqrtf(q / (float)C + eps);
    }
}

__global__ __launch_bounds__(256) void k_ln_wgrad(const bf16_t* __restrict__ x, const int* __restrict__ tok_src,
                                                  const bf16_t* __restrict__ dn, const float* __restrict__ stats, long T, int C,
                                                  int Nqp, int P, long vol, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, bf16_t* __restrict__ n_out,
                                                  float* __restrict__ part) {
    __shared__ float lsum[256 * 16];
    const int G = C / 8;
    const long items = T * G;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const int cg = (int)(gtid % G);
    float gm[8], bt[8], s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { gm[i] = gamma[cg * 8 + i]; bt[i] = beta[cg * 8 + i]; s1[i] = 0.f; s2[i] = 0.f; }
    for (long it = gtid; it < items; it += stride) {
        const long t = (unsigned)it / (unsigned)G;                 // host checks items < 2^31
        const float mean = stats[2 * t], rstd = stats[2 * t + 1];
        bf16x8 nv = zero8();
        if (rstd >= 0.f) {
            const LnRow row = ln_row(x, tok_src, t, C, Nqp, P, vol);
            const bf16x8 xv = row.cls == 0 ? ld8(row.p + cg * 8) : zero8();
            const bf16x8 dv = ld8(dn + t * C + cg * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = ((float)xv[i] - mean) * rstd;
                const float g = (float)dv[i];
                nv[i] = (bf16_t)(xh * gm[i] + bt[i]);
                s1[i] += g;               // dbeta
                s2[i] += g * xh;          // dgamma
            }
        }
        st8(n_out + t * C + cg * 8, nv);
    }
    block_reduce_groups(lsum, s1, s2, G, C, (long)blockIdx.x * 256, part + (long)blockIdx.x * 2 * C);
}

// Relative-position-bias table gradients from the window-summed gradient of the K' augmentation columns
// (mivp_relbias_aug's layout).  dka [heads][Nkp][32] f32.  One wave per (head, table entry): its lanes stride the
// keys that read the entry, then a fixed butterfly adds the 64 lane sums (deterministic).
__global__ __launch_bounds__(64) void k_relbias_grad(MivpSwinDesc d, const float* __restrict__ dka, float* __restrict__ d_th,
                                                     float* __restrict__ d_tw, float* __restrict__ d_td) {
    const int w0 = d.win[0], w1 = d.win[1], w2 = d.win[2];
    const int n0 = 2 * w0 - 1, n1 = 2 * w1 - 1, n2 = 2 * w2 - 1, nall = n0 + n1 + n2;
    const int head = blockIdx.x / nall, e = blockIdx.x - head * nall;
    const float* g = dka + (long)head * d.Nkp * 32;
    float acc = 0.f;
    for (int m = threadIdx.x; m < d.Nq; m += 64) {
        const int k2 = m % w2, k1 = (m / w2) % w1, k0 = m / (w2 * w1);
        const float* gm = g + m * 32;
        if (e < n0) {
            const int a = k0 + w0 - 1 - e;                           // th[k0 - a + w0 - 1] == th[e]
            if (a >= 0 && a < w0) acc += gm[a];
        } else if (e < n0 + n1) {
            const int a = k1 + w1 - 1 - (e - n0);
            if (a >= 0 && a < w1) acc += gm[w0 + a];
        } else {
            const int ee = e - n0 - n1;
            const int a = k2 + w2 - 1 - ee;                          // query i2 = a < w2 - 1 has its own column
            if (a >= 0 && a < w2 - 1) acc += gm[w0 + w1 + a];
            if (k2 == ee) {                                          // the folded i2 = w2-1 term: + every i0 column, - every i2 column
                for (int c = 0; c < w0; ++c) acc += gm[c];
                for (int c = 0; c < w2 - 1; ++c) acc -= gm[w0 + w1 + c];
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) {
        if (e < n0) d_th[head * n0 + e] = acc;
        else if (e < n0 + n1) d_tw[head * n1 + (e - n0)] = acc;
        else d_td[head * n2 + (e - n0 - n1)] = acc;
    }
}

}  // namespace

extern "C" int mivp_ln_wgrad(const void* x, const int32_t* tok_src, const void* dn, int64_t T, int32_t C, int32_t Nqp,
                             int32_t P, int64_t vol, float eps, const float* gamma, const float* beta, float* stats,
                             void* n_out, int32_t nblk, float* part, mivp_stream_t stream) {
    MIVP_REQUIRE(x && dn && gamma && beta && stats && n_out && part && T > 0 && C > 0 && C % 8 == 0);
    MIVP_REQUIRE(tok_src == nullptr || (Nqp > 0 && P > 0 && vol > 0 && T % Nqp == 0));
    MIVP_REQUIRE(nblk > 0 && ((long)nblk * 256) % (C / 8) == 0);
    MIVP_REQUIRE(T * (C / 8) < (1L << 31));                       // 32-bit decode in the kernels
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ln_rowstats, dim3((unsigned)((T + 15) / 16)), dim3(256), 0, st, (const bf16_t*)x, tok_src, (long)T,
                       (int)C, (int)Nqp, (int)P, (long)vol, eps, stats);
    hipLaunchKernelGGL(k_ln_wgrad, dim3(nblk), dim3(256), 0, st, (const bf16_t*)x, tok_src, (const bf16_t*)dn,
                       (const float*)stats, (long)T, (int)C, (int)Nqp, (int)P, (long)vol, gamma, beta, (bf16_t*)n_out, part);
    return mivp_check_launch("mivp_ln_wgrad");
}

extern "C" int mivp_relbias_grad(const MivpSwinDesc* d, const float* dka, float* d_th, float* d_tw, float* d_td,
                                 mivp_stream_t stream) {
    MIVP_REQUIRE(d && dka && d_th && d_tw && d_td && d->augp <= 32);
    const int nall = 2 * (d->win[0] + d->win[1] + d->win[2]) - 3;
    hipLaunchKernelGGL(k_relbias_grad, dim3(d->heads * nall), dim3(64), 0, (hipStream_t)stream, *d, dka, d_th, d_tw, d_td);
    return mivp_check_launch("mivp_relbias_grad");
}
