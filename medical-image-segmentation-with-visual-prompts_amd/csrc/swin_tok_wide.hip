// Token GEMM kernels of the WIDE stages (C = 96, 192, 384: encoder stages 1-3 and the decoder blocks of the same widths).
// Same operators as k_swin_qkv_fwd / k_swin_proj_mlp_fwd (swin_fwd.hip) and k_swin_proj_mlp_bwd / k_swin_qkv_bwd
// (swin_bwd.hip) -- reference swin_transformer/swin_block.py:145-255 and its autograd -- in another work split.
//
// Why (r02 counters, DESIGN.md "token kernels"): the one-wave-per-16-tokens form is bound by the texture-address unit --
// TA_BUSY is ~80 % of the kernel time -- because every global access is made in the MFMA operand lane map (lane (r, g):
// token row r, 16-byte chunk g), where ADJACENT LANES TOUCH DIFFERENT ROWS: a wave load costs one TA cycle per lane
// instead of one per 64 bytes.  At the deep stages it also leaves a few hundred single-wave-per-SIMD workgroups, each
// walking every column tile behind ~30 dependent load round trips.  Here:
//   * global memory is only touched in ROW-CONTIGUOUS patterns: TPR adjacent lanes per token row read / write 16-byte
//     pieces of it; the head-major q / k / v tensors are touched as the contiguous [32 tokens][head_dim] chunk each
//     (tensor, head) has per 32-token granule (windows hold a multiple of 32 slots);
//   * the MFMA lane map only ever meets LDS: row images [rows][K] (+16 B per row: conflict-free ds_read_b128 fragments);
//   * a workgroup owns 32 tokens per token group and its four waves split the output COLUMNS (NCG column groups x NTG
//     token groups; CT % 4 == 0: 4 x 1, otherwise 2 x 2), so a weight fragment (fragment image: one coalesced 1 KB load)
//     feeds two MFMAs and per-token work (LayerNorm statistics, gathers) is done once per workgroup;
//   * row reductions across column groups meet in LDS in column-group order (bit-reproducible);
//   * every global load is unconditional (common.hpp "Branch-free loads").
// The launchers at the bottom are called by the C entries of swin_fwd.hip / swin_bwd.hip when mivp_tok_wide_supported().
#include "common.hpp"
#include <cstdlib>

namespace {

template <int CT>
struct WideGeom {
    static constexpr int C = 16 * CT;
    static constexpr int KP = (C + 31) / 32 * 32;                // row images pad K to whole k-steps (C = 48: zero columns 48..63)
    static constexpr int KS = KP / 32;
    static constexpr int NCG = (CT % 4 == 0) ? 4 : (CT % 2 == 0 ? 2 : 1);     // column groups
    static constexpr int NTG = 4 / NCG;                          // token groups (32-token granules per workgroup)
    static constexpr int NCT = CT / NCG;                         // column tiles of a [.][C] output per wave
    static constexpr int ROWS = 32 * NTG;                        // token rows per workgroup
    static constexpr int TPR = 256 / ROWS;                       // adjacent lanes per token row in the row phases
    static constexpr int XPT = C / 8 / TPR;                      // 16-byte pieces of a [C] row per lane
    static_assert(CT % NCG == 0 && (C / 8) % TPR == 0, "even splits");
};

// LDS row image: [rows][K] bf16, +16 B per row (the 16 rows of a fragment read then cover all banks)
template <int K>
struct RowImg {
    static constexpr int ROWB = 2 * K + 16;
    static MIVP_DEV bf16x8 frag(const char* img, int row, int k0) { return *reinterpret_cast<const bf16x8*>(img + row * ROWB + 2 * k0); }
    static MIVP_DEV bf16x4 get4(const char* img, int row, int k0) { return *reinterpret_cast<const bf16x4*>(img + row * ROWB + 2 * k0); }
    static MIVP_DEV void put4(char* img, int row, int k0, bf16x4 v) { *reinterpret_cast<bf16x4*>(img + row * ROWB + 2 * k0) = v; }
    static MIVP_DEV void put8(char* img, int row, int k0, bf16x8 v) { *reinterpret_cast<bf16x8*>(img + row * ROWB + 2 * k0) = v; }
};

struct RowTok { long tt, bp, b; int slot, pw; bool live; };
MIVP_DEV RowTok row_token(const MivpSwinDesc& d, long t) {      // a dead row decodes as token 0 (valid addresses)
    RowTok ti;
    const long T = (long)d.B * d.P * d.Nqp;
    ti.live = t < T;
    const unsigned tu = ti.live ? (unsigned)t : 0u;
    const unsigned bpu = tu / (unsigned)d.Nqp;
    ti.tt = tu;
    ti.bp = bpu;
    ti.slot = (int)(tu - bpu * (unsigned)d.Nqp);
    ti.pw = (int)(bpu % (unsigned)d.P);
    ti.b = bpu / (unsigned)d.P;
    return ti;
}
template <int TPR>
MIVP_DEV float row_sum(float v) {                                // sum over the TPR adjacent lanes of a token row
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
MIVP_DEV void to_f32(bf16x8 v, float (&out)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = (float)v[e];
}

// Piece-major walk over the workgroup's ROWS x C/8 sixteen-byte row pieces (the narrow stages: with two or four lanes per row
// a wave instruction is 32- or 64-byte runs 2C bytes apart and the texture-address unit stays ~60 % busy): piece
// P = tid + 256 i, row P / (C/8) -- adjacent lanes take adjacent pieces, so the token-major tensors (o, t1, dt1, dO) move as
// whole 1 KB wave accesses and gathered / scattered voxel rows as runs of at least one row.
template <int CT>
struct FlatRows {
    using G = WideGeom<CT>;
    static constexpr int PPR = G::C / 8, N = G::XPT;
    static_assert(G::ROWS * PPR == 256 * N, "pieces split evenly over the threads");
    int prow[N], col[N];
    RowTok tk[N];
    MIVP_DEV FlatRows(const MivpSwinDesc& d, long row0, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int P = tid + 256 * i;
            prow[i] = P / PPR;
            col[i] = 8 * (P - prow[i] * PPR);
            tk[i] = row_token(d, row0 + prow[i]);
        }
    }
};
template <int CT> constexpr bool flat_rows() { return CT == 3; }

// Head-major q | k | v pieces of this workgroup's granules.  Per 32-token granule and (tensor, head) the tensor
// [B*P*heads][Nqp][hd] holds one contiguous chunk of 32 * hd elements (Nqp % 32 == 0: a granule never leaves its window).
// Piece P (8 bytes) of a granule: chunk = P / (8 hd), q = P % (8 hd) the piece within the chunk (token q / hd4, columns
// 4 (q % hd4)): consecutive P are consecutive addresses.
// visit(i, element offset within the tensor, tensor 0..2, image row, image column, live)
template <int C, int NTG>
struct HeadMajor {
    long gbase[NTG];                                              // element offset of (granule, head 0) in a q / k / v tensor
    bool glive[NTG];
    int hd, hd4, heads;
    long head_stride;
    FastDiv by_chunk, by_hd4, by_heads;
    static constexpr int PER_GRAN = 24 * C;                       // 3 tensors x 32 tokens x C / 4 pieces
    static constexpr int PPT = PER_GRAN * NTG / 256;              // pieces per thread
    MIVP_DEV HeadMajor(const MivpSwinDesc& d, long row0)
        : hd(C / d.heads), hd4(C / d.heads / 4), heads(d.heads), head_stride((long)d.Nqp * (C / d.heads)),
          by_chunk(8 * (C / d.heads)), by_hd4(C / d.heads / 4), by_heads(d.heads) {
        const long T = (long)d.B * d.P * d.Nqp;
#pragma unroll
        for (int gi = 0; gi < NTG; ++gi) {                        // the only full divisions: once per granule, not per piece
            const long t0 = row0 + 32 * gi;
            glive[gi] = t0 < T;
            const unsigned tu = glive[gi] ? (unsigned)t0 : 0u;
            const unsigned bp = tu / (unsigned)d.Nqp, slot0 = tu - bp * (unsigned)d.Nqp;
            gbase[gi] = ((long)bp * heads * d.Nqp + slot0) * hd;
        }
    }
    // visit(i, element offset within the tensor, tensor 0..2, image row, image column, live) for this thread's pieces
    template <class F>
    MIVP_DEV void each(int tid, F&& visit) const {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int P = tid + 256 * i;
            const int gran = P / PER_GRAN, pg = P - gran * PER_GRAN;
            const int chunk = by_chunk.div(pg), q = pg - chunk * 8 * hd;
            const int tensor = by_heads.div(chunk), head = chunk - tensor * heads;
            const int tok = by_hd4.div(q), piece = q - tok * hd4;
            long base = gbase[0];
            bool live = glive[0];
#pragma unroll
            for (int gi = 1; gi < NTG; ++gi) { base = sel(gran == gi, gbase[gi], base); live = gran == gi ? glive[gi] : live; }
            visit(i, base + head * head_stride + 4 * q, tensor, 32 * gran + tok, tensor * C + head * hd + 4 * piece, live);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// gather + LayerNorm + QKV                                             [k_swin_qkv_fwd, swin_fwd.hip]
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256, CT <= 3 ? 4 : (CT <= 12 ? 3 : 2)) void k_qkv_fwd_wide(MivpSwinDesc d, const bf16_t* __restrict__ x, const int* __restrict__ tok_src,
                                                         const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                         const bf16_t* __restrict__ wqkv, bf16_t* __restrict__ q,
                                                         bf16_t* __restrict__ k, bf16_t* __restrict__ v) {
    using G = WideGeom<CT>;
    constexpr int C = G::C, KS = G::KS, ROWS = G::ROWS, TPR = G::TPR, NCG = G::NCG, XPT = G::XPT, NT3 = 3 * G::NCT;
    using XI = RowImg<G::KP>;
    using OI = RowImg<3 * C>;
    // one column group (C = 48): a wave's B fragments live in registers, so the output image may overlay the input image
    // (four workgroups per CU instead of two); the barrier below the fragment reads makes that safe
    constexpr bool OVERLAY = NCG == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ximg = smem;                                           // [ROWS][C]  LayerNorm output (the GEMM's B operand)
    char* Oimg = OVERLAY ? smem : Ximg + ROWS * XI::ROWB;        // [ROWS][3C] q * scale | k * log2 e | v
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * ROWS;
    {   // ---- rows: gather, LayerNorm ----
        const int row = tid / TPR, sub = tid % TPR;
        const RowTok ti = row_token(d, row0 + row);
        const int src = sel(ti.live, tok_src[ti.pw * d.Nqp + ti.slot], -2);
        const long xoff = (ti.b * d.vol_in + max(src, 0)) * (long)C;
        bf16x8 xr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) xr[i] = ld8(x + xoff + 8 * (sub + TPR * i));
        f32x4 w4[XPT][2], b4[XPT][2];
#pragma unroll
        for (int i = 0; i < XPT; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                w4[i][h] = *reinterpret_cast<const f32x4*>(ln_w + 8 * (sub + TPR * i) + 4 * h);
                b4[i][h] = *reinterpret_cast<const f32x4*>(ln_b + 8 * (sub + TPR * i) + 4 * h);
            }
        float xs[XPT][8], sum = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            to_f32(keep_if(xr[i], src >= 0), xs[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += xs[i][e];
        }
        const float mean = row_sum<TPR>(sum) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dvv = xs[i][e] - mean; var += dvv * dvv; }
        const float rstd = rsqrtf(row_sum<TPR>(var) / (float)C + d.ln_eps);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            bf16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (bf16_t)((xs[i][e] - mean) * rstd * w4[i][e >> 2][e & 3] + b4[i][e >> 2][e & 3]);
            XI::put8(Ximg, row, 8 * (sub + TPR * i), keep_if(y, src >= -1));      // -1: a zero-pad token still goes through LN (-> beta)
        }
        if (G::KP != C && sub == 0)
            for (int c = C; c < G::KP; c += 8) XI::put8(Ximg, row, c, zero8());          // the k-step padding of the B operand
    }
    __syncthreads();
    {   // ---- GEMM: this wave's column tiles of the 3C outputs, two token tiles ----
        const int cg = wave % NCG, tg = wave / NCG;
        const int R0 = tg * 32 + r, R1 = R0 + 16;
        bf16x8 b0[KS], b1[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { b0[s] = XI::frag(Ximg, R0, 32 * s + 8 * g); b1[s] = XI::frag(Ximg, R1, 32 * s + 8 * g); }
        if (OVERLAY) __syncthreads();
#pragma unroll
        for (int j = 0; j < NT3; ++j) {
            const int nt = cg + NCG * j;
            f32x4 a0 = fzero4(), a1 = fzero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 a = wfrag(wqkv, KS, nt, s, lane);
                a0 = mfma16(a, b0[s], a0);
                a1 = mfma16(a, b1[s], a1);
            }
            const int n0 = 16 * nt + 4 * g;
            const float sc = sel(n0 < C, d.q_scale, sel(n0 < 2 * C, MIVP_LOG2E, 1.0f));      // K carries log2(e): common.hpp
            OI::put4(Oimg, R0, n0, pack4(a0 * sc));
            OI::put4(Oimg, R1, n0, pack4(a1 * sc));
        }
    }
    __syncthreads();
    // ---- q | k | v: contiguous head-major chunks ----
    const long to_k = k - q, to_v = v - q;
    const HeadMajor<C, G::NTG> hm(d, row0);
    hm.each(tid, [&](int, long off, int tensor, int irow, int icol, bool live) {
        if (live) st4(q + sel(tensor == 0, 0L, sel(tensor == 1, to_k, to_v)) + off, OI::get4(Oimg, irow, icol));
    });
}

// ---------------------------------------------------------------------------------------------
// gather + LayerNorm + QKV, WEIGHT-STATIONARY form for C = 96 / 192 (round 3)
// k_qkv_fwd_wide streams the whole 6 C^2-byte fragment image through every workgroup of 32 (C = 192) or 64 tokens: at
// stage 2 that is 704 x 221 KB = 155 MB of L2 reads per launch for 31 MB of tensors, and the launch ran at the L2's ~8 TB/s
// (19.4 us).  Its [ROWS][3C] output image (the head-major q | k | v chunks are written from it) is also what keeps it at
// 32 rows.  Here a workgroup owns 64 tokens, a wave a set of column-tile PAIRS (32 output channels = one head at
// head_dim 32): the pair's 2 KS weight fragments sit in registers (the next pair's travel meanwhile) and the four token
// tiles pass under them, B fragments from the LDS row image (one read per two MFMAs).  The accumulators leave from the
// MFMA lane map -- lane (r, g): token r, channels 4 g .. 4 g + 3 -- as 8-byte pieces: 16 tokens x 32 contiguous bytes per
// store, the two tiles of a pair completing 64-byte head rows.  No output image, no second index decode, half (C = 192)
// the weight traffic, LDS = the 64 x C input image only.
// ---------------------------------------------------------------------------------------------
template <int KS>
MIVP_DEV void ws_load_pair(const bf16_t* __restrict__ wqkv, int pair, int lane, bf16x8 (&a)[2][KS]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) a[t][s] = wfrag(wqkv, KS, 2 * pair + t, s, lane);
}
// one column-tile pair of k_qkv_fwd_ws: the four token tiles under the pair's fragments, stores from the MFMA lane map
template <int KS>
struct WsOut {
    const char* Ximg;
    bf16_t* q;
    long to_k, to_v;                                              // k - q, v - q (one global pointer: a select between three decays to flat)
    long head_stride, row0, T;
    float q_scale;
    int hd, r, g, heads, Nqp;
    unsigned bp0, slot0;                                          // window and slot of the workgroup's first row (Nqp >= 64: one wrap at most)
    template <int C>
    MIVP_DEV void run(int pair, const bf16x8 (&a)[2][KS]) const {
        using XI = RowImg<C>;
        // channels of the pair: tile t covers n0 = 32 pair + 16 t + 4 g .. + 3 of the 3 C outputs
        long off[2];
        float sc[2];
        const FastDiv by_hd(hd);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n0 = 32 * pair + 16 * t + 4 * g;
            const int tensor = n0 >= 2 * C ? 2 : (n0 >= C ? 1 : 0), cc = n0 - tensor * C;
            const int head = by_hd.div(cc);
            off[t] = head * head_stride + (cc - head * hd) + sel(tensor == 0, 0L, sel(tensor == 1, to_k, to_v));
            sc[t] = tensor == 0 ? q_scale : (tensor == 1 ? MIVP_LOG2E : 1.0f);      // K carries log2(e): common.hpp
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            f32x4 c0 = fzero4(), c1 = fzero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 b = XI::frag(Ximg, 16 * tt + r, 32 * s + 8 * g);
                c0 = mfma16(a[0][s], b, c0);
                c1 = mfma16(a[1][s], b, c1);
            }
            // token r of token tile tt: head-0 element offset in a q / k / v tensor
            unsigned slot = slot0 + 16 * tt + r, bp = bp0;
            if (slot >= (unsigned)Nqp) { slot -= (unsigned)Nqp; ++bp; }
            const long tbase = ((long)bp * heads * Nqp + slot) * hd;
            if (row0 + 16 * tt + r < T) {
                st4(q + tbase + off[0], pack4(c0 * sc[0]));
                st4(q + tbase + off[1], pack4(c1 * sc[1]));
            }
        }
    }
};
template <int CT>
__global__ __launch_bounds__(256, CT <= 6 ? 3 : 2) void k_qkv_fwd_ws(MivpSwinDesc d, const bf16_t* __restrict__ x, const int* __restrict__ tok_src,
                                                        const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                        const bf16_t* __restrict__ wqkv, bf16_t* __restrict__ q,
                                                        bf16_t* __restrict__ k, bf16_t* __restrict__ v) {
    using G = WideGeom<CT>;
    constexpr int C = G::C, KS = G::KS, ROWS = 64, TPR = 4, XPT = C / 8 / TPR, NPAIR = 3 * CT / 2;
    using XI = RowImg<G::KP>;
    static_assert(G::KP == C && C % 32 == 0, "whole k-steps");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ximg = smem;                                           // [64][C]  LayerNorm output (the GEMM's B operand)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * ROWS;
    // the first pair's fragments do not depend on the rows: they travel during the gather
    bf16x8 a0[2][KS], a1[2][KS];
    ws_load_pair<KS>(wqkv, wave, lane, a0);
    {   // ---- rows: gather, LayerNorm ----
        const int row = tid / TPR, sub = tid % TPR;
        const RowTok ti = row_token(d, row0 + row);
        const int src = sel(ti.live, tok_src[ti.pw * d.Nqp + ti.slot], -2);
        const long xoff = (ti.b * d.vol_in + max(src, 0)) * (long)C;
        bf16x8 xr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) xr[i] = ld8(x + xoff + 8 * (sub + TPR * i));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            xr[i] = keep_if(xr[i], src >= 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)xr[i][e];
        }
        const float mean = row_sum<TPR>(sum) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dvv = (float)xr[i][e] - mean; var += dvv * dvv; }
        const float rstd = rsqrtf(row_sum<TPR>(var) / (float)C + d.ln_eps);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = 8 * (sub + TPR * i);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(ln_w + c), w1 = *reinterpret_cast<const f32x4*>(ln_w + c + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln_b + c), b1 = *reinterpret_cast<const f32x4*>(ln_b + c + 4);
            bf16x8 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (bf16_t)(((float)xr[i][e] - mean) * rstd * w0[e] + b0[e]);
                y[4 + e] = (bf16_t)(((float)xr[i][4 + e] - mean) * rstd * w1[e] + b1[e]);
            }
            XI::put8(Ximg, row, c, keep_if(y, src >= -1));       // -1: a zero-pad token still goes through LN (-> beta)
        }
    }
    const int hd = C / d.heads;
    const unsigned bp0 = (unsigned)(row0 / d.Nqp);
    __syncthreads();
    const WsOut<KS> out{Ximg, q, k - q, v - q, (long)d.Nqp * hd, row0, (long)d.B * d.P * d.Nqp, d.q_scale, hd, r, g, d.heads, d.Nqp,
                        bp0, (unsigned)(row0 - (long)bp0 * d.Nqp)};
    // pairs wave, wave + 4, ...: two register sets, the next pair's loads issued before the current pair is multiplied
    for (int pair = wave; pair < NPAIR; pair += 8) {
        if (pair + 4 < NPAIR) ws_load_pair<KS>(wqkv, pair + 4, lane, a1);
        out.template run<C>(pair, a0);
        if (pair + 4 < NPAIR) {
            if (pair + 8 < NPAIR) ws_load_pair<KS>(wqkv, pair + 8, lane, a0);
            out.template run<C>(pair + 4, a1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// proj + residual -> LayerNorm -> Linear + residual -> scatter          [k_swin_proj_mlp_fwd, swin_fwd.hip]
// (proj dropout, round 3: the counter-hash mask of element (token row, channel) -- the index of k_swin_proj_mlp_fwd and
//  mivp_dropout_masks -- applied to proj(o) + b in the first GEMM's epilogue; a uniform branch when the rate is zero)
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256, 3) void k_proj_mlp_fwd_wide(MivpSwinDesc d, const bf16_t* __restrict__ o, const bf16_t* __restrict__ x,
                                                              const int* __restrict__ tok_src, const int* __restrict__ tok_dst,
                                                              const bf16_t* __restrict__ wproj, const float* __restrict__ bproj,
                                                              const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                              const bf16_t* __restrict__ wmlp, const float* __restrict__ bmlp,
                                                              bf16_t* __restrict__ t1_out, bf16_t* __restrict__ y) {
    using G = WideGeom<CT>;
    constexpr int C = G::C, KS = G::KS, ROWS = G::ROWS, TPR = G::TPR, NCG = G::NCG, NCT = G::NCT, XPT = G::XPT;
    using RI = RowImg<G::KP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Aimg = smem;                                           // [ROWS][C]  o rows, later LN(t1) rows: the GEMMs' B operands
    char* Ximg = Aimg + ROWS * RI::ROWB;                         // [ROWS][C]  shortcut rows, then t1 rows, then the output rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * ROWS;
    const int row = tid / TPR, sub = tid % TPR;                  // the row phases' lane map
    const RowTok ti = row_token(d, row0 + row);
    const uint32_t proj_key = drop_seed(d.proj_seed, d.seed_epoch);      // (uniform; unused without dropout)
    constexpr bool FLAT = flat_rows<CT>();
    // C = 96: the wave's weight fragments of a GEMM are requested a phase EARLY (proj: before the rows come in, MLP: before the
    // LayerNorm) -- read in the k loop each GEMM started with an L2 round trip that nothing overlapped
    constexpr bool PREW = CT == 6;                   // (measured: C = 96 15.9 -> 15.0 us; C = 192 neutral at 46 more registers; C = 384 spills)
    bf16x8 wa[PREW ? NCT : 1][PREW ? KS : 1];
    if constexpr (PREW) {
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) wa[j][s] = wfrag(wproj, KS, (wave % NCG) + NCG * j, s, lane);
    }
    int src = -2, dst = -1;
    int fdst[XPT], fb[XPT];                                      // FLAT: scatter target and batch index of this thread's pieces
    if (!FLAT) {
        src = sel(ti.live, tok_src[ti.pw * d.Nqp + ti.slot], -2);
        dst = sel(ti.live, tok_dst[ti.pw * d.Nqp + ti.slot], -1);
    }
    if (FLAT) {   // ---- rows in: o, shortcut (piece-major) ----
        const FlatRows<CT> fr(d, row0, tid);
        int fsrc[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            fsrc[i] = sel(fr.tk[i].live, tok_src[fr.tk[i].pw * d.Nqp + fr.tk[i].slot], -2);
            fdst[i] = sel(fr.tk[i].live, tok_dst[fr.tk[i].pw * d.Nqp + fr.tk[i].slot], -1);
        }
        bf16x8 orow[XPT], xr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            orow[i] = ld8(o + fr.tk[i].tt * (long)C + fr.col[i]);
            xr[i] = ld8(x + (fr.tk[i].b * d.vol_in + max(fsrc[i], 0)) * (long)C + fr.col[i]);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            RI::put8(Aimg, fr.prow[i], fr.col[i], keep_if(orow[i], fr.tk[i].live));
            RI::put8(Ximg, fr.prow[i], fr.col[i], keep_if(xr[i], fsrc[i] >= 0));
            fb[i] = (int)fr.tk[i].b;
        }
        if (G::KP != C && sub == 0)
            for (int c = C; c < G::KP; c += 8) RI::put8(Aimg, row, c, zero8());          // the k-step padding of the B operand
    } else {   // ---- rows in: o, shortcut ----
        const long xoff = (ti.b * d.vol_in + max(src, 0)) * (long)C;
        bf16x8 orow[XPT], xr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            orow[i] = ld8(o + ti.tt * (long)C + 8 * (sub + TPR * i));
            xr[i] = ld8(x + xoff + 8 * (sub + TPR * i));
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            RI::put8(Aimg, row, 8 * (sub + TPR * i), keep_if(orow[i], ti.live));
            RI::put8(Ximg, row, 8 * (sub + TPR * i), keep_if(xr[i], src >= 0));
        }
        if (G::KP != C && sub == 0)
            for (int c = C; c < G::KP; c += 8) RI::put8(Aimg, row, c, zero8());          // the k-step padding of the B operand
    }
    __syncthreads();
    const int cg = wave % NCG, tg = wave / NCG;
    const int R0 = tg * 32 + r, R1 = R0 + 16;
    f32x4 t1[NCT][2];
    {   // ---- GEMM 1: t1 = o Wproj^T + b + shortcut (kept as bf16 values) ----
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int nt = cg + NCG * j, n0 = 16 * nt + 4 * g;
            const f32x4 bp4 = *reinterpret_cast<const f32x4*>(bproj + n0);
            f32x4 a0 = fzero4(), a1 = fzero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 a;
                if constexpr (PREW) a = wa[j][s]; else a = wfrag(wproj, KS, nt, s, lane);
                a0 = mfma16(a, RI::frag(Aimg, R0, 32 * s + 8 * g), a0);
                a1 = mfma16(a, RI::frag(Aimg, R1, 32 * s + 8 * g), a1);
            }
            const bf16x4 s0 = RI::get4(Ximg, R0, n0), s1 = RI::get4(Ximg, R1, n0);
            if (d.proj_drop_thr) {                               // proj dropout: on proj(o) + b, before the residual
                const uint32_t p0 = (uint32_t)(((row0 + R0) * C + n0) >> 1), p1 = (uint32_t)(((row0 + R1) * C + n0) >> 1);
                const uint32_t h00 = drop_hash(p0, proj_key), h01 = drop_hash(p0 + 1, proj_key);
                const uint32_t h10 = drop_hash(p1, proj_key), h11 = drop_hash(p1 + 1, proj_key);
                const float sc = d.proj_drop_scale;
                const float k0[4] = {drop_keep(h00, 0, d.proj_drop_thr) ? sc : 0.f, drop_keep(h00, 1, d.proj_drop_thr) ? sc : 0.f,
                                     drop_keep(h01, 0, d.proj_drop_thr) ? sc : 0.f, drop_keep(h01, 1, d.proj_drop_thr) ? sc : 0.f};
                const float k1[4] = {drop_keep(h10, 0, d.proj_drop_thr) ? sc : 0.f, drop_keep(h10, 1, d.proj_drop_thr) ? sc : 0.f,
                                     drop_keep(h11, 0, d.proj_drop_thr) ? sc : 0.f, drop_keep(h11, 1, d.proj_drop_thr) ? sc : 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] = (float)(bf16_t)((a0[e] + bp4[e]) * k0[e] + (float)s0[e]);
                    a1[e] = (float)(bf16_t)((a1[e] + bp4[e]) * k1[e] + (float)s1[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] = (float)(bf16_t)(a0[e] + bp4[e] + (float)s0[e]);
                    a1[e] = (float)(bf16_t)(a1[e] + bp4[e] + (float)s1[e]);
                }
            }
            t1[j][0] = a0;
            t1[j][1] = a1;
            RI::put4(Ximg, R0, n0, pack4(a0));                  // in place: this lane just read these eight bytes
            RI::put4(Ximg, R1, n0, pack4(a1));
        }
    }
    if constexpr (PREW) {
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) wa[j][s] = wfrag(wmlp, KS, cg + NCG * j, s, lane);
    }
    __syncthreads();
    {   // ---- rows: t1 out, LayerNorm(t1) -> Aimg ----
        float ts[XPT][8], sum = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const bf16x8 raw = RI::frag(Ximg, row, 8 * (sub + TPR * i));
            if (t1_out && ti.live) st8(t1_out + ti.tt * (long)C + 8 * (sub + TPR * i), raw);
            to_f32(raw, ts[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += ts[i][e];
        }
        const float mean = row_sum<TPR>(sum) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dvv = ts[i][e] - mean; var += dvv * dvv; }
        const float rstd = rsqrtf(row_sum<TPR>(var) / (float)C + d.ln_eps);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = 8 * (sub + TPR * i);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(ln_w + c), w1 = *reinterpret_cast<const f32x4*>(ln_w + c + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln_b + c), b1 = *reinterpret_cast<const f32x4*>(ln_b + c + 4);
            bf16x8 yy;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                yy[e] = (bf16_t)((ts[i][e] - mean) * rstd * w0[e] + b0[e]);
                yy[4 + e] = (bf16_t)((ts[i][4 + e] - mean) * rstd * w1[e] + b1[e]);
            }
            RI::put8(Aimg, row, c, yy);
        }
    }
    __syncthreads();
    {   // ---- GEMM 2: t2 = t1 + LN(t1) Wmlp^T + b ----
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int nt = cg + NCG * j, n0 = 16 * nt + 4 * g;
            const f32x4 bm4 = *reinterpret_cast<const f32x4*>(bmlp + n0);
            f32x4 a0 = fzero4(), a1 = fzero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 a;
                if constexpr (PREW) a = wa[j][s]; else a = wfrag(wmlp, KS, nt, s, lane);
                a0 = mfma16(a, RI::frag(Aimg, R0, 32 * s + 8 * g), a0);
                a1 = mfma16(a, RI::frag(Aimg, R1, 32 * s + 8 * g), a1);
            }
            RI::put4(Ximg, R0, n0, pack4(a0 + t1[j][0] + bm4));
            RI::put4(Ximg, R1, n0, pack4(a1 + t1[j][1] + bm4));
        }
    }
    __syncthreads();
    if (FLAT) {                                                  // ---- rows out: scatter (piece-major) ----
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int P = tid + 256 * i, prow = P / (C / 8), col = 8 * (P - prow * (C / 8));
            if (fdst[i] >= 0) st8(y + ((long)fb[i] * d.vol_out + fdst[i]) * (long)C + col, RI::frag(Ximg, prow, col));
        }
    } else if (dst >= 0) {                                       // ---- rows out: scatter ----
        const long yoff = (ti.b * d.vol_out + dst) * (long)C;
#pragma unroll
        for (int i = 0; i < XPT; ++i) st8(y + yoff + 8 * (sub + TPR * i), RI::frag(Ximg, row, 8 * (sub + TPR * i)));
    }
}

// ---------------------------------------------------------------------------------------------
// proj + MLP backward:  dy -> (dO, dt1)                                 [k_swin_proj_mlp_bwd, swin_bwd.hip]
// (no weight-gradient outputs here: the launcher leaves those calls to the 16-token kernel.  Proj dropout, round 3: the residual
//  branch sees dt1 as it is; the operand of dO = dt1 Wproj is dt1 under the forward's mask, kept in the dy image's place)
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256, 3) void k_proj_mlp_bwd_wide(MivpSwinDesc d, const bf16_t* __restrict__ dy, const int* __restrict__ tok_dst,
                                                              const bf16_t* __restrict__ t1, const float* __restrict__ ln_w,
                                                              const bf16_t* __restrict__ wmlp_t, const bf16_t* __restrict__ wproj_t,
                                                              bf16_t* __restrict__ d_o, bf16_t* __restrict__ d_t1) {
    using G = WideGeom<CT>;
    constexpr int C = G::C, KS = G::KS, ROWS = G::ROWS, TPR = G::TPR, NCG = G::NCG, NCT = G::NCT, XPT = G::XPT;
    using RI = RowImg<G::KP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Dimg = smem;                                           // [ROWS][C]  dy rows (B of GEMM A), later the dO rows
    char* Timg = Dimg + ROWS * RI::ROWB;                         // [ROWS][C]  t1 rows, then dt1 rows (B of GEMM B)
    float* stat = reinterpret_cast<float*>(Timg + ROWS * RI::ROWB);      // [ROWS][2] mean, rstd of t1
    float* red = stat + 2 * ROWS;                                // [ROWS][NCG][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * ROWS;
    const int row = tid / TPR, sub = tid % TPR;
    const RowTok ti = row_token(d, row0 + row);
    const uint32_t proj_key = drop_seed(d.proj_seed, d.seed_epoch);      // (uniform; unused without dropout)
    const bool drop = d.proj_drop_thr != 0;
    {   // ---- rows in: dy (gathered), t1 + its LayerNorm statistics ----
        const int dst = sel(ti.live, tok_dst[ti.pw * d.Nqp + ti.slot], -1);
        const long yoff = (ti.b * d.vol_out + max(dst, 0)) * (long)C;
        bf16x8 dr[XPT], tr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            dr[i] = ld8(dy + yoff + 8 * (sub + TPR * i));
            tr[i] = ld8(t1 + ti.tt * (long)C + 8 * (sub + TPR * i));
        }
        float ts[XPT][8], sum = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const bf16x8 tv = keep_if(tr[i], ti.live);
            RI::put8(Dimg, row, 8 * (sub + TPR * i), keep_if(dr[i], dst >= 0));
            RI::put8(Timg, row, 8 * (sub + TPR * i), tv);
            to_f32(tv, ts[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += ts[i][e];
        }
        const float mean = row_sum<TPR>(sum) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dvv = ts[i][e] - mean; var += dvv * dvv; }
        const float rstd = rsqrtf(row_sum<TPR>(var) / (float)C + d.ln_eps);
        if (sub == 0) {
            stat[2 * row] = mean;
            stat[2 * row + 1] = rstd;
            for (int c = C; c < G::KP; c += 8) { RI::put8(Dimg, row, c, zero8()); RI::put8(Timg, row, c, zero8()); }     // k-step padding
        }
    }
    __syncthreads();
    const int cg = wave % NCG, tg = wave / NCG;
    const int R0 = tg * 32 + r, R1 = R0 + 16;
    {   // ---- GEMM A: dh = dy Wmlp, LayerNorm backward, dt1 = dy + ... ----
        f32x4 dh[NCT][2], xh[NCT][2];
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        const float mean0 = stat[2 * R0], rstd0 = stat[2 * R0 + 1], mean1 = stat[2 * R1], rstd1 = stat[2 * R1 + 1];
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int nt = cg + NCG * j, n0 = 16 * nt + 4 * g;
            const f32x4 lw = *reinterpret_cast<const f32x4*>(ln_w + n0);
            f32x4 a0 = fzero4(), a1 = fzero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 a = wfrag(wmlp_t, KS, nt, s, lane);
                a0 = mfma16(a, RI::frag(Dimg, R0, 32 * s + 8 * g), a0);
                a1 = mfma16(a, RI::frag(Dimg, R1, 32 * s + 8 * g), a1);
            }
            const bf16x4 t0 = RI::get4(Timg, R0, n0), t1v = RI::get4(Timg, R1, n0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = a0[e] * lw[e], d1 = a1[e] * lw[e];
                const float h0 = ((float)t0[e] - mean0) * rstd0, h1 = ((float)t1v[e] - mean1) * rstd1;
                a0[e] = d0; a1[e] = d1;
                xh[j][0][e] = h0; xh[j][1][e] = h1;
                s1[0] += d0; s2[0] += d0 * h0;
                s1[1] += d1; s2[1] += d1 * h1;
            }
            dh[j][0] = a0;
            dh[j][1] = a1;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float a = col_sum(s1[u]), b = col_sum(s2[u]);
            if (g == 0) { float* dstp = red + ((u ? R1 : R0) * NCG + cg) * 2; dstp[0] = a; dstp[1] = b; }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int R = u ? R1 : R0;
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int c = 0; c < NCG; ++c) { m1 += red[(R * NCG + c) * 2]; m2 += red[(R * NCG + c) * 2 + 1]; }
            m1 /= (float)C;
            m2 /= (float)C;
            const float rstd = u ? rstd1 : rstd0;
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int n0 = 16 * (cg + NCG * j) + 4 * g;
                const bf16x4 dyv = RI::get4(Dimg, R, n0);
                f32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = (float)dyv[e] + rstd * (dh[j][u][e] - m1 - xh[j][u][e] * m2);
                const bf16x4 outb = pack4(out);
                RI::put4(Timg, R, n0, outb);                    // in place: this lane read these eight bytes of t1 above
                if (drop) {
                    // the GEMM B operand under the forward's mask, in the place of the dy piece this lane just read (from the
                    // f32 values, then rounded, as k_swin_proj_mlp_bwd does)
                    const uint32_t pi = (uint32_t)(((row0 + R) * C + n0) >> 1);
                    const uint32_t h0 = drop_hash(pi, proj_key), h1 = drop_hash(pi + 1, proj_key);
                    f32x4 mv;
                    mv[0] = drop_keep(h0, 0, d.proj_drop_thr) ? out[0] * d.proj_drop_scale : 0.f;
                    mv[1] = drop_keep(h0, 1, d.proj_drop_thr) ? out[1] * d.proj_drop_scale : 0.f;
                    mv[2] = drop_keep(h1, 0, d.proj_drop_thr) ? out[2] * d.proj_drop_scale : 0.f;
                    mv[3] = drop_keep(h1, 1, d.proj_drop_thr) ? out[3] * d.proj_drop_scale : 0.f;
                    RI::put4(Dimg, R, n0, pack4(mv));
                }
            }
        }
    }
    __syncthreads();
    // ---- rows out: dt1 (the residual branch) ----
    if (ti.live) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) st8(d_t1 + ti.tt * (long)C + 8 * (sub + TPR * i), RI::frag(Timg, row, 8 * (sub + TPR * i)));
    }
    {   // ---- GEMM B: dO = dt1 Wproj ----
        f32x4 acc[NCT][2];
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int nt = cg + NCG * j;
            f32x4 a0 = fzero4(), a1 = fzero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 a = wfrag(wproj_t, KS, nt, s, lane);
                const char* Bimg = drop ? Dimg : Timg;
                a0 = mfma16(a, RI::frag(Bimg, R0, 32 * s + 8 * g), a0);
                a1 = mfma16(a, RI::frag(Bimg, R1, 32 * s + 8 * g), a1);
            }
            acc[j][0] = a0;
            acc[j][1] = a1;
        }
        // Dimg was last read (dy) before the barrier above: free for the dO rows -- unless it holds the masked operand, which every
        // wave must have finished reading first
        if (drop) __syncthreads();
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int n0 = 16 * (cg + NCG * j) + 4 * g;
            RI::put4(Dimg, R0, n0, pack4(acc[j][0]));
            RI::put4(Dimg, R1, n0, pack4(acc[j][1]));
        }
    }
    __syncthreads();
    if (ti.live) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) st8(d_o + ti.tt * (long)C + 8 * (sub + TPR * i), RI::frag(Dimg, row, 8 * (sub + TPR * i)));
    }
}

// ---------------------------------------------------------------------------------------------
// QKV + LayerNorm + gather backward:  (dq, dk, dv, dt1) -> dx           [k_swin_qkv_bwd, swin_bwd.hip]
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256, CT <= 12 ? 3 : 2) void k_qkv_bwd_wide(MivpSwinDesc d, const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk,
                                                         const bf16_t* __restrict__ dv, const bf16_t* __restrict__ x,
                                                         const int* __restrict__ tok_src, const float* __restrict__ ln_w,
                                                         const bf16_t* __restrict__ wqkv_t, const bf16_t* __restrict__ d_t1,
                                                         bf16_t* __restrict__ dx, bf16_t* __restrict__ dn_out) {
    using G = WideGeom<CT>;
    constexpr int C = G::C, K3 = 3 * C, K3P = (K3 + 31) / 32 * 32, KS3 = K3P / 32, ROWS = G::ROWS, TPR = G::TPR, NCG = G::NCG,
                  NCT = G::NCT, XPT = G::XPT;
    using BI = RowImg<K3P>;
    using RI = RowImg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Bimg = smem;                                                   // [ROWS][3C] = dq * scale | dk | dv of the token
    char* Ximg = Bimg + ROWS * BI::ROWB;                                 // [ROWS][C]  x rows, then the dx rows
    float* stat = reinterpret_cast<float*>(Ximg + ROWS * RI::ROWB);      // [ROWS][2] mean, rstd of x
    float* red = stat + 2 * ROWS;                                        // [ROWS][NCG][2] partial s1, s2
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * ROWS;
    const int row = tid / TPR, sub = tid % TPR;
    const RowTok ti = row_token(d, row0 + row);
    const int src = sel(ti.live, tok_src[ti.pw * d.Nqp + ti.slot], -2);
    const long xoff = (ti.b * d.vol_in + max(src, 0)) * (long)C;

    // ---- head-major gradient chunks -> Bimg (loads first, the q part scaled on the way) ----
    {
        const HeadMajor<C, G::NTG> hm(d, row0);
        const long to_dk = dk - dq, to_dv = dv - dq;
        bf16x4 pc[HeadMajor<C, G::NTG>::PPT];
        hm.each(tid, [&](int i, long off, int tensor, int, int, bool) {
            pc[i] = ld4(dq + sel(tensor == 0, 0L, sel(tensor == 1, to_dk, to_dv)) + off);
        });
        // rows: x (LayerNorm statistics + image)
        bf16x8 xr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) xr[i] = ld8(x + xoff + 8 * (sub + TPR * i));
        hm.each(tid, [&](int i, long, int tensor, int irow, int icol, bool live) {
            bf16x4 val = keep_if(pc[i], live);
            const float sc = sel(tensor == 0, d.q_scale, 1.0f);           // the q part carries the attention scale (x 1 is exact)
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = (bf16_t)((float)val[j] * sc);
            BI::put4(Bimg, irow, icol, val);
        });
        float xs[XPT][8], sum = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const bf16x8 xv = keep_if(xr[i], src >= 0);
            RI::put8(Ximg, row, 8 * (sub + TPR * i), xv);
            to_f32(xv, xs[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += xs[i][e];
        }
        const float mean = row_sum<TPR>(sum) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dvv = xs[i][e] - mean; var += dvv * dvv; }
        const float rstd = rsqrtf(row_sum<TPR>(var) / (float)C + d.ln_eps);
        if (sub == 0) {
            stat[2 * row] = mean;
            stat[2 * row + 1] = rstd;
            for (int c = K3; c < K3P; c += 8) BI::put8(Bimg, row, c, zero8());           // the k-step padding of the B operand
        }
    }
    __syncthreads();

    // ---- GEMM: dn[token][c] = sum_n B[token][n] * wqkv_t[c][n], this wave's column tiles, two token tiles ----
    const int cg = wave % NCG, tg = wave / NCG;
    const int R0 = tg * 32 + r, R1 = R0 + 16;                             // this lane's token rows of the two tiles
    const long T = (long)d.B * d.P * d.Nqp;
    // dt1 in the accumulator lane map, straight from global (an LDS image of it would cost the third workgroup per CU:
    // the 704 workgroups of a 12 x 12 x 24-token stage then run as two rounds); issued before the GEMM
    bf16x4 tq[NCT][2];
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long tt = row0 + (u ? R1 : R0);
            tq[j][u] = ld4(d_t1 + sel(tt < T, tt, 0L) * C + 16 * (cg + NCG * j) + 4 * g);
        }
    f32x4 acc[NCT][2];
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
        f32x4 a0 = fzero4(), a1 = fzero4();
#pragma unroll
        for (int s = 0; s < KS3; ++s) {
            const bf16x8 a = wfrag(wqkv_t, KS3, cg + NCG * j, s, lane);       // fragment image: 1 KB contiguous per wave
            a0 = mfma16(a, BI::frag(Bimg, R0, 32 * s + 8 * g), a0);
            a1 = mfma16(a, BI::frag(Bimg, R1, 32 * s + 8 * g), a1);
        }
        acc[j][0] = a0;
        acc[j][1] = a1;
    }

    // ---- LayerNorm backward (rows reduced over the column groups through LDS) ----
    const float mean0 = stat[2 * R0], rstd0 = stat[2 * R0 + 1], mean1 = stat[2 * R1], rstd1 = stat[2 * R1 + 1];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    f32x4 xh[NCT][2];
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
        const int c0 = 16 * (cg + NCG * j) + 4 * g;
        const f32x4 lw = *reinterpret_cast<const f32x4*>(ln_w + c0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int R = u ? R1 : R0;
            if (dn_out && row0 + R < T) st4(dn_out + (row0 + R) * C + c0, pack4(acc[j][u]));      // (weight-gradient mode)
            const bf16x4 raw = RI::get4(Ximg, R, c0);
            const float mean = u ? mean1 : mean0, rstd = u ? rstd1 : rstd0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dxh = acc[j][u][e] * lw[e];
                const float h = ((float)raw[e] - mean) * rstd;
                acc[j][u][e] = dxh;
                xh[j][u][e] = h;
                s1[u] += dxh;
                s2[u] += dxh * h;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float a = col_sum(s1[u]), b = col_sum(s2[u]);
        if (g == 0) { float* dstp = red + ((u ? R1 : R0) * NCG + cg) * 2; dstp[0] = a; dstp[1] = b; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int R = u ? R1 : R0;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCG; ++c) { m1 += red[(R * NCG + c) * 2]; m2 += red[(R * NCG + c) * 2 + 1]; }
        m1 /= (float)C;
        m2 /= (float)C;
        const float rstd = u ? rstd1 : rstd0;
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int c0 = 16 * (cg + NCG * j) + 4 * g;
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = (float)tq[j][u][e] + rstd * (acc[j][u][e] - m1 - xh[j][u][e] * m2);
            RI::put4(Ximg, R, c0, pack4(out));                  // in place: this lane read these eight bytes of x above
        }
    }
    __syncthreads();
    if (src >= 0) {                                              // ---- rows out (zero-pad / padding-slot tokens have no voxel) ----
#pragma unroll
        for (int i = 0; i < XPT; ++i) st8(dx + xoff + 8 * (sub + TPR * i), RI::frag(Ximg, row, 8 * (sub + TPR * i)));
    }
}

template <int CT> size_t lds_qkv_fwd() {
    using G = WideGeom<CT>;
    const size_t xi = (size_t)G::ROWS * RowImg<G::KP>::ROWB, oi = (size_t)G::ROWS * RowImg<3 * G::C>::ROWB;
    return G::NCG == 1 ? (xi > oi ? xi : oi) : xi + oi;
}
template <int CT> size_t lds_proj_mlp_fwd() { using G = WideGeom<CT>; return (size_t)G::ROWS * 2 * RowImg<G::KP>::ROWB; }
template <int CT> size_t lds_proj_mlp_bwd() {
    using G = WideGeom<CT>;
    return (size_t)G::ROWS * 2 * RowImg<G::KP>::ROWB + (2 * G::ROWS + 2 * G::ROWS * G::NCG) * sizeof(float);
}
template <int CT> size_t lds_qkv_bwd() {
    using G = WideGeom<CT>;
    return (size_t)G::ROWS * (RowImg<(3 * G::C + 31) / 32 * 32>::ROWB + RowImg<G::C>::ROWB) + (2 * G::ROWS + 2 * G::ROWS * G::NCG) * sizeof(float);
}

// w row-major [rows][cols] -> fragment image [rows/16 (up)][k_steps][64 lanes][8]
__global__ __launch_bounds__(256) void k_pack_weight_frags(const bf16_t* __restrict__ w, int rows, int cols, int k_steps, int paired,
                                                           bf16_t* __restrict__ out, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long frag = i >> 9;
        const int s = (int)(frag % k_steps), nt = (int)(frag / k_steps);
        const int r = lane & 15, g = lane >> 4;
        const int row = 16 * nt + r;
        const int col = 32 * s + (paired ? 16 * (e >> 2) + 4 * g + (e & 3) : 8 * g + e);
        out[i] = (row < rows && col < cols) ? w[(long)row * cols + col] : (bf16_t)0.0f;
    }
}

// All kernel-ready images of one Swin block's five weight matrices in ONE launch (a training step where every parameter
// changes rebuilt them with ~25 launches per block: transposes, casts, concatenations, one pack per image).
struct PackJob {
    bf16_t* out;          // destination
    long elems;           // number of output elements
    int src;              // 0: [wq; wk; wv] stacked [3C][C], 1: wproj, 2: wmlp
    int transposed;       // image of the transpose
    int kind;             // 0: row-major bf16 copy, 1: natural fragment image, 2: paired fragment image
    int rows, cols;       // of the (possibly transposed) matrix the image describes
    int k_steps;
};
struct PackJobs { PackJob job[12]; int n; int C; const float* w[5]; };

__global__ __launch_bounds__(256) void k_pack_block_weights(PackJobs jobs) {
    const int C = jobs.C;
    long total = 0;
    for (int j = 0; j < jobs.n; ++j) total += jobs.job[j].elems;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long rest = i;
        int j = 0;
        while (rest >= jobs.job[j].elems) { rest -= jobs.job[j].elems; ++j; }
        const PackJob& jb = jobs.job[j];
        int row, col;
        if (jb.kind == 0) {
            row = (int)(rest / jb.cols);
            col = (int)(rest - (long)row * jb.cols);
        } else {
            const int e = (int)(rest & 7), lane = (int)((rest >> 3) & 63);
            const long frag = rest >> 9;
            const int s = (int)(frag % jb.k_steps), nt = (int)(frag / jb.k_steps);
            row = 16 * nt + (lane & 15);
            const int g = lane >> 4;
            col = 32 * s + (jb.kind == 2 ? 16 * (e >> 2) + 4 * g + (e & 3) : 8 * g + e);
        }
        float val = 0.f;
        if (row < jb.rows && col < jb.cols) {
            const int r0 = jb.transposed ? col : row, c0 = jb.transposed ? row : col;      // element of the untransposed matrix
            const float* m = jb.src == 0 ? jobs.w[r0 / C] : jobs.w[2 + jb.src];
            val = m[(long)(jb.src == 0 ? r0 % C : r0) * C + c0];
        }
        jb.out[rest] = (bf16_t)val;
    }
}

}  // namespace

/* wq, wk, wv, wproj, wmlp: f32 [C][C] row-major ([out][in]).  Outputs (bf16; any may be NULL = skipped):
 *   wqkv_rm  [3C][C] row-major (prompt K/V kernels)      wqkv_f   natural image of [3C][C]
 *   wproj_f  natural image                                wmlp_f   paired image (+ natural behind it when with_natural)
 *   wqkv_t   natural image of [C][3C], k_steps = ceil(3 * 16 * ceil(C / 16) / 32)
 *   wmlp_t   natural image of Wmlp^T                      wproj_t  paired (+ natural) image of Wproj^T               */
extern "C" int mivp_pack_block_weights(int32_t C, const float* wq, const float* wk, const float* wv, const float* wproj,
                                       const float* wmlp, int32_t with_natural, void* wqkv_rm, void* wqkv_f, void* wproj_f,
                                       void* wmlp_f, void* wqkv_t, void* wmlp_t, void* wproj_t, mivp_stream_t stream) {
    MIVP_REQUIRE(C > 0 && wq && wk && wv && wproj && wmlp);
    PackJobs jobs;
    jobs.n = 0;
    jobs.C = C;
    jobs.w[0] = wq; jobs.w[1] = wk; jobs.w[2] = wv; jobs.w[3] = wproj; jobs.w[4] = wmlp;
    const int ct = (C + 15) / 16, ks = (C + 31) / 32, ks3 = (3 * 16 * ct + 31) / 32, ct3 = (3 * C + 15) / 16;
    auto add = [&](void* out, int src, int transposed, int kind, int rows, int cols, int k_steps) {
        if (!out) return;
        PackJob& j = jobs.job[jobs.n++];
        j.out = (bf16_t*)out; j.src = src; j.transposed = transposed; j.kind = kind; j.rows = rows; j.cols = cols; j.k_steps = k_steps;
        j.elems = kind == 0 ? (long)rows * cols : (long)((rows + 15) / 16) * k_steps * 512;
    };
    const long img = (long)ct * ks * 512;
    add(wqkv_rm, 0, 0, 0, 3 * C, C, 0);
    add(wqkv_f, 0, 0, 1, 3 * C, C, ks);
    add(wproj_f, 1, 0, 1, C, C, ks);
    add(wmlp_f, 2, 0, 2, C, C, ks);
    if (with_natural && wmlp_f) add((bf16_t*)wmlp_f + img, 2, 0, 1, C, C, ks);
    add(wqkv_t, 0, 1, 1, C, 3 * C, ks3);
    add(wmlp_t, 2, 1, 1, C, C, ks);
    add(wproj_t, 1, 1, 2, C, C, ks);
    if (with_natural && wproj_t) add((bf16_t*)wproj_t + img, 1, 1, 1, C, C, ks);
    (void)ct3;
    long total = 0;
    for (int j = 0; j < jobs.n; ++j) total += jobs.job[j].elems;
    if (total == 0) return MIVP_OK;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_pack_block_weights, dim3(grid), dim3(256), 0, (hipStream_t)stream, jobs);
    return mivp_check_launch("pack_block_weights");
}

extern "C" int mivp_pack_weight_frags(const void* w, int32_t rows, int32_t cols, int32_t k_steps, int32_t paired, void* out,
                                      mivp_stream_t stream) {
    MIVP_REQUIRE(w && out && rows > 0 && cols > 0 && k_steps >= (cols + 31) / 32);
    const long total = (long)((rows + 15) / 16) * k_steps * 512;
    const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_pack_weight_frags, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, rows, cols, k_steps,
                       paired, (bf16_t*)out, total);
    return mivp_check_launch("pack_weight_frags");
}

// C = 96 / 192 / 384, windows of a multiple of 32 slots (a 32-token granule then never leaves its window)
int mivp_tok_wide_supported(const MivpSwinDesc* d) {
    const int hd = d->C / d->heads;
    if (!(d->C == 48 || d->C == 96 || d->C == 192 || d->C == 384)) return 0;
    static const bool off = getenv("MIVP_NO_WIDE_TOKEN_KERNELS") != nullptr;     // A/B switch for profiling and tests
    if (off) return 0;
    // C = 48: the QKV pair measured SLOWER in this form (forward 55 vs 50 us, backward 88 vs 77 us at 48^3 x 4: the
    // head-major piece decode, the LDS round trip and half the occupancy cost more than the contiguous stores save); opt-in
    if (d->C == 48 && !getenv("MIVP_C48_QKV_ROW_KERNELS")) return 0;
    return (hd % 4 == 0 && d->Nqp % 32 == 0 && 8 * hd < 65536 && 72 * d->C < 65536) ? 1 : 0;
}
// the proj / MLP pair in the same form also at C = 48 (one column group: a wave owns its 32 tokens' three column tiles)
int mivp_tok_rows_supported(const MivpSwinDesc* d) {
    if (mivp_tok_wide_supported(d)) return 1;
    static const bool off = getenv("MIVP_NO_WIDE_TOKEN_KERNELS") != nullptr;
    return (!off && d->C == 48) ? 1 : 0;
}
// element offset of the natural-order image behind the paired one (swin_ops.paired_and_natural)
long mivp_tok_natural_offset(int C) { return (long)(C / 16) * ((C + 31) / 32) * 512; }

#define ROWS_SWITCH(LAUNCH)                                                                  \
    switch (d->C) {                                                                          \
        case 48: LAUNCH(3); break;                                                           \
        case 96: LAUNCH(6); break;                                                           \
        case 192: LAUNCH(12); break;                                                         \
        case 384: LAUNCH(24); break;                                                         \
        default: mivp_set_error("tok_rows: C not in {48, 96, 192, 384}"); return MIVP_EUNSUPPORTED; \
    }
#define WIDE_SWITCH(LAUNCH)                                                                  \
    switch (d->C) {                                                                          \
        case 96: LAUNCH(6); break;                                                           \
        case 192: LAUNCH(12); break;                                                         \
        case 384: LAUNCH(24); break;                                                         \
        default: mivp_set_error("tok_wide: C not in {96, 192, 384}"); return MIVP_EUNSUPPORTED; \
    }
#define WIDE_GRID(CTV) dim3((unsigned)((T + WideGeom<CTV>::ROWS - 1) / WideGeom<CTV>::ROWS))

int mivp_tok_wide_qkv_fwd(const MivpSwinDesc* d, const void* x, const int32_t* tok_src, const float* ln_w, const float* ln_b,
                          const void* wqkv, void* q, void* k, void* v, hipStream_t st) {
    const long T = (long)d->B * d->P * d->Nqp;
#define L_W(CTV)                                                                                                             \
    do {                                                                                                                     \
        const size_t lds = lds_qkv_fwd<CTV>();                                                                               \
        MIVP_LDS_OPT_IN(k_qkv_fwd_wide<CTV>, lds);                                                                           \
        hipLaunchKernelGGL((k_qkv_fwd_wide<CTV>), WIDE_GRID(CTV), dim3(256), lds, st, *d, (const bf16_t*)x, tok_src, ln_w,   \
                           ln_b, (const bf16_t*)wqkv, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v);                                   \
    } while (0)
    static const bool no_ws = getenv("MIVP_QKV_FWD_STREAMED_WEIGHTS") != nullptr;      // A/B switch: the round-2 work split
    const int hd = d->C / d->heads;
    if (!no_ws && (d->C == 96 || d->C == 192) && hd % 4 == 0 && d->Nqp % 16 == 0 && d->Nqp >= 64) {
#define L_WS(CTV)                                                                                                            \
    do {                                                                                                                     \
        const size_t lds = (size_t)64 * RowImg<WideGeom<CTV>::KP>::ROWB;                                                     \
        hipLaunchKernelGGL((k_qkv_fwd_ws<CTV>), dim3((unsigned)((T + 63) / 64)), dim3(256), lds, st, *d, (const bf16_t*)x,   \
                           tok_src, ln_w, ln_b, (const bf16_t*)wqkv, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v);                    \
    } while (0)
        if (d->C == 96) L_WS(6); else L_WS(12);
#undef L_WS
        return mivp_check_launch("swin_qkv_fwd(weight-stationary)");
    }
    ROWS_SWITCH(L_W)
#undef L_W
    return mivp_check_launch("swin_qkv_fwd(wide)");
}

int mivp_tok_wide_proj_mlp_fwd(const MivpSwinDesc* d, const void* o, const void* x, const int32_t* tok_src, const int32_t* tok_dst,
                               const void* wproj, const float* bproj, const float* ln_w, const float* ln_b, const void* wmlp,
                               const float* bmlp, void* t1_out, void* y, hipStream_t st) {
    const long T = (long)d->B * d->P * d->Nqp;
#define L_W(CTV)                                                                                                              \
    do {                                                                                                                      \
        const size_t lds = lds_proj_mlp_fwd<CTV>();                                                                           \
        MIVP_LDS_OPT_IN(k_proj_mlp_fwd_wide<CTV>, lds);                                                                       \
        hipLaunchKernelGGL((k_proj_mlp_fwd_wide<CTV>), WIDE_GRID(CTV), dim3(256), lds, st, *d, (const bf16_t*)o,              \
                           (const bf16_t*)x, tok_src, tok_dst, (const bf16_t*)wproj, bproj, ln_w, ln_b, (const bf16_t*)wmlp,  \
                           bmlp, (bf16_t*)t1_out, (bf16_t*)y);                                                                \
    } while (0)
    ROWS_SWITCH(L_W)
#undef L_W
    return mivp_check_launch("swin_proj_mlp_fwd(wide)");
}

int mivp_tok_wide_proj_mlp_bwd(const MivpSwinDesc* d, const void* dy, const int32_t* tok_dst, const void* t1, const float* ln_w,
                               const void* wmlp_t, const void* wproj_t, void* d_o, void* d_t1, hipStream_t st) {
    const long T = (long)d->B * d->P * d->Nqp;
#define L_W(CTV)                                                                                                             \
    do {                                                                                                                     \
        const size_t lds = lds_proj_mlp_bwd<CTV>();                                                                          \
        MIVP_LDS_OPT_IN(k_proj_mlp_bwd_wide<CTV>, lds);                                                                      \
        hipLaunchKernelGGL((k_proj_mlp_bwd_wide<CTV>), WIDE_GRID(CTV), dim3(256), lds, st, *d, (const bf16_t*)dy, tok_dst,   \
                           (const bf16_t*)t1, ln_w, (const bf16_t*)wmlp_t, (const bf16_t*)wproj_t, (bf16_t*)d_o,             \
                           (bf16_t*)d_t1);                                                                                   \
    } while (0)
    ROWS_SWITCH(L_W)
#undef L_W
    return mivp_check_launch("swin_proj_mlp_bwd(wide)");
}

int mivp_tok_wide_qkv_bwd(const MivpSwinDesc* d, const void* dq, const void* dk, const void* dv, const void* x, const int32_t* tok_src,
                          const float* ln_w, const void* wqkv_t, const void* d_t1, void* dx, void* dn_out, hipStream_t st) {
    const long T = (long)d->B * d->P * d->Nqp;
#define L_W(CTV)                                                                                                             \
    do {                                                                                                                     \
        const size_t lds = lds_qkv_bwd<CTV>();                                                                               \
        MIVP_LDS_OPT_IN(k_qkv_bwd_wide<CTV>, lds);                                                                           \
        hipLaunchKernelGGL((k_qkv_bwd_wide<CTV>), WIDE_GRID(CTV), dim3(256), lds, st, *d, (const bf16_t*)dq, (const bf16_t*)dk, \
                           (const bf16_t*)dv, (const bf16_t*)x, tok_src, ln_w, (const bf16_t*)wqkv_t, (const bf16_t*)d_t1,   \
                           (bf16_t*)dx, (bf16_t*)dn_out);                                                                    \
    } while (0)
    ROWS_SWITCH(L_W)
#undef L_W
    return mivp_check_launch("swin_qkv_bwd(wide)");
}
