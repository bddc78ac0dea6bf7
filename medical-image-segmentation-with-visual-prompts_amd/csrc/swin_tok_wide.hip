// Token GEMM kernels of the WIDE stages (C = 96, 192, 384: encoder stages 1-3 and the decoder blocks of the same widths).
// Same operators as k_swin_qkv_fwd / k_swin_proj_mlp_fwd (swin_fwd.hip) and k_swin_proj_mlp_bwd / k_swin_qkv_bwd
// (swin_bwd.hip) -- reference swin_transformer/swin_block.py:145-255 and its autograd -- in another work split.
//
// Why: the one-wave-per-16-tokens form gives the deep stages a few hundred workgroups whose single wave per SIMD walks
// EVERY column tile (C = 192: 3100 VALU instructions and ~30 dependent load round trips per wave, 296 VGPRs -> one
// workgroup per CU, two rounds; r02 PMC: 47 us for 2.5 GFLOP).  With so few waves nothing overlaps, so the kernel time is
// (instructions + round trips) PER WAVE.  Here a workgroup owns 32 tokens per token group and its waves split the output
// COLUMNS:
//   * NCG column groups x NTG token groups = 4 waves (CT % 4 == 0: 4 x 1, 32 tokens; otherwise 2 x 2, 64 tokens);
//     wave (cg, tg) computes column tiles cg, cg + NCG, ... for the two 16-token tiles of token group tg, so every weight
//     fragment (straight from L2, no slab barriers) feeds two MFMAs;
//   * everything that is per TOKEN is done once per workgroup by 256 / rows threads per token row and shared through LDS:
//     the gather of the GEMM's B operand (row image, ds_read_b128 fragments), LayerNorm statistics, source / destination
//     voxels;
//   * row reductions that span column groups (LayerNorm-backward sums, LayerNorm statistics of a GEMM output) meet in LDS
//     in column-group order (fixed order: bit-reproducible);
//   * every global load is unconditional (common.hpp "Branch-free loads").
// The launchers at the bottom are called by the C entries of swin_fwd.hip / swin_bwd.hip when mivp_tok_wide_supported().
#include "common.hpp"
#include <cstdlib>

namespace {

template <int CT>
struct WideGeom {
    static constexpr int C = 16 * CT;
    static constexpr int NCG = (CT % 4 == 0) ? 4 : 2;           // column groups
    static constexpr int NTG = 4 / NCG;                          // token groups
    static constexpr int NCT = CT / NCG;                         // column tiles per wave
    static constexpr int ROWS = 32 * NTG;                        // token rows per workgroup
    static constexpr int TPR = 256 / ROWS;                       // threads per token row in the per-token phases
    static_assert(CT % NCG == 0, "column tiles must split evenly");
};

// LDS row image of a GEMM B operand: [rows][K] bf16, +16 B per row (the 16 rows of a fragment read then cover all banks)
template <int K>
struct RowImg {
    static constexpr int ROWB = 2 * K + 16;
    static MIVP_DEV bf16x8 frag(const char* img, int row, int k0) { return *reinterpret_cast<const bf16x8*>(img + row * ROWB + 2 * k0); }
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
// sum over the TPR consecutive lanes that share a token row
template <int TPR>
MIVP_DEV float row_sum(float v) {
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// QKV + LayerNorm + gather backward:  (dq, dk, dv, dt1) -> dx        [k_swin_qkv_bwd, swin_bwd.hip]
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256, 3) void k_qkv_bwd_wide(MivpSwinDesc d, const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk,
                                                         const bf16_t* __restrict__ dv, const bf16_t* __restrict__ x,
                                                         const int* __restrict__ tok_src, const float* __restrict__ ln_w,
                                                         const bf16_t* __restrict__ wqkv_t, const bf16_t* __restrict__ d_t1,
                                                         bf16_t* __restrict__ dx, bf16_t* __restrict__ dn_out) {
    using G = WideGeom<CT>;
    constexpr int C = G::C, K3 = 3 * C, KS3 = K3 / 32, ROWS = G::ROWS, TPR = G::TPR, NCG = G::NCG, NCT = G::NCT;
    using BI = RowImg<K3>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Bimg = smem;                                                   // [ROWS][3C] = dq * scale | dk | dv of the token
    float* stat = reinterpret_cast<float*>(Bimg + ROWS * BI::ROWB);      // [ROWS][2] mean, rstd of x
    float* red = stat + 2 * ROWS;                                        // [ROWS][NCG][2] partial s1, s2
    long* xbase = reinterpret_cast<long*>(red + 2 * ROWS * NCG);         // [ROWS] element offset of the x / dx row, -1: none
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int hd = C / d.heads;
    const long row0 = (long)blockIdx.x * ROWS;

    // ---- gradient pieces -> Bimg.  Lanes run along TOKENS for a fixed (q|k|v, head, 4-column piece): the head-major
    //      layout keeps a window's consecutive tokens hd elements apart, so a wave-load touches ~7 lines per piece instead
    //      of one line per 2-3 lanes (the texture-address unit takes about one cycle per line: r02 TA_BUSY counters) ----
    {
        const int row = tid % ROWS, pg = tid / ROWS;
        constexpr int NPG = 256 / ROWS;                                   // piece groups (lanes sharing a token: none)
        const RowTok ti = row_token(d, row0 + row);
        const FastDiv by_hd(hd);
        const long to_dk = dk - dq, to_dv = dv - dq;
        const long row_qkv = (ti.bp * d.heads * d.Nqp + ti.slot) * (long)hd;
        constexpr int PPT = (K3 / 4) / NPG;
        bf16x4 pc[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int n0 = 4 * (pg + NPG * i);
            const int which = n0 / C, cc = n0 - which * C, head = by_hd.div(cc), j0 = cc - head * hd;
            const long base = sel(which == 0, 0L, sel(which == 1, to_dk, to_dv));
            pc[i] = ld4(dq + base + row_qkv + (long)head * d.Nqp * hd + j0);
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int n0 = 4 * (pg + NPG * i);
            bf16x4 val = keep_if(pc[i], ti.live);
            const float sc = sel(n0 < C, d.q_scale, 1.0f);                // the q part carries the attention scale
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = (bf16_t)((float)val[j] * sc);
            BI::put4(Bimg, row, n0, val);
        }
    }
    // ---- LayerNorm statistics of x: TPR adjacent lanes per token row read it in contiguous 16-byte pieces ----
    {
        const int row = tid / TPR, sub = tid % TPR;
        const RowTok ti = row_token(d, row0 + row);
        const int src = sel(ti.live, tok_src[ti.pw * d.Nqp + ti.slot], -2);
        constexpr int XPT = C / TPR / 8;                                  // 16-byte pieces of the x row per thread
        const long xoff = (ti.b * d.vol_in + max(src, 0)) * (long)C;
        bf16x8 xr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) xr[i] = ld8(x + xoff + 8 * (sub + TPR * i));
        float xs[XPT][8], sum = 0.f;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const bf16x8 raw = keep_if(xr[i], src >= 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) { xs[i][e] = (float)raw[e]; sum += xs[i][e]; }
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
            xbase[row] = src >= 0 ? xoff : -1L;
        }
    }
    __syncthreads();

    // ---- GEMM: dn[token][c] = sum_n B[token][n] * wqkv_t[c][n], this wave's column tiles, two token tiles ----
    const int cg = wave % NCG, tg = wave / NCG;
    const int R0 = tg * 32 + r, R1 = R0 + 16;                             // this lane's token rows of the two tiles
    const long xb0 = xbase[R0], xb1 = xbase[R1];
    const long tt0 = row0 + R0, tt1 = row0 + R1;
    const long T = (long)d.B * d.P * d.Nqp;
    const bool live0 = tt0 < T, live1 = tt1 < T;
    // operands of the LayerNorm backward, issued before the GEMM (they arrive while it runs)
    bf16x4 xq[NCT][2], tq[NCT][2];
    f32x4 lw[NCT];
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
        const int c0 = 16 * (cg + NCG * j) + 4 * g;
        xq[j][0] = ld4(x + max(xb0, 0L) + c0);
        xq[j][1] = ld4(x + max(xb1, 0L) + c0);
        tq[j][0] = ld4(d_t1 + sel(live0, tt0, 0L) * C + c0);
        tq[j][1] = ld4(d_t1 + sel(live1, tt1, 0L) * C + c0);
        lw[j] = *reinterpret_cast<const f32x4*>(ln_w + c0);
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
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (dn_out && (u ? live1 : live0)) st4(dn_out + (u ? tt1 : tt0) * C + c0, pack4(acc[j][u]));
            const bf16x4 raw = keep_if(xq[j][u], (u ? xb1 : xb0) >= 0);
            const float mean = u ? mean1 : mean0, rstd = u ? rstd1 : rstd0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dxh = acc[j][u][e] * lw[j][e];
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
        if (g == 0) {
            float* dst = red + ((u ? R1 : R0) * NCG + cg) * 2;
            dst[0] = a;
            dst[1] = b;
        }
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
        const long xb = u ? xb1 : xb0;
        const float rstd = u ? rstd1 : rstd0;
        if (xb < 0) continue;                                             // zero-pad / padding-slot tokens have no voxel
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int c0 = 16 * (cg + NCG * j) + 4 * g;
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = (float)tq[j][u][e] + rstd * (acc[j][u][e] - m1 - xh[j][u][e] * m2);
            st4(dx + xb + c0, pack4(out));
        }
    }
}

template <int CT>
size_t qkv_bwd_wide_lds() {
    using G = WideGeom<CT>;
    return (size_t)G::ROWS * RowImg<3 * G::C>::ROWB + (2 * G::ROWS + 2 * G::ROWS * G::NCG) * sizeof(float) + G::ROWS * sizeof(long);
}

}  // namespace

int mivp_tok_wide_supported(const MivpSwinDesc* d) {
    const int hd = d->C / d->heads;
    if (!(d->C == 96 || d->C == 192 || d->C == 384)) return 0;
    static const bool off = getenv("MIVP_NO_WIDE_TOKEN_KERNELS") != nullptr;     // A/B switch for profiling and tests
    if (off) return 0;
    return (hd % 4 == 0 && hd < 65536) ? 1 : 0;
}

int mivp_tok_wide_qkv_bwd(const MivpSwinDesc* d, const void* dq, const void* dk, const void* dv, const void* x, const int32_t* tok_src,
                          const float* ln_w, const void* wqkv_t, const void* d_t1, void* dx, void* dn_out, hipStream_t st) {
    const long T = (long)d->B * d->P * d->Nqp;
#define L_WQB(CTV)                                                                                                          \
    do {                                                                                                                    \
        const size_t lds = qkv_bwd_wide_lds<CTV>();                                                                         \
        MIVP_LDS_OPT_IN(k_qkv_bwd_wide<CTV>, lds);                                                                          \
        const unsigned grid = (unsigned)((T + WideGeom<CTV>::ROWS - 1) / WideGeom<CTV>::ROWS);                              \
        hipLaunchKernelGGL((k_qkv_bwd_wide<CTV>), dim3(grid), dim3(256), lds, st, *d, (const bf16_t*)dq, (const bf16_t*)dk, \
                           (const bf16_t*)dv, (const bf16_t*)x, tok_src, ln_w, (const bf16_t*)wqkv_t, (const bf16_t*)d_t1,  \
                           (bf16_t*)dx, (bf16_t*)dn_out);                                                                   \
    } while (0)
    switch (d->C) {
        case 96: L_WQB(6); break;
        case 192: L_WQB(12); break;
        case 384: L_WQB(24); break;
        default: mivp_set_error("tok_wide: C not in {96, 192, 384}"); return MIVP_EUNSUPPORTED;
    }
#undef L_WQB
    return mivp_check_launch("swin_qkv_bwd(wide)");
}
