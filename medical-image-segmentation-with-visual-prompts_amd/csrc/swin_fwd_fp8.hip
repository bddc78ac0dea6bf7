// EXPERIMENTAL fp8 (OCP E4M3) variant of the window-attention forward for head_dim <= 16 (BASELINE.json configs[4]:
// "fp8 MFMA window-attention"): Q', K' (head dims + bias columns), V and P are E4M3, S and O accumulate in f32
// (v_mfma_f32_16x16x32_fp8_fp8).  Same structure and semantics as k_win_attn_fwd<1,1,...> in swin_fwd.hip (one workgroup
// of 8 waves per (window, head), S^T = K' Q'^T so that one query sits on each lane, lazily rescaled online softmax in log2
// units, multiplicative shift mask = logit -> 0, softmax denominator from a ones row of V^T); reference semantics
// window_attention.py:49-61, swin_block.py:187-225.
//
// Why it exists: to SETTLE the fp8 question with data.  On gfx950 the non-scaled fp8 MFMA runs at the
// bf16 rate (MI355X_MICROARCH.md, Matrix cores), the kernel is bound by v_exp_f32 / VALU issue, and P needs as many
// conversion instructions in fp8 as in bf16 -- so the only gain is half the LDS image.  tools/fp8_attn.py measures time and
// error next to the bf16 kernel; the numbers are in profiles/r02_fp8_attention.json.  Not used unless
// mivp_amd.swin_ops.USE_FP8_ATTN_FWD is set.
#include "common.hpp"

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

MIVP_DEV unsigned pk4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
MIVP_DEV float sat448(float x) { return fminf(fmaxf(x, -448.f), 448.f); }          // E4M3 has no infinity: saturate explicitly
MIVP_DEV unsigned pk4_fp8(bf16x4 v) { return pk4_fp8(sat448((float)v[0]), sat448((float)v[1]), sat448((float)v[2]), sat448((float)v[3])); }
MIVP_DEV f32x4 mfma_fp8(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}
// K' image: 32-byte rows (32 fp8), the two 16-byte halves swapped in rows 8..15 of every 16 (conflict-free ds_read_b64)
MIVP_DEV int krow_off(int row, int byte) { return row * 32 + (byte ^ (((row >> 3) & 1) << 4)); }

}  // namespace

template <bool MASKED>
__global__ __launch_bounds__(512, 2) void k_win_attn_fwd_fp8(MivpSwinDesc d, const bf16_t* __restrict__ q,
                                                             const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                             const bf16_t* __restrict__ kp, const bf16_t* __restrict__ vp,
                                                             const bf16_t* __restrict__ qa, const bf16_t* __restrict__ ka,
                                                             const int* __restrict__ tok_rid, bf16_t* __restrict__ o,
                                                             float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8;
    const int Nkp = d.Nkp, Nqp = d.Nqp;
    const int VROW = Nkp + 16;                               // bytes per V^T row (fp8)
    char* Kimg = smem;
    char* Vt = Kimg + (size_t)Nkp * 32;
    uint8_t* ridk = reinterpret_cast<uint8_t*>(Vt + (size_t)16 * VROW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, heads = d.heads, hd = C / heads;
    const long bph = blockIdx.x;
    const int head = (int)(bph % heads);
    const long bp = bph / heads;
    const int pw = (int)(bp % d.P);
    const int A = d.augp;
    const int hd4 = hd / 4, a4 = A / 4;
    const bf16_t* kb = k + bph * (long)Nqp * hd;
    const bf16_t* vb = v + bph * (long)Nqp * hd;
    const bf16_t* kpb = d.Np > 0 ? kp + (long)head * d.Npp * hd : kb;
    const bf16_t* vpb = d.Np > 0 ? vp + (long)head * d.Npp * hd : vb;
    const bf16_t* kab = ka + (long)head * Nkp * A;
    const int n_prompt_rows = d.Np > 0 ? d.Npp : 0;
    // ---- stage K' (fp8): a thread converts one 4-element piece ----
    for (int e = tid; e < Nkp * 8; e += 64 * NW) {
        const int row = e >> 3, c4 = e & 7;
        bf16x4 val = zero4();
        if (c4 < hd4) {
            if (row < Nqp) val = ld4(kb + ((uint32_t)row * hd + 4 * c4));
            else if (row < Nqp + n_prompt_rows) val = ld4(kpb + ((uint32_t)(row - Nqp) * hd + 4 * c4));
        } else if (c4 < hd4 + a4) {
            val = ld4(kab + ((uint32_t)row * A + 4 * (c4 - hd4)));
            // the padding-key bias (-30000 log2 e) saturates E4M3 at -448: still far below every real logit
        }
        *reinterpret_cast<unsigned*>(Kimg + krow_off(row, 4 * c4)) = pk4_fp8(val);
    }
    // ---- stage V^T (fp8): four consecutive keys of one 4-channel group, transposed 4x4; row hd = 1 (softmax denominator) ----
    for (int e = tid; e < (Nkp / 4) * 4; e += 64 * NW) {
        const int c4 = e & 3, k4 = e >> 2;
        bf16x4 in[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * k4 + i;
            bf16x4 val = zero4();
            if (c4 < hd4) {
                if (row < Nqp) val = ld4(vb + ((uint32_t)row * hd + 4 * c4));
                else if (row < Nqp + n_prompt_rows) val = ld4(vpb + ((uint32_t)(row - Nqp) * hd + 4 * c4));
            }
            if (c4 == hd4) val[0] = (bf16_t)1.0f;
            in[i] = val;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<unsigned*>(Vt + (size_t)(4 * c4 + j) * VROW + 4 * k4) =
                pk4_fp8((float)in[0][j], (float)in[1][j], (float)in[2][j], (float)in[3][j]);
    }
    for (int m = tid; m < Nkp; m += 64 * NW)
        ridk[m] = (uint8_t)((m < d.Nq && MASKED) ? tok_rid[pw * Nqp + m] : (m < d.Nq ? 0 : 254));
    __syncthreads();
    bool cut = false;
    if (MASKED) {
        int differs = 0;
        for (int m = tid; m < d.Nq; m += 64 * NW) differs |= ridk[m] != ridk[0];
        cut = __syncthreads_or(differs) != 0;
    }
    const int npairs = Nkp / 32;
    constexpr float RESCALE_LOG2 = 8.f;                      // P <= 2^8 = 256 < 448 (E4M3 maximum)
    const int nqt = Nqp / 16;
    const bf16_t* qb = q + bph * (long)Nqp * hd;
    bf16_t* ob = o + bp * (long)Nqp * C + head * hd;
    for (int qt = wave; qt < nqt; qt += NW) {
        const int qrow = qt * 16 + r;
        const uint32_t rq = (MASKED && qrow < d.Nq) ? (uint32_t)tok_rid[pw * Nqp + qrow] : 0u;
        u32x2 qf;                                            // Q' row of this lane: columns 8g .. 8g+7
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int c4 = 2 * g + hlf;
            bf16x4 val = zero4();
            if (c4 < hd4) val = ld4(qb + ((uint32_t)qrow * hd + 4 * c4));
            else if (c4 < hd4 + a4) val = ld4(qa + ((uint32_t)qrow * A + 4 * (c4 - hd4)));
            qf[hlf] = pk4_fp8(val);
        }
        f32x4 oacc = fzero4(), negm = fzero4();
        float mrun = 0.f;
        bool first = true;
        for (int u = 0; u < npairs; ++u) {
            f32x4 sv[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const u32x2 kf = *reinterpret_cast<const u32x2*>(Kimg + krow_off(16 * (2 * u + hh) + r, 8 * g));
                f32x4 acc = mfma_fp8(kf, qf, negm);
                if (MASKED && cut) {
                    const uint32_t kr = *reinterpret_cast<const uint32_t*>(ridk + 16 * (2 * u + hh) + 4 * g);
                    const uint32_t k0 = kr & 0xFFu, k1 = (kr >> 8) & 0xFFu, k2 = (kr >> 16) & 0xFFu, k3 = kr >> 24;
                    acc[0] = (k0 == rq || k0 == 254u) ? acc[0] : negm[0];
                    acc[1] = (k1 == rq || k1 == 254u) ? acc[1] : negm[0];
                    acc[2] = (k2 == rq || k2 == 254u) ? acc[2] : negm[0];
                    acc[3] = (k3 == rq || k3 == 254u) ? acc[3] : negm[0];
                }
                sv[hh] = acc;
            }
            float pm = max3_raw(sv[0][0], sv[0][1], sv[0][2]);
            pm = max3_raw(pm, sv[0][3], sv[1][0]);
            pm = max3_raw(pm, sv[1][1], sv[1][2]);
            pm = max2_raw(pm, sv[1][3]);
            if (first || __any(pm > RESCALE_LOG2)) {
                asm volatile("" ::: "memory");
                pm = max2_raw(pm, __shfl_xor(pm, 16));
                pm = max2_raw(pm, __shfl_xor(pm, 32));
                const float up = first ? pm : max2_raw(pm, 0.f);
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-up);
                mrun += up;
                negm = negm - up;
                sv[0] = sv[0] - up;
                sv[1] = sv[1] - up;
                oacc = oacc * alpha;
                first = false;
            }
            const char* vrow = Vt + (size_t)r * VROW;
            u32x2 vf;
            vf[0] = *reinterpret_cast<const unsigned*>(vrow + 32 * u + 4 * g);
            vf[1] = *reinterpret_cast<const unsigned*>(vrow + 32 * u + 16 + 4 * g);
            u32x2 pb;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                pb[hh] = pk4_fp8(__builtin_amdgcn_exp2f(sv[hh][0]), __builtin_amdgcn_exp2f(sv[hh][1]),
                                 __builtin_amdgcn_exp2f(sv[hh][2]), __builtin_amdgcn_exp2f(sv[hh][3]));
            oacc = mfma_fp8(vf, pb, oacc);
        }
        // sum_k P sits in O's row hd: lane (r, g = (hd % 16) / 4), element hd % 4
        const int e1 = hd & 3, g1 = (hd & 15) >> 2;
        float pick = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) pick = (e == e1) ? oacc[e] : pick;
        const float ls = __shfl(pick, r + 16 * g1);
        const float inv = __builtin_amdgcn_rcpf(ls);
        if (4 * g < hd) st4(ob + ((uint32_t)qrow * C + 4 * g), pack4(oacc * inv));
        if (g == 0) lse[bph * Nqp + qrow] = (mrun + __builtin_amdgcn_logf(ls)) * MIVP_LN2;
    }
}

extern "C" int mivp_win_attn_fwd_fp8(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp,
                                     const void* vp, const void* qa, const void* ka, const int32_t* tok_rid, void* o,
                                     float* lse, mivp_stream_t stream) {
    MIVP_REQUIRE(d && q && k && v && qa && ka && o && lse);
    MIVP_REQUIRE(d->B > 0 && d->P > 0 && d->heads > 0 && d->C % d->heads == 0);
    const int hd = d->C / d->heads;
    MIVP_REQUIRE(hd % 4 == 0 && hd < 16 && hd + d->augp <= 32 && d->augp % 4 == 0);
    MIVP_REQUIRE(d->Nqp % 16 == 0 && d->Nkp % 32 == 0 && d->Nkp >= d->Nqp + d->Npp);
    MIVP_REQUIRE(d->Np == 0 || (kp && vp));
    MIVP_REQUIRE(!d->has_mask || tok_rid);
    MIVP_REQUIRE(!d->attn_drop_thr);
    const size_t lds = (size_t)d->Nkp * 32 + (size_t)16 * (d->Nkp + 16) + (size_t)d->Nkp;
    auto kern = d->has_mask ? k_win_attn_fwd_fp8<true> : k_win_attn_fwd_fp8<false>;
    MIVP_LDS_OPT_IN(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)d->B * d->P * d->heads)), dim3(512), lds, (hipStream_t)stream, *d,
                       (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)kp, (const bf16_t*)vp,
                       (const bf16_t*)qa, (const bf16_t*)ka, tok_rid, (bf16_t*)o, lse);
    return mivp_check_launch("win_attn_fwd_fp8");
}
