// Swin block backward: data gradients (dx) and the prompt-token / prompt-bias gradients.
// Mirrors swin_fwd.hip stage by stage; reference semantics as there (swin_block.py:145-255,
// window_attention.py:35-61).  Attention backward recomputes P from q, k and the saved
// log-sum-exp (flash-style), in two owner passes:
//   dq  pass: one wave owns 16 queries, walks all keys   (S^T = K' Q'^T : query on the lane)
//   dkv pass: one wave owns 16 keys,    walks all queries (S   = Q' K'^T : key on the lane)
// so every reduction a wave needs is over the MFMA row index of its own accumulators and the
// probabilities feed the next MFMA without leaving their lanes (common.hpp convention).
#include "common.hpp"

int mivp_attn_tile_config(const MivpSwinDesc* d, int* dks, int* nt);
// swin_tok_wide.hip: the column-split token kernels of the wide stages
int mivp_tok_wide_supported(const MivpSwinDesc* d);
int mivp_tok_rows_supported(const MivpSwinDesc* d);
long mivp_tok_natural_offset(int C);
int mivp_tok_wide_qkv_bwd(const MivpSwinDesc* d, const void* dq, const void* dk, const void* dv, const void* x, const int32_t* tok_src,
                          const float* ln_w, const void* wqkv_t, const void* d_t1, void* dx, void* dn_out, hipStream_t st);
int mivp_tok_wide_proj_mlp_bwd(const MivpSwinDesc* d, const void* dy, const int32_t* tok_dst, const void* t1, const float* ln_w,
                               const void* wmlp_t, const void* wproj_t, void* d_o, void* d_t1, hipStream_t st);

namespace {

struct TokInfo { long tt, bp, b; int slot, pw; bool live; };

MIVP_DEV TokInfo token_info(const MivpSwinDesc& d, long t) {
    TokInfo ti;
    const long T = (long)d.B * d.P * d.Nqp;
    ti.live = t < T;
    const unsigned tu = ti.live ? (unsigned)t : 0u;          // T fits 32 bits (bwd_checks): 32-bit divisions, not 64-bit ones
    const unsigned bpu = tu / (unsigned)d.Nqp;
    ti.tt = tu;
    ti.bp = bpu;
    ti.slot = (int)(tu - bpu * (unsigned)d.Nqp);
    ti.pw = (int)(bpu % (unsigned)d.P);
    ti.b = bpu / (unsigned)d.P;
    return ti;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// proj + MLP backward:  dy -> (dO, dt1)
//   t2 = t1 + Linear(LN(t1)) ; t1 = proj(O) + b + t0
//   dh  = dt2 Wmlp            (A = Wmlp^T rows = input channel)
//   dt1 = dt2 + LNbwd(dh ; t1) ; dO = dt1 Wproj
// ---------------------------------------------------------------------------------------------
// (second launch bound = waves per SIMD the register allocation must leave room for: the C = 192 form sat just above 256
// VGPRs -- one workgroup per CU, and the 352 workgroups of a 12 x 12 x 24-token stage ran as two rounds)
template <int CT>
__global__ __launch_bounds__(256, CT >= 12 ? 2 : (CT >= 6 ? 3 : 4)) void k_swin_proj_mlp_bwd(MivpSwinDesc d, const bf16_t* __restrict__ dy,
                                                           const int* __restrict__ tok_dst, const bf16_t* __restrict__ t1,
                                                           const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                           const bf16_t* __restrict__ wmlp_t, const bf16_t* __restrict__ wproj_t,
                                                           bf16_t* __restrict__ d_o, bf16_t* __restrict__ d_t1,
                                                           bf16_t* __restrict__ dn_out, bf16_t* __restrict__ dyw,
                                                           bf16_t* __restrict__ d_pj) {
    // dropout keys of this call (common.hpp drop_seed: uniform, scalar-ALU work; unused without dropout)
    const uint32_t attn_key = drop_seed(d.attn_seed, d.seed_epoch), proj_key = drop_seed(d.proj_seed, d.seed_epoch);
    (void)attn_key; (void)proj_key;
    // dn_out / dyw (optional, weight-gradient mode): gradient w.r.t. the mlp_norm output and dy in window order;
    // d_pj (optional): dt1 under the proj-dropout mask = gradient w.r.t. the proj output
    constexpr int KS = (CT + 1) / 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C;
    const TokInfo ti = token_info(d, ((long)blockIdx.x * 4 + wave) * 16 + r);
    const int dst = sel(ti.live, tok_dst[ti.pw * d.Nqp + ti.slot], -1);
    const bf16_t* dyrow = dy + (ti.b * d.vol_out + max(dst, 0)) * (long)C;

    // every load of the token tile up front, unconditional (common.hpp "Branch-free loads"): the t1 row first (it does not
    // wait for tok_dst), then dy as the GEMM's B operand and again in the output lane map (the residual term of dt1)
    bf16x4 t1raw[CT], dyraw[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) t1raw[ct] = ld4(t1 + ti.tt * (long)C + min(16 * ct + 4 * g, C - 4));
    bf16x8 dyb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * g;
        dyb[s] = keep_if(ld8(dyrow + min(c, C - 8)), dst >= 0 && c < C);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dyraw[ct] = ld4(dyrow + min(16 * ct + 4 * g, C - 4));
    f32x4 dh[CT], tv[CT];
    float sum = 0.f;
    constexpr bool LDSW = CT >= 6;                              // wide stages share the weights through LDS (common.hpp)
    using WS = WeightSlabs<KS>;
    __shared__ __attribute__((aligned(16))) char wsm[LDSW ? WS::BYTES : 16];
    WS ws;
    if (LDSW) { ws.fetch(wmlp_t, 0); ws.store(wsm, 0); __syncthreads(); }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc = fzero4();
        const int row = 16 * ct + r;
        if (LDSW) {
            if (ct + 1 < CT) ws.fetch(wmlp_t, ct + 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = mfma16(WS::frag(wsm, ct & 1, s, lane), dyb[s], acc);
            if (ct + 1 < CT) ws.store(wsm, (ct + 1) & 1);
            __syncthreads();
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = mfma16(wfrag(wmlp_t, KS, ct, s, lane), dyb[s], acc);
            }
        }
        dh[ct] = acc;
        const int n0 = 16 * ct + 4 * g;
        if (dn_out && ti.live && n0 < C) st4(dn_out + ti.tt * (long)C + n0, pack4(acc));
        f32x4 v;
        { const bf16x4 raw = keep_if(t1raw[ct], ti.live && n0 < C); for (int j = 0; j < 4; ++j) v[j] = (float)raw[j]; }
        tv[ct] = v;
        sum += v[0] + v[1] + v[2] + v[3];
    }
    const float mean = col_sum(sum) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
        if (16 * ct + 4 * g < C)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dv = tv[ct][j] - mean; var += dv * dv; }
    const float rstd = rsqrtf(col_sum(var) / (float)C + d.ln_eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int n0 = 16 * ct + 4 * g;
        if (n0 < C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dxh = dh[ct][j] * ln_w[n0 + j];
                const float xh = (tv[ct][j] - mean) * rstd;
                dh[ct][j] = dxh;
                tv[ct][j] = xh;
                s1 += dxh;
                s2 += dxh * xh;
            }
        }
    }
    const float m1 = col_sum(s1) / (float)C, m2 = col_sum(s2) / (float)C;
    bf16x4 g1[2 * KS];
#pragma unroll
    for (int ct = 0; ct < 2 * KS; ++ct) {
        bf16x4 out = zero4();
        if (ct < CT) {
            const int n0 = 16 * ct + 4 * g;
            if (n0 < C) {
                f32x4 v;
                { const bf16x4 raw = keep_if(dyraw[ct], dst >= 0); for (int j = 0; j < 4; ++j) v[j] = (float)raw[j]; }
                if (dyw && ti.live) st4(dyw + ti.tt * (long)C + n0, pack4(v));
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += rstd * (dh[ct][j] - m1 - tv[ct][j] * m2);
                out = pack4(v);
                if (ti.live) st4(d_t1 + ti.tt * (long)C + n0, out);      // the residual branch sees dt1 unmasked
                if (d.proj_drop_thr) {
                    const uint32_t pi = (uint32_t)((ti.tt * C + n0) >> 1);
                    const uint32_t h0 = drop_hash(pi, proj_key), h1 = drop_hash(pi + 1, proj_key);
                    v[0] = drop_keep(h0, 0, d.proj_drop_thr) ? v[0] * d.proj_drop_scale : 0.f;
                    v[1] = drop_keep(h0, 1, d.proj_drop_thr) ? v[1] * d.proj_drop_scale : 0.f;
                    v[2] = drop_keep(h1, 0, d.proj_drop_thr) ? v[2] * d.proj_drop_scale : 0.f;
                    v[3] = drop_keep(h1, 1, d.proj_drop_thr) ? v[3] * d.proj_drop_scale : 0.f;
                    out = pack4(v);
                    if (d_pj && ti.live) st4(d_pj + ti.tt * (long)C + n0, out);
                }
            }
        }
        g1[ct] = out;
    }
    if (LDSW) { ws.fetch(wproj_t, 0); ws.store(wsm, 0); __syncthreads(); }    // buffer 0: last read before GEMM A's final barrier
#pragma unroll
    for (int mt = 0; mt < CT; ++mt) {
        f32x4 acc = fzero4();
        const int row = 16 * mt + r;
        if (LDSW) {
            if (mt + 1 < CT) ws.fetch(wproj_t, mt + 1);
#pragma unroll
            for (int s = 0; s < KS; ++s)                          // (paired image: the fragment is already lo | hi)
                acc = mfma16(WS::frag(wsm, mt & 1, s, lane), cat44(g1[2 * s], g1[2 * s + 1]), acc);
            if (mt + 1 < CT) ws.store(wsm, (mt + 1) & 1);
            __syncthreads();
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = mfma16(wfrag(wproj_t, KS, mt, s, lane), cat44(g1[2 * s], g1[2 * s + 1]), acc);
            }
        }
        const int n0 = 16 * mt + 4 * g;
        if (ti.live && n0 < C) st4(d_o + ti.tt * (long)C + n0, pack4(acc));
    }
}

// ---------------------------------------------------------------------------------------------
// delta[bp][head][q] = sum_j dO * O over the head's channels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_win_attn_delta(MivpSwinDesc d, const bf16_t* __restrict__ o,
                                                        const bf16_t* __restrict__ d_o, float* __restrict__ delta) {
    // one thread per (token, head), head fastest: a wave reads whole contiguous token rows of o and dO
    const int hd = d.C / d.heads;
    const long tokens = (long)d.B * d.P * d.Nqp, total = tokens * d.heads;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned tok = (unsigned)i / (unsigned)d.heads;        // host checks total < 2^31: 32-bit decode
        const int head = (int)((unsigned)i - tok * (unsigned)d.heads);
        const long bp = tok / (unsigned)d.Nqp;
        const int qrow = (int)(tok - (unsigned)bp * (unsigned)d.Nqp);
        const long off = tok * (long)d.C + head * hd;
        float acc = 0.f;
        for (int j = 0; j < hd; j += 4) {
            const bf16x4 a = ld4(o + off + j), b = ld4(d_o + off + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += (float)a[e] * (float)b[e];
        }
        delta[(bp * d.heads + head) * d.Nqp + qrow] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// dq pass.  grid = B*P*heads; one workgroup stages the keys of its (window, head) ONCE per chunk and
// every wave walks its own query tiles (wave + 4i) against that chunk.  LDS: K' rows (S), V rows (dP),
// K^T (dq), key classes.  For head_dim <= 16 (DVT == 1) the dP product uses the K=16 MFMA so the V rows
// are only 16 wide.  The pass also produces delta[q] = sum_j dO*O (needed again by the dkv pass).
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
MIVP_DEV f32x4 mfma16k16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

template <int DKS, int DVT>
struct AttnBwdGeom {
    static constexpr int DK = 32 * DKS;
    static constexpr bool V16 = (DVT == 1);                 // head_dim <= 16: 16-wide value rows + K=16 MFMA
    static constexpr int DVS = (DVT + 1) / 2;
    static constexpr int DVP = V16 ? 16 : 32 * DVS;
    using KR = OperandRows<DK>;
    static constexpr int KROW = KR::ROW;                     // bytes per K'/Q' row
    static constexpr int VROWB = (DVP + 8) * 2;             // bytes per V/dO row
};

// (DKS = 2, head_dim 24 / 32: capped at 128 VGPRs -- 5 to 12 spilled dwords outside the key loop -- for the second workgroup per
//  CU the 78 KB LDS budget below is sized for: 96 -> 81 us un-shifted, 122 -> 102 us shifted at 24^3 x 4, round 3)
template <int DKS, int DVT, int QPW, int NW, bool DROP, bool MASKED>
__global__ __launch_bounds__(64 * NW, (DKS <= 2 && !DROP) ? 4 : 2) void k_win_attn_bwd_dq(MivpSwinDesc d, int chunk_tiles, const bf16_t* __restrict__ q,
                                                         const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                         const bf16_t* __restrict__ kp, const bf16_t* __restrict__ vp,
                                                         const bf16_t* __restrict__ qa, const bf16_t* __restrict__ ka,
                                                         const int* __restrict__ tok_rid, const bf16_t* __restrict__ o,
                                                         const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                         float* __restrict__ delta, bf16_t* __restrict__ dq) {
    // dropout keys of this call (common.hpp drop_seed: uniform, scalar-ALU work; unused without dropout)
    const uint32_t attn_key = drop_seed(d.attn_seed, d.seed_epoch), proj_key = drop_seed(d.proj_seed, d.seed_epoch);
    (void)attn_key; (void)proj_key;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = AttnBwdGeom<DKS, DVT>;
    constexpr int DK = G::DK, DVS = G::DVS, DVP = G::DVP, KROW = G::KROW, VROWB = G::VROWB;
    const int ckeys = chunk_tiles * 16;
    const int KTROW = (ckeys + 8) * 2;
    char* Kimg = smem;
    char* Vimg = Kimg + (size_t)ckeys * KROW;
    char* Kt = Vimg + (size_t)ckeys * VROWB;
    int* ridk = reinterpret_cast<int*>(Kt + (size_t)(16 * DVT) * KTROW);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, heads = d.heads, hd = C / heads, Nqp = d.Nqp, Nkp = d.Nkp, A = d.augp;
    const int hd4 = hd / 4, dk4 = DK / 4, a4 = A / 4, dvp4 = DVP / 4;
    const long bph = blockIdx.x;
    const int head = (int)(bph % heads);
    const long bp = bph / heads;
    const int pw = (int)(bp % d.P);
    const int nqt = Nqp / 16;

    f32x4 dqacc[QPW][DVT];
    float lse_b[QPW], dl[QPW];
    int rq[QPW];
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
#pragma unroll
        for (int dd = 0; dd < DVT; ++dd) dqacc[i][dd] = fzero4();
        const int qt = wave + NW * i;
        const int qrow = qt < nqt ? qt * 16 + r : 0;
        rq[i] = (MASKED && qrow < d.Nq) ? tok_rid[pw * Nqp + qrow] : 0;
        lse_b[i] = -lse[bph * Nqp + qrow] * MIVP_LOG2E;       // S accumulators start here: exp2(S) is P (common.hpp)
        // delta = sum_j dO * O over this head's channels: each lane covers 4g.. of every 16, then the 4 g-lanes add up
        float acc = 0.f;
        for (int c4 = g; c4 < hd4; c4 += 4) {
            const bf16x4 ov = ld4(o + ((bp * Nqp + qrow) * (long)C + head * hd + 4 * c4));
            const bf16x4 gv = ld4(d_o + ((bp * Nqp + qrow) * (long)C + head * hd + 4 * c4));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += (float)ov[e] * (float)gv[e];
        }
        acc = col_sum(acc);
        dl[i] = acc;
        if (qt < nqt && g == 0) delta[bph * Nqp + qrow] = acc;
    }

    const int nt = Nkp / 16;
    const int nt_full = d.Nq / 16;
    const bf16_t* kb = k + bph * (long)Nqp * hd;             // this (window, head)'s rows; uniform
    const bf16_t* vb = v + bph * (long)Nqp * hd;
    const bf16_t* qb = q + bph * (long)Nqp * hd;
    const bf16_t* kpb = d.Np > 0 ? kp + (long)head * d.Npp * hd : kb;
    const bf16_t* vpb = d.Np > 0 ? vp + (long)head * d.Npp * hd : vb;
    const bf16_t* kab = ka + (long)head * Nkp * A;
    const bf16_t* dob = d_o + bp * (long)Nqp * C + head * hd;
    const int n_prompt_rows = d.Np > 0 ? d.Npp : 0;
    // This wave's Q' / dO fragments do not depend on the key chunk: loaded ONCE, unconditionally and back to back (they
    // used to be re-read for every chunk by one conditional load -- one memory round trip -- per 8-byte piece: 8 dependent
    // round trips in front of every query tile's key loop).  common.hpp "Branch-free loads".
    bf16x8 qf_all[QPW][DKS];
    bf16x8 dof_all[QPW][DVS];
    bf16x4 dof4_all[QPW];
    {
        const long to_qa = qa - qb;
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int qt = wave + NW * i;
            const int qrow = (qt < nqt ? qt : 0) * 16 + r;
#pragma unroll
            for (int s = 0; s < DKS; ++s) {
                bf16x4 piece[2];
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const int c4 = 8 * s + 2 * g + hlf;
                    const int ca = min(max(c4 - hd4, 0), a4 - 1);
                    const long off = sel(c4 < hd4, (long)(qrow * hd + 4 * min(c4, hd4 - 1)), to_qa + (long)(qrow * A + 4 * ca));
                    piece[hlf] = keep_if(ld4(qb + off), c4 < hd4 + a4);
                }
                qf_all[i][s] = cat44(piece[0], piece[1]);
            }
            dof4_all[i] = zero4();
            if (G::V16) {
                dof4_all[i] = keep_if(ld4(dob + ((uint32_t)qrow * C + 4 * min(g, hd4 - 1))), g < hd4);
            } else {
#pragma unroll
                for (int s = 0; s < DVS; ++s) {
                    bf16x4 piece[2];
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const int c4 = 8 * s + 2 * g + hlf;
                        piece[hlf] = keep_if(ld4(dob + ((uint32_t)qrow * C + 4 * min(c4, hd4 - 1))), c4 < hd4);
                    }
                    dof_all[i][s] = cat44(piece[0], piece[1]);
                }
            }
        }
    }
    // a shifted block's window that the volume boundary does not cut has ONE region id: its mask is a no-op
    bool cut = false;
    if (MASKED) {
        const int first = tok_rid[pw * Nqp];
        int differs = 0;
        for (int m = tid; m < d.Nq; m += 64 * NW) differs |= tok_rid[pw * Nqp + m] != first;
        cut = __syncthreads_or(differs) != 0;
    }
    for (int t0 = 0; t0 < nt; t0 += chunk_tiles) {
        const int ntc = (nt - t0) < chunk_tiles ? (nt - t0) : chunk_tiles;       // tiles in this chunk (even)
        const int key0 = t0 * 16, nkeys = ntc * 16;
        __syncthreads();
        {   // K' rows (+ K^T of the head dims): a thread keeps one 8-byte column of the image; four unconditional loads in
            // flight per thread before the first LDS write (clamped rows: the ragged tail rewrites the last row)
            constexpr int RPP = 64 * NW / dk4;
            const int c4 = tid % dk4;
            const bool from_k = c4 < hd4, from_a = !from_k && c4 < hd4 + a4;
            const int cc = min(c4, hd4 - 1), ca = min(max(c4 - hd4, 0), a4 - 1);
            const long to_kp = kpb - kb, to_ka = kab - kb;
            const int max_prow = n_prompt_rows > 0 ? n_prompt_rows - 1 : 0;
            for (int rowb = tid / dk4; rowb < nkeys; rowb += 4 * RPP) {
                bf16x4 vals[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = key0 + min(rowb + i * RPP, nkeys - 1);
                    const int pr = min(max(row - Nqp, 0), max_prow);
                    const long off_k = sel(row < Nqp, (long)(row * hd + 4 * cc), to_kp + (long)(pr * hd + 4 * cc));
                    const long off = sel(from_k, off_k, to_ka + (long)(row * A + 4 * ca));
                    vals[i] = keep_if(ld4(kb + off), sel(from_k, (int)(row < Nqp + n_prompt_rows), (int)from_a) != 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int lrow = min(rowb + i * RPP, nkeys - 1);
                    if (from_k) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) *reinterpret_cast<bf16_t*>(Kt + (size_t)(4 * c4 + e) * KTROW + 2 * lrow) = vals[i][e];
                    }
                    *reinterpret_cast<bf16x4*>(Kimg + G::KR::off(lrow, 4 * c4)) = vals[i];
                }
            }
        }
        // K^T rows between hd and 16*DVT must be zero (they multiply dS in the dq MFMA)
        for (int e = tid; e < (16 * DVT - hd) * nkeys; e += 64 * NW) {
            const int rr = hd + e / nkeys, col = e % nkeys;
            *reinterpret_cast<bf16_t*>(Kt + (size_t)rr * KTROW + 2 * col) = (bf16_t)0.0f;
        }
        {
            constexpr int RPP = 64 * NW / dvp4;
            const int c4 = tid % dvp4, cc = min(c4, hd4 - 1);
            const long to_vp = vpb - vb;
            const int max_prow = n_prompt_rows > 0 ? n_prompt_rows - 1 : 0;
            for (int rowb = tid / dvp4; rowb < nkeys; rowb += 4 * RPP) {
                bf16x4 vals[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = key0 + min(rowb + i * RPP, nkeys - 1);
                    const int pr = min(max(row - Nqp, 0), max_prow);
                    const long off = sel(row < Nqp, (long)(row * hd + 4 * cc), to_vp + (long)(pr * hd + 4 * cc));
                    vals[i] = keep_if(ld4(vb + off), c4 < hd4 && row < Nqp + n_prompt_rows);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<bf16x4*>(Vimg + (size_t)min(rowb + i * RPP, nkeys - 1) * VROWB + 8 * c4) = vals[i];
            }
        }
        for (int m = tid; m < nkeys; m += 64 * NW) {
            const int row = key0 + m;
            // content key: region id; prompt and padding keys: -2 = never masked (padding keys are excluded by their bias)
            ridk[m] = row < d.Nq ? (MASKED ? tok_rid[pw * Nqp + row] : 0) : -2;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int qt = wave + NW * i;
            if (qt >= nqt) continue;
            const int qrow = qt * 16 + r;
            const bf16x8 (&qf)[DKS] = qf_all[i];
            const bf16x8 (&dof)[DVS] = dof_all[i];
            const bf16x4 dof4 = dof4_all[i];
            const int rqi = rq[i];
            const float lsei = lse_b[i], dli = dl[i];
            const uint32_t drow = DROP ? attn_row(bph, qrow, Nqp, Nkp) : 0u;
            for (int u = 0; u < ntc / 2; ++u) {
                f32x4 ds[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int lt = 2 * u + hh;
                    f32x4 s = {lsei, lsei, lsei, lsei}, dp = fzero4();
#pragma unroll
                    for (int ks = 0; ks < DKS; ++ks)
                        s = mfma16(*reinterpret_cast<const bf16x8*>(Kimg + G::KR::off(16 * lt + r, 32 * ks + 8 * g)), qf[ks], s);
                    if (G::V16) {
                        dp = mfma16k16(*reinterpret_cast<const bf16x4*>(Vimg + (size_t)(16 * lt + r) * VROWB + 8 * g), dof4, dp);
                    } else {
#pragma unroll
                        for (int ks = 0; ks < DVS; ++ks)
                            dp = mfma16(*reinterpret_cast<const bf16x8*>(Vimg + (size_t)(16 * lt + r) * VROWB + (32 * ks + 8 * g) * 2), dof[ks], dp);
                    }
                    if (DROP) {                              // dP = dropout'(dO V^T): same mask and scale as the forward
                        const uint32_t pi = attn_pair(drow, 16 * (t0 + lt) + 4 * g);
                        const uint32_t h0 = drop_hash(pi, attn_key), h1 = drop_hash(pi + 1, attn_key);
                        dp[0] = drop_keep(h0, 0, d.attn_drop_thr) ? dp[0] * d.attn_drop_scale : 0.f;
                        dp[1] = drop_keep(h0, 1, d.attn_drop_thr) ? dp[1] * d.attn_drop_scale : 0.f;
                        dp[2] = drop_keep(h1, 0, d.attn_drop_thr) ? dp[2] * d.attn_drop_scale : 0.f;
                        dp[3] = drop_keep(h1, 1, d.attn_drop_thr) ? dp[3] * d.attn_drop_scale : 0.f;
                    }
                    const int key_lo = 16 * (t0 + lt);
                    if (!(MASKED && cut) || key_lo >= d.Nq) {
                        // no shift mask in effect (or a tile of prompt / padding keys only, which is never masked); padding
                        // keys are already at P = 0 through their bias (common.hpp)
#pragma unroll
                        for (int j = 0; j < 4; ++j) ds[hh][j] = __builtin_amdgcn_exp2f(s[j]) * (dp[j] - dli);
                    } else {
                        // a masked logit is the constant 0: it keeps its share of the softmax but carries no gradient
                        const int4 kr4 = *reinterpret_cast<const int4*>(ridk + 16 * lt + 4 * g);
                        const int krs[4] = {kr4.x, kr4.y, kr4.z, kr4.w};
                        if (key_lo + 16 <= d.Nq) {                       // content keys only: one compare per element
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float val = __builtin_amdgcn_exp2f(s[j]) * (dp[j] - dli);
                                ds[hh][j] = krs[j] == rqi ? val : 0.f;
                            }
                        } else {                                         // the one tile that mixes content and padding keys
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bool live = (krs[j] == rqi) | (krs[j] == -2);
                                const float val = __builtin_amdgcn_exp2f(s[j]) * (dp[j] - dli);
                                ds[hh][j] = live ? val : 0.f;
                            }
                        }
                    }
                }
                const bf16x8 pb = cat44(pack4(ds[0]), pack4(ds[1]));
#pragma unroll
                for (int dd = 0; dd < DVT; ++dd) {
                    const char* krow = Kt + (size_t)(16 * dd + r) * KTROW;
                    const bf16x8 a = cat44(*reinterpret_cast<const bf16x4*>(krow + (32 * u + 4 * g) * 2),
                                           *reinterpret_cast<const bf16x4*>(krow + (32 * u + 16 + 4 * g) * 2));
                    dqacc[i][dd] = mfma16(a, pb, dqacc[i][dd]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
        const int qt = wave + NW * i;
        if (qt >= nqt) continue;
        const int qrow = qt * 16 + r;
#pragma unroll
        for (int dd = 0; dd < DVT; ++dd) {
            const int j0 = 16 * dd + 4 * g;
            if (j0 < hd) st4(dq + ((bph * Nqp + qrow) * (long)hd + j0), pack4(dqacc[i][dd] * MIVP_LN2));   // K carries log2(e)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dkv pass.  grid = (B*P*heads, ksplit); the workgroup stages the queries of its (window, head) once per
// chunk and every wave walks its own key tiles against that chunk.  LDS: Q' rows (S), dO rows (dP),
// Q^T (dk), dO^T (dv), log-sum-exp, delta, query region ids.
// Window keys -> dk, dv (bf16); prompt keys -> per-window f32 partials + column sums of dS (the
// gradient of the prompt-token bias score).
// ---------------------------------------------------------------------------------------------
template <int DKS, int DVT, int KPW, int NW, bool AUG, bool DROP, bool MASKED>
__global__ __launch_bounds__(64 * NW, (DKS == 1 && !DROP && !(AUG && MASKED)) ? 4 : 2) void k_win_attn_bwd_dkv(MivpSwinDesc d, int chunk_tiles, int kt0,
                                                          const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                          const bf16_t* __restrict__ v, const bf16_t* __restrict__ kp,
                                                          const bf16_t* __restrict__ vp, const bf16_t* __restrict__ qa,
                                                          const bf16_t* __restrict__ ka, const int* __restrict__ tok_rid,
                                                          const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, bf16_t* __restrict__ dk,
                                                          bf16_t* __restrict__ dv, float* __restrict__ dkp_part,
                                                          float* __restrict__ dvp_part, float* __restrict__ dtok_part,
                                                          float* __restrict__ dka_part) {
    // dropout keys of this call (common.hpp drop_seed: uniform, scalar-ALU work; unused without dropout)
    const uint32_t attn_key = drop_seed(d.attn_seed, d.seed_epoch), proj_key = drop_seed(d.proj_seed, d.seed_epoch);
    (void)attn_key; (void)proj_key;
    // AUG (weight-gradient mode): also accumulates dK' over the 32 bias-augmentation columns, i.e.
    // dka_part[bph][key][a] = sum_n dS[n, key] * qa[n][a]  (f32 [B*P*heads][Nkp][32]): the window-local gradient of
    // the relative-position-bias table values that mivp_relbias_aug laid out on the key side.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = AttnBwdGeom<DKS, DVT>;
    constexpr int DK = G::DK, DVS = G::DVS, DVP = G::DVP, QROW = G::KROW, OROW = G::VROWB;
    constexpr int QTR = 16 * DVT + (AUG ? 32 : 0);            // rows of the transposed Q' image
    const int cq = chunk_tiles * 16;
    const int TROW = (cq + 8) * 2;
    char* Qimg = smem;
    char* Oimg = Qimg + (size_t)cq * QROW;
    char* Qt = Oimg + (size_t)cq * OROW;
    char* Ot = Qt + (size_t)QTR * TROW;
    float* lse_s = reinterpret_cast<float*>(Ot + (size_t)(16 * DVT) * TROW);
    float* del_s = lse_s + cq;
    int* ridq = reinterpret_cast<int*>(del_s + cq);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, heads = d.heads, hd = C / heads, Nqp = d.Nqp, Nkp = d.Nkp, A = d.augp;
    const int hd4 = hd / 4, dk4 = DK / 4, a4 = A / 4, dvp4 = DVP / 4;
    const long bph = blockIdx.x;
    const int head = (int)(bph % heads);
    const long bp = bph / heads;
    const int pw = (int)(bp % d.P);
    const int nt = Nkp / 16;
    const int kstride = NW * gridDim.y;
    const int kfirst = kt0 + NW * blockIdx.y + wave;

    f32x4 dkacc[KPW][DVT], dvacc[KPW][DVT], dkaug[KPW][2];
    float dtok[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        dtok[i] = 0.f;
        dkaug[i][0] = fzero4();
        dkaug[i][1] = fzero4();
#pragma unroll
        for (int dd = 0; dd < DVT; ++dd) { dkacc[i][dd] = fzero4(); dvacc[i][dd] = fzero4(); }
    }

    const int nqt = (Nqp + 31) / 32 * 2;                      // query tiles rounded to pairs
    // uniform per-(window, head) bases + 32-bit offsets: 64-bit per-lane addresses cost VALU ops and registers (they spilled)
    const bf16_t* qb = q + bph * (long)Nqp * hd;
    const bf16_t* kb = k + bph * (long)Nqp * hd;
    const bf16_t* vb = v + bph * (long)Nqp * hd;
    const bf16_t* kpb = d.Np > 0 ? kp + (long)head * d.Npp * hd : kb;
    const bf16_t* vpb = d.Np > 0 ? vp + (long)head * d.Npp * hd : vb;
    const bf16_t* kab = ka + (long)head * Nkp * A;
    const bf16_t* dob = d_o + bp * (long)Nqp * C + head * hd;
    const float* lseb = lse + bph * (long)Nqp;
    const float* delb = delta + bph * (long)Nqp;
    const int n_prompt_rows = d.Np > 0 ? d.Npp : 0;
    // This wave's K' / V fragments and key classes do not depend on the query chunk: loaded once, unconditionally and back to
    // back (see the dq kernel).
    bf16x8 kf_all[KPW][DKS];
    bf16x8 vf_all[KPW][DVS];
    bf16x4 vf4_all[KPW];
    int kcls_all[KPW];
    {
        const long to_kp = kpb - kb, to_ka = kab - kb, to_vp = vpb - vb;
        const int max_prow = n_prompt_rows > 0 ? n_prompt_rows - 1 : 0;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int kt = kfirst + kstride * i;
            const int krow = (kt < nt ? kt : 0) * 16 + r;
            const bool staged = krow < Nqp + n_prompt_rows;
            const int pr = min(max(krow - Nqp, 0), max_prow);
            kcls_all[i] = sel(krow < d.Nq, d.has_mask ? tok_rid[pw * Nqp + min(krow, d.Nq - 1)] : 0, -2);
#pragma unroll
            for (int s = 0; s < DKS; ++s) {
                bf16x4 piece[2];
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const int c4 = 8 * s + 2 * g + hlf, cc = min(c4, hd4 - 1);
                    const int ca = min(max(c4 - hd4, 0), a4 - 1);
                    const long off_k = sel(krow < Nqp, (long)(krow * hd + 4 * cc), to_kp + (long)(pr * hd + 4 * cc));
                    const long off = sel(c4 < hd4, off_k, to_ka + (long)(krow * A + 4 * ca));
                    piece[hlf] = keep_if(ld4(kb + off), sel(c4 < hd4, (int)staged, (int)(c4 < hd4 + a4)) != 0);
                }
                kf_all[i][s] = cat44(piece[0], piece[1]);
            }
            auto vload = [&](int c4) -> bf16x4 {
                const int cc = min(c4, hd4 - 1);
                const long off = sel(krow < Nqp, (long)(krow * hd + 4 * cc), to_vp + (long)(pr * hd + 4 * cc));
                return keep_if(ld4(vb + off), c4 < hd4 && staged);
            };
            vf4_all[i] = zero4();
            if (G::V16) vf4_all[i] = vload(g);
            else {
#pragma unroll
                for (int s = 0; s < DVS; ++s) vf_all[i][s] = cat44(vload(8 * s + 2 * g), vload(8 * s + 2 * g + 1));
            }
        }
    }
    // a shifted block's window that the volume boundary does not cut has ONE region id: its mask is a no-op and the
    // workgroup takes the un-shifted block's lean path
    bool cut = false;
    if (MASKED) {
        const int first = tok_rid[pw * Nqp];
        int differs = 0;
        for (int m = tid; m < d.Nq; m += 64 * NW) differs |= tok_rid[pw * Nqp + m] != first;
        cut = __syncthreads_or(differs) != 0;
    }
    for (int t0 = 0; t0 < nqt; t0 += chunk_tiles) {
        const int ntc = (nqt - t0) < chunk_tiles ? (nqt - t0) : chunk_tiles;
        const int q0 = t0 * 16, nq = ntc * 16;
        __syncthreads();
        {   // Q' rows (+ Q^T): four unconditional loads in flight per thread before the first LDS write
            const long to_qa = qa - qb;
            const int n_e = nq * dk4;
            for (int e0 = tid; e0 < n_e; e0 += 4 * 64 * NW) {
                bf16x4 vals[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = min(e0 + i * 64 * NW, n_e - 1);
                    const int lrow = e / dk4, c4 = e - lrow * dk4, row = min(q0 + lrow, Nqp - 1);
                    const int ca = min(max(c4 - hd4, 0), a4 - 1);
                    const long off = sel(c4 < hd4, (long)(row * hd + 4 * min(c4, hd4 - 1)), to_qa + (long)(row * A + 4 * ca));
                    vals[i] = keep_if(ld4(qb + off), q0 + lrow < Nqp && c4 < hd4 + a4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = min(e0 + i * 64 * NW, n_e - 1);
                    const int lrow = e / dk4, c4 = e - lrow * dk4;
                    const bf16x4 val = vals[i];
                    if (c4 < 4 * DVT) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<bf16_t*>(Qt + (size_t)(4 * c4 + j) * TROW + 2 * lrow) = (c4 < hd4) ? val[j] : (bf16_t)0.0f;
                    }
                    if (AUG && c4 >= hd4 && c4 < hd4 + a4) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<bf16_t*>(Qt + (size_t)(16 * DVT + 4 * (c4 - hd4) + j) * TROW + 2 * lrow) = val[j];
                    }
                    *reinterpret_cast<bf16x4*>(Qimg + G::KR::off(lrow, 4 * c4)) = val;
                }
            }
        }
        if (AUG) {                                             // augmentation rows A .. 31 of the transposed image are zero
            for (int e = tid; e < (32 - A) * nq; e += 64 * NW) {
                const int rr = 16 * DVT + A + e / nq, col = e % nq;
                *reinterpret_cast<bf16_t*>(Qt + (size_t)rr * TROW + 2 * col) = (bf16_t)0.0f;
            }
        }
        {
            const int n_e = nq * dvp4;
            for (int e0 = tid; e0 < n_e; e0 += 4 * 64 * NW) {
                bf16x4 vals[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = min(e0 + i * 64 * NW, n_e - 1);
                    const int lrow = e / dvp4, c4 = e - lrow * dvp4, row = min(q0 + lrow, Nqp - 1);
                    vals[i] = keep_if(ld4(dob + ((uint32_t)row * C + 4 * min(c4, hd4 - 1))), q0 + lrow < Nqp && c4 < hd4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = min(e0 + i * 64 * NW, n_e - 1);
                    const int lrow = e / dvp4, c4 = e - lrow * dvp4;
                    if (c4 < 4 * DVT) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16_t*>(Ot + (size_t)(4 * c4 + j) * TROW + 2 * lrow) = vals[i][j];
                    }
                    *reinterpret_cast<bf16x4*>(Oimg + (size_t)lrow * OROW + 8 * c4) = vals[i];
                }
            }
        }
        for (int m = tid; m < nq; m += 64 * NW) {
            const int row = q0 + m;
            const bool ok = row < Nqp;
            lse_s[m] = ok ? -lseb[row] * MIVP_LOG2E : -INFINITY;   // the S accumulators start from it; padding query rows: P = 0
            del_s[m] = ok ? delb[row] : 0.f;
            ridq[m] = (ok && row < d.Nq) ? (d.has_mask ? tok_rid[pw * Nqp + row] : 0) : -1;   // -1: padding query row
        }
        __syncthreads();
        auto compute = [&](auto cut_c) {
            constexpr bool CUT = decltype(cut_c)::value;    // the shift mask is in effect for this window
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                const int kt = kfirst + kstride * i;
                if (kt >= nt) continue;
                const int krow = kt * 16 + r;
                const int kcls = kcls_all[i];                 // content key: region id; prompt / padding keys: -2 = never masked
                const uint32_t dbase = DROP ? attn_row(bph, 0, Nqp, Nkp) : 0u;
                const bf16x8 (&kf)[DKS] = kf_all[i];
                const bf16x8 (&vf)[DVS] = vf_all[i];
                const bf16x4 vf4 = vf4_all[i];
                // A key tile made of prompt / padding keys only is never masked: it takes the lean path even in a cut window.
                // Elsewhere "live" = (key class == query region) | (prompt or padding key) is ONE compare on pre-or-ed
                // operands -- (rq | pm) == kk with pm = all ones and kk = all ones for prompt / padding lanes -- so that no
                // scalar mask arithmetic (and none of its VALU->SALU->VALU hazard stalls) sits in the per-element stream.
                const uint32_t pm = kcls == -2 ? 0xFFFFFFFFu : 0u, kk = kcls == -2 ? 0xFFFFFFFFu : (uint32_t)kcls;
                auto uloop = [&](auto masked_c) {
                constexpr bool MSK = decltype(masked_c)::value;
                for (int u = 0; u < ntc / 2; ++u) {
                    f32x4 pv[2], ds[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int lt = 2 * u + hh;
                        // rows of this accumulator tile are queries 16*lt + 4g + j
                        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 16 * lt + 4 * g);
                        f32x4 s = {l4.x, l4.y, l4.z, l4.w}, dp = fzero4();
#pragma unroll
                        for (int ks = 0; ks < DKS; ++ks)
                            s = mfma16(*reinterpret_cast<const bf16x8*>(Qimg + G::KR::off(16 * lt + r, 32 * ks + 8 * g)), kf[ks], s);
                        if (G::V16) {
                            dp = mfma16k16(*reinterpret_cast<const bf16x4*>(Oimg + (size_t)(16 * lt + r) * OROW + 8 * g), vf4, dp);
                        } else {
#pragma unroll
                            for (int ks = 0; ks < DVS; ++ks)
                                dp = mfma16(*reinterpret_cast<const bf16x8*>(Oimg + (size_t)(16 * lt + r) * OROW + (32 * ks + 8 * g) * 2), vf[ks], dp);
                        }
                        const float4 d4 = *reinterpret_cast<const float4*>(del_s + 16 * lt + 4 * g);
                        const int4 r4 = *reinterpret_cast<const int4*>(ridq + 16 * lt + 4 * g);
                        const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, dls[4] = {d4.x, d4.y, d4.z, d4.w};
                        const int rqs[4] = {r4.x, r4.y, r4.z, r4.w};
                        // dropout hashes of elements (query q0 + 16lt + 4g + j, key krow): adjacent lanes hold the two keys of a hash
                        // pair, so the even lane hashes j = 0, 1, the odd lane j = 2, 3 and quad-permute moves hand them around
                        // (swin_bwd_fused.hip): 2 hashes + 4 moves per tile instead of 4 hashes
                        uint32_t hj[4] = {0u, 0u, 0u, 0u};
                        if (DROP) {
                            const uint32_t row0 = dbase + (uint32_t)(q0 + 16 * lt + 4 * g + 2 * (r & 1)) * (uint32_t)(Nkp >> 1);
                            const uint32_t h0 = drop_hash(attn_pair(row0, krow), attn_key);
                            const uint32_t h1 = drop_hash(attn_pair(row0 + (uint32_t)(Nkp >> 1), krow), attn_key);
                            hj[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0xA0, 0xF, 0xF, false);
                            hj[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0xA0, 0xF, 0xF, false);
                            hj[2] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0xF5, 0xF, 0xF, false);
                            hj[3] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0xF5, 0xF, 0xF, false);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float keep = 1.f;
                            if (DROP) keep = drop_keep(hj[j], krow & 1, d.attn_drop_thr) ? d.attn_drop_scale : 0.f;
                            float p, dsv;
                            if (MSK) {
                                // a masked logit is the constant 0 (accumulator value ls[j]): it keeps its P, carries no gradient
                                const bool live = ((uint32_t)rqs[j] | pm) == kk;
                                p = __builtin_amdgcn_exp2f(live ? s[j] : ls[j]);
                                dsv = live ? p * (dp[j] * keep - dls[j]) : 0.f;
                            } else {
                                // no shift mask in effect: P = 0 wherever the key or the query row is padding (bias / -inf start)
                                p = __builtin_amdgcn_exp2f(s[j]);
                                dsv = p * (dp[j] * keep - dls[j]);
                            }
                            pv[hh][j] = p * keep;
                            ds[hh][j] = dsv;
                            dtok[i] += dsv;
                        }
                    }
                    const bf16x8 pb = cat44(pack4(pv[0]), pack4(pv[1]));
                    const bf16x8 sb = cat44(pack4(ds[0]), pack4(ds[1]));
#pragma unroll
                    for (int dd = 0; dd < DVT; ++dd) {
                        const char* qrow_t = Qt + (size_t)(16 * dd + r) * TROW;
                        const char* orow_t = Ot + (size_t)(16 * dd + r) * TROW;
                        const bf16x8 aq = cat44(*reinterpret_cast<const bf16x4*>(qrow_t + (32 * u + 4 * g) * 2),
                                                *reinterpret_cast<const bf16x4*>(qrow_t + (32 * u + 16 + 4 * g) * 2));
                        const bf16x8 ao = cat44(*reinterpret_cast<const bf16x4*>(orow_t + (32 * u + 4 * g) * 2),
                                                *reinterpret_cast<const bf16x4*>(orow_t + (32 * u + 16 + 4 * g) * 2));
                        dkacc[i][dd] = mfma16(aq, sb, dkacc[i][dd]);
                        dvacc[i][dd] = mfma16(ao, pb, dvacc[i][dd]);
                    }
                    if (AUG) {
                        // The table gradients are signed sums of dS with heavy cancellation (each softmax row of dS sums
                        // to zero), so dS enters this product as a bf16 hi + lo pair (qa is an exact one-hot).
                        f32x4 lo[2];
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                            for (int j = 0; j < 4; ++j) lo[hh][j] = ds[hh][j] - (float)(bf16_t)ds[hh][j];
                        const bf16x8 sb_lo = cat44(pack4(lo[0]), pack4(lo[1]));
#pragma unroll
                        for (int dd = 0; dd < 2; ++dd) {
                            const char* arow_t = Qt + (size_t)(16 * DVT + 16 * dd + r) * TROW;
                            const bf16x8 aa = cat44(*reinterpret_cast<const bf16x4*>(arow_t + (32 * u + 4 * g) * 2),
                                                    *reinterpret_cast<const bf16x4*>(arow_t + (32 * u + 16 + 4 * g) * 2));
                            dkaug[i][dd] = mfma16(aa, sb, dkaug[i][dd]);
                            dkaug[i][dd] = mfma16(aa, sb_lo, dkaug[i][dd]);
                        }
                    }
                }
                };
                if (CUT && kt * 16 < d.Nq) uloop(std::true_type{});
                else uloop(std::false_type{});
            }
        };
        if (MASKED && cut) compute(std::true_type{});
        else compute(std::false_type{});
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kt = kfirst + kstride * i;
        if (kt >= nt) continue;
        const int krow = kt * 16 + r;
        const float dt = col_sum(dtok[i]);
        if (AUG) {
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
                *reinterpret_cast<f32x4*>(dka_part + ((bph * Nkp + krow) * 32 + 16 * dd + 4 * g)) = dkaug[i][dd];
        }
        if (krow < Nqp) {
            if (dk && dv) {
#pragma unroll
                for (int dd = 0; dd < DVT; ++dd) {
                    const int j0 = 16 * dd + 4 * g;
                    if (j0 < hd) {
                        st4(dk + ((bph * Nqp + krow) * (long)hd + j0), pack4(dkacc[i][dd]));
                        st4(dv + ((bph * Nqp + krow) * (long)hd + j0), pack4(dvacc[i][dd]));
                    }
                }
            }
        } else if (krow < Nqp + d.Npp) {
            const int t = krow - Nqp;
#pragma unroll
            for (int dd = 0; dd < DVT; ++dd) {
                const int j0 = 16 * dd + 4 * g;
                if (j0 < hd) {
                    *reinterpret_cast<f32x4*>(dkp_part + ((bph * d.Npp + t) * (long)hd + j0)) = dkacc[i][dd];
                    *reinterpret_cast<f32x4*>(dvp_part + ((bph * d.Npp + t) * (long)hd + j0)) = dvacc[i][dd];
                }
            }
            if (g == 0) dtok_part[bph * d.Npp + t] = dt;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// QKV + LayerNorm + gather backward:  (dq, dk, dv, dt1) -> dx
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256, CT >= 12 ? 2 : (CT >= 6 ? 3 : 4)) void k_swin_qkv_bwd(MivpSwinDesc d, const bf16_t* __restrict__ dq,
                                                      const bf16_t* __restrict__ dk, const bf16_t* __restrict__ dv,
                                                      const bf16_t* __restrict__ x, const int* __restrict__ tok_src,
                                                      const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                      const bf16_t* __restrict__ wqkv_t, const bf16_t* __restrict__ d_t1,
                                                      bf16_t* __restrict__ dx, bf16_t* __restrict__ dn_out) {
    // dn_out (optional, weight-gradient mode): gradient w.r.t. the attn_norm output, window order
    constexpr int KS3 = (3 * 16 * CT + 31) / 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, hd = C / d.heads, n3 = 3 * C;
    const TokInfo ti = token_info(d, ((long)blockIdx.x * 4 + wave) * 16 + r);      // (a dead lane decodes as token 0)
    const int src = sel(ti.live, tok_src[ti.pw * d.Nqp + ti.slot], -2);

    // Every load of the token tile is unconditional and issued up front (common.hpp "Branch-free loads"): the dq | dk |
    // dv pieces, the dt1 row, then -- one round trip later, behind tok_src -- the x row.
    const FastDiv by_c(C), by_hd(hd);
    const long to_dk = dk - dq, to_dv = dv - dq;                  // the three gradients as one base + offset
    const long row_qkv = (ti.bp * d.heads * d.Nqp + ti.slot) * (long)hd;         // + head * Nqp * hd + j0
    bf16x4 gpc[KS3][2];
#pragma unroll
    for (int s = 0; s < KS3; ++s)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int n0 = min(32 * s + 8 * g + 4 * hlf, n3 - 4);
            const int which = by_c.div(n0), cc = n0 - which * C, head = by_hd.div(cc), j0 = cc - head * hd;
            const long base = sel(which == 0, 0L, sel(which == 1, to_dk, to_dv));
            gpc[s][hlf] = ld4(dq + base + row_qkv + (long)head * d.Nqp * hd + j0);
        }
    bf16x4 t1raw[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) t1raw[ct] = ld4(d_t1 + ti.tt * (long)C + min(16 * ct + 4 * g, C - 4));
    bf16x4 xraw[CT];
    const long xrow = (ti.b * d.vol_in + max(src, 0)) * (long)C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) xraw[ct] = ld4(x + xrow + min(16 * ct + 4 * g, C - 4));
    bf16x8 gb[KS3];
#pragma unroll
    for (int s = 0; s < KS3; ++s) {
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int n0 = 32 * s + 8 * g + 4 * hlf;
            bf16x4 val = keep_if(gpc[s][hlf], ti.live && n0 < n3);
            const float sc = sel(n0 < C, d.q_scale, 1.0f);        // the q part carries the attention scale (x 1 is exact)
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = (bf16_t)((float)val[j] * sc);
            gpc[s][hlf] = val;
        }
        gb[s] = cat44(gpc[s][0], gpc[s][1]);
    }
    f32x4 dyv[CT], xv[CT];
    float sum = 0.f;
    constexpr bool LDSW = CT >= 6;                              // wide stages share the weights through LDS (common.hpp)
    using WS = WeightSlabs<KS3>;
    __shared__ __attribute__((aligned(16))) char wsm[LDSW ? WS::BYTES : 16];
    WS ws;
    if (LDSW) { ws.fetch(wqkv_t, 0); ws.store(wsm, 0); __syncthreads(); }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc = fzero4();
        const int row = 16 * ct + r;
        if (LDSW) {
            if (ct + 1 < CT) ws.fetch(wqkv_t, ct + 1);
#pragma unroll
            for (int s = 0; s < KS3; ++s) acc = mfma16(WS::frag(wsm, ct & 1, s, lane), gb[s], acc);
            if (ct + 1 < CT) ws.store(wsm, (ct + 1) & 1);
            __syncthreads();
        } else {
#pragma unroll
            for (int s = 0; s < KS3; ++s) {
                acc = mfma16(wfrag(wqkv_t, KS3, ct, s, lane), gb[s], acc);
            }
        }
        dyv[ct] = acc;
        const int c0 = 16 * ct + 4 * g;
        if (dn_out && ti.live && c0 < C) st4(dn_out + ti.tt * (long)C + c0, pack4(acc));
        f32x4 xx;
        { const bf16x4 raw = keep_if(xraw[ct], src >= 0 && c0 < C); for (int j = 0; j < 4; ++j) xx[j] = (float)raw[j]; }
        xv[ct] = xx;
        sum += xx[0] + xx[1] + xx[2] + xx[3];
    }
    const float mean = col_sum(sum) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
        if (16 * ct + 4 * g < C)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dvv = xv[ct][j] - mean; var += dvv * dvv; }
    const float rstd = rsqrtf(col_sum(var) / (float)C + d.ln_eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c0 = 16 * ct + 4 * g;
        if (c0 < C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dxh = dyv[ct][j] * ln_w[c0 + j];
                const float xh = (xv[ct][j] - mean) * rstd;
                dyv[ct][j] = dxh;
                xv[ct][j] = xh;
                s1 += dxh;
                s2 += dxh * xh;
            }
        }
    }
    const float m1 = col_sum(s1) / (float)C, m2 = col_sum(s2) / (float)C;
    if (src < 0) return;                       // zero-pad / padding-slot tokens have no voxel to write
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c0 = 16 * ct + 4 * g;
        if (c0 < C) {
            const bf16x4 raw = t1raw[ct];
            f32x4 out;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = (float)raw[j] + rstd * (dyv[ct][j] - m1 - xv[ct][j] * m2);
            st4(dx + ((ti.b * d.vol_in + src) * (long)C + c0), pack4(out));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// prompt K/V backward: (dKp, dVp) -> to_k / to_v -> LayerNorm backward -> dprompt   (64 rows: VALU)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_prompt_kv_bwd(MivpSwinDesc d, const float* __restrict__ dkp,
                                                        const float* __restrict__ dvp, const float* __restrict__ prompt,
                                                        const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                        const bf16_t* __restrict__ wqkv, float* __restrict__ dprompt,
                                                        bf16_t* __restrict__ wg_a, bf16_t* __restrict__ wg_n,
                                                        float* __restrict__ wg_ln) {
    // weight-gradient mode (all three or none): wg_a [2][Np][C] bf16 = dK rows, dV rows (head-merged), wg_n [Np][C] bf16 =
    // LN(prompt) (the to_k / to_v input), wg_ln [2][Np][C] f32 = per-row dbeta terms, dgamma terms of attn_norm
    // One workgroup (16 waves) per prompt row.  The two C x C mat-vecs are the cost: lane = channel, the 16 waves split
    // the contraction index and their partial sums meet in LDS in wave order (the single-wave-per-channel form was a
    // 2C-long serial chain per thread: 28 us per launch, twelve launches per step with prompts on both sides).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gk = reinterpret_cast<float*>(smem);            // [C] dK row (head-merged)
    float* gv = gk + d.C;                                    // [C]
    float* dy = gv + d.C;                                    // [C] gradient w.r.t. the LayerNorm output
    float* red = dy + d.C;                                   // [16]
    float* accs = red + 16;                                  // [16][64]
    const int t = blockIdx.x, C = d.C, hd = C / d.heads, tid = threadIdx.x;
    for (int n = tid; n < C; n += 1024) {
        const int head = n / hd, j = n - head * hd;
        gk[n] = dkp[((long)head * d.Npp + t) * hd + j];
        gv[n] = dvp[((long)head * d.Npp + t) * hd + j];
    }
    // LayerNorm statistics of the prompt row (waves 0-3)
    float part = 0.f;
    if (tid < 256) {
        for (int c = tid; c < C; c += 256) part += prompt[(long)t * C + c];
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((tid & 63) == 0) red[tid >> 6] = part;
    }
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    if (tid < 256) {
        part = 0.f;
        for (int c = tid; c < C; c += 256) { const float dvv = prompt[(long)t * C + c] - mean; part += dvv * dvv; }
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((tid & 63) == 0) red[4 + (tid >> 6)] = part;
    }
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)C + d.ln_eps);
    // dy[c] = sum_n gk[n] Wk[n][c] + gv[n] Wv[n][c]
    const int cl = tid & 63, slice = tid >> 6;
    for (int cb = 0; cb < C; cb += 64) {
        const int c = cb + cl;
        float acc = 0.f;
        if (c < C)
            for (int n = slice; n < C; n += 16)
                acc += gk[n] * (float)wqkv[(long)(C + n) * C + c] + gv[n] * (float)wqkv[(long)(2 * C + n) * C + c];
        accs[slice * 64 + cl] = acc;
        __syncthreads();
        if (slice == 0 && c < C) {
            float tot = 0.f;
            for (int k = 0; k < 16; ++k) tot += accs[k * 64 + cl];
            dy[c] = tot;
        }
        __syncthreads();
    }
    // LayerNorm backward (waves 0-3; each thread owns channels c, c+256, ...)
    float s1 = 0.f, s2 = 0.f;
    float dxh_loc[4], xh_loc[4];
    int cnt = 0;
    if (tid < 256) {
        for (int c = tid; c < C; c += 256, ++cnt) {
            const float acc = dy[c];
            const float dxh = acc * ln_w[c];
            const float xh = (prompt[(long)t * C + c] - mean) * rstd;
            if (wg_a) {
                wg_a[(long)t * C + c] = (bf16_t)gk[c];
                wg_a[((long)d.Np + t) * C + c] = (bf16_t)gv[c];
                wg_n[(long)t * C + c] = (bf16_t)(xh * ln_w[c] + ln_b[c]);
                wg_ln[(long)t * C + c] = acc;
                wg_ln[((long)d.Np + t) * C + c] = acc * xh;
            }
            dxh_loc[cnt] = dxh;
            xh_loc[cnt] = xh;
            s1 += dxh;
            s2 += dxh * xh;
        }
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        if ((tid & 63) == 0) { red[8 + (tid >> 6)] = s1; red[12 + (tid >> 6)] = s2; }
    }
    __syncthreads();
    if (tid >= 256) return;
    const float m1 = (red[8] + red[9] + red[10] + red[11]) / (float)C;
    const float m2 = (red[12] + red[13] + red[14] + red[15]) / (float)C;
    cnt = 0;
    for (int c = tid; c < C; c += 256, ++cnt) dprompt[(long)t * C + c] = rstd * (dxh_loc[cnt] - m1 - xh_loc[cnt] * m2);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int bwd_checks(const MivpSwinDesc* d) {
    MIVP_REQUIRE(d != nullptr);
    MIVP_REQUIRE(d->B > 0 && d->C > 0 && d->heads > 0 && d->P > 0);
    MIVP_REQUIRE(d->C % 8 == 0 && d->C % d->heads == 0 && (d->C / d->heads) % 4 == 0);
    MIVP_REQUIRE(d->Nqp % 16 == 0 && d->Npp % 16 == 0 && d->Nkp % 32 == 0 && d->augp % 4 == 0);
    MIVP_REQUIRE((long)d->B * d->P * d->Nqp < (1L << 31));      // kernels decode token indices in 32 bits
    return MIVP_OK;
}

#define CT_SWITCH(CTV, LAUNCH)                                                              \
    switch (CTV) {                                                                          \
        case 1: LAUNCH(1); break;                                                           \
        case 2: LAUNCH(2); break;                                                           \
        case 3: LAUNCH(3); break;                                                           \
        case 4: LAUNCH(4); break;                                                           \
        case 6: LAUNCH(6); break;                                                           \
        case 8: LAUNCH(8); break;                                                           \
        case 12: LAUNCH(12); break;                                                         \
        default: mivp_set_error("C/16 not in {1,2,3,4,6,8,12}"); return MIVP_EUNSUPPORTED; \
    }

extern "C" int mivp_swin_proj_mlp_bwd(const MivpSwinDesc* d, const void* dy, const int32_t* tok_dst, const void* t1,
                                      const float* ln_w, const float* ln_b, const void* wmlp_t, const void* wproj_t,
                                      void* d_o, void* d_t1, void* dn_out, void* dyw, void* d_pj,
                                      mivp_stream_t stream) {
    int rc = bwd_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(dy && tok_dst && t1 && ln_w && ln_b && wmlp_t && wproj_t && d_o && d_t1);
    MIVP_REQUIRE((dn_out == nullptr) == (dyw == nullptr));
    const long T = (long)d->B * d->P * d->Nqp;
    const unsigned grid = (unsigned)((T + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    // C = 48 / 96 / 192 / 384, no weight-gradient outputs: row-image kernel (proj dropout included, round 3), natural-order image of wproj_t
    if (mivp_tok_rows_supported(d) && !dn_out && !dyw && !d_pj)
        return mivp_tok_wide_proj_mlp_bwd(d, dy, tok_dst, t1, ln_w, wmlp_t, (const bf16_t*)wproj_t + mivp_tok_natural_offset(d->C),
                                          d_o, d_t1, st);
#define L_PMB(K) hipLaunchKernelGGL((k_swin_proj_mlp_bwd<K>), dim3(grid), dim3(256), 0, st, *d, (const bf16_t*)dy, tok_dst, \
                                     (const bf16_t*)t1, ln_w, ln_b, (const bf16_t*)wmlp_t, (const bf16_t*)wproj_t,          \
                                     (bf16_t*)d_o, (bf16_t*)d_t1, (bf16_t*)dn_out, (bf16_t*)dyw, (bf16_t*)d_pj)
    CT_SWITCH((d->C + 15) / 16, L_PMB)
#undef L_PMB
    return mivp_check_launch("swin_proj_mlp_bwd");
}

extern "C" int mivp_win_attn_delta(const MivpSwinDesc* d, const void* o, const void* d_o, float* delta,
                                   mivp_stream_t stream) {
    int rc = bwd_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(o && d_o && delta);
    const long total = (long)d->B * d->P * d->heads * d->Nqp;
    MIVP_REQUIRE(total < (1L << 31));                            // 32-bit decode in the kernel
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(k_win_attn_delta, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, (const bf16_t*)o,
                       (const bf16_t*)d_o, delta);
    return mivp_check_launch("win_attn_delta");
}

static int pick_chunk(int total_tiles, size_t fixed_bytes, size_t bytes_per_tile, size_t budget) {
    int c = (int)((budget - fixed_bytes) / bytes_per_tile);
    c &= ~1;
    if (c > total_tiles) c = total_tiles;
    if (c < 2) c = 2;
    // balance the chunks: smallest even chunk size that needs the same number of passes
    const int passes = (total_tiles + c - 1) / c;
    int bal = (total_tiles + passes - 1) / passes;
    bal = (bal + 1) & ~1;
    return bal < c ? bal : c;
}

static constexpr size_t ATTN_BWD_LDS_BUDGET = 78 * 1024;      // two workgroups per CU (160 KiB LDS)

template <int DKS, int DVT>
static int launch_dq(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp, const void* vp,
                     const void* qa, const void* ka, const int32_t* tok_rid, const void* o, const void* d_o,
                     const float* lse, float* delta, void* dq, hipStream_t st) {
    using G = AttnBwdGeom<DKS, DVT>;
    constexpr int NW = 8, QPW = 3;
    if (d->Nqp / 16 > NW * QPW) { mivp_set_error("win_attn_bwd_dq: more than 384 queries per window"); return MIVP_EUNSUPPORTED; }
    const size_t per_tile = 16 * (size_t)G::KROW + 16 * (size_t)G::VROWB + (size_t)16 * DVT * 32 + 64;
    const size_t fixed = (size_t)16 * DVT * 16;
    const int nt = d->Nkp / 16;
    const int chunk = pick_chunk(nt, fixed, per_tile, ATTN_BWD_LDS_BUDGET);
    const size_t lds = fixed + per_tile * chunk;
    const bool msk = d->has_mask != 0;
    auto kern = d->attn_drop_thr ? (msk ? k_win_attn_bwd_dq<DKS, DVT, QPW, NW, true, true> : k_win_attn_bwd_dq<DKS, DVT, QPW, NW, true, false>)
                                 : (msk ? k_win_attn_bwd_dq<DKS, DVT, QPW, NW, false, true> : k_win_attn_bwd_dq<DKS, DVT, QPW, NW, false, false>);
    MIVP_LDS_OPT_IN(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)d->B * d->P * d->heads)), dim3(64 * NW), lds, st, *d, chunk, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)kp, (const bf16_t*)vp, (const bf16_t*)qa,
                       (const bf16_t*)ka, tok_rid, (const bf16_t*)o, (const bf16_t*)d_o, lse, delta, (bf16_t*)dq);
    return mivp_check_launch("win_attn_bwd_dq");
}

extern "C" int mivp_win_attn_bwd_dq(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp,
                                    const void* vp, const void* qa, const void* ka, const int32_t* tok_rid,
                                    const void* o, const void* d_o, const float* lse, float* delta, void* dq,
                                    mivp_stream_t stream) {
    int rc = bwd_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(q && k && v && qa && ka && o && d_o && lse && delta && dq);
    MIVP_REQUIRE(d->Np == 0 || (kp && vp));
    MIVP_REQUIRE(!d->has_mask || tok_rid);
    int dks, nt;
    if (mivp_attn_tile_config(d, &dks, &nt)) { mivp_set_error("win_attn_bwd_dq: shape outside the instantiated set"); return MIVP_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    if (dks == 1) return launch_dq<1, 1>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, d_o, lse, delta, dq, st);
    if (dks == 2) return launch_dq<2, 2>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, d_o, lse, delta, dq, st);
    return launch_dq<3, 3>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, d_o, lse, delta, dq, st);
}

template <int DKS, int DVT, int KPW, bool AUG, bool DROP, bool MASKED>
static int launch_dkv(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp, const void* vp,
                      const void* qa, const void* ka, const int32_t* tok_rid, const void* d_o, const float* lse,
                      const float* delta, void* dk, void* dv, float* dkp_part, float* dvp_part, float* dtok_part,
                      float* dka_part, hipStream_t st) {
    using G = AttnBwdGeom<DKS, DVT>;
    const size_t per_tile = 16 * (size_t)G::KROW + 16 * (size_t)G::VROWB + 2 * (size_t)16 * DVT * 32 + 3 * 64 + (AUG ? 32 * 32 : 0);
    const size_t fixed = 2 * (size_t)16 * DVT * 16 + (AUG ? 32 * 16 : 0);
    const int nqt = (d->Nqp + 31) / 32 * 2;
    const int chunk = pick_chunk(nqt, fixed, per_tile, ATTN_BWD_LDS_BUDGET);
    const size_t lds = fixed + per_tile * chunk;
    const int nt = d->Nkp / 16;
    const int kt0 = (dk && dv) ? 0 : d->Nqp / 16;            // prompt-only mode skips the window keys
    const int ktiles = nt - kt0;
    if (ktiles <= 0) return MIVP_OK;
    constexpr int NW = 8;
    const int ksplit = (ktiles + NW * KPW - 1) / (NW * KPW);
    auto kern = k_win_attn_bwd_dkv<DKS, DVT, KPW, NW, AUG, DROP, MASKED>;
    MIVP_LDS_OPT_IN(kern, lds);
    dim3 grid((unsigned)((long)d->B * d->P * d->heads), (unsigned)ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, *d, chunk, kt0, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       (const bf16_t*)kp, (const bf16_t*)vp, (const bf16_t*)qa, (const bf16_t*)ka, tok_rid, (const bf16_t*)d_o,
                       lse, delta, (bf16_t*)dk, (bf16_t*)dv, dkp_part, dvp_part, dtok_part, dka_part);
    return mivp_check_launch("win_attn_bwd_dkv");
}

extern "C" int mivp_win_attn_bwd_dkv(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp,
                                     const void* vp, const void* qa, const void* ka, const int32_t* tok_rid,
                                     const void* d_o, const float* lse, const float* delta, void* dk, void* dv,
                                     float* dkp_part, float* dvp_part, float* dtok_part, float* dka_part,
                                     mivp_stream_t stream) {
    int rc = bwd_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(q && k && v && qa && ka && d_o && lse && delta);
    MIVP_REQUIRE((dk == nullptr) == (dv == nullptr));
    MIVP_REQUIRE(d->Np == 0 || (kp && vp && dkp_part && dvp_part && dtok_part));
    MIVP_REQUIRE(!d->has_mask || tok_rid);
    MIVP_REQUIRE(dka_part == nullptr || (dk != nullptr && d->augp <= 32));
    int dks, nt;
    if (mivp_attn_tile_config(d, &dks, &nt)) { mivp_set_error("win_attn_bwd_dkv: shape outside the instantiated set"); return MIVP_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    // The shifted head_dim 24 / 32 kernel runs one workgroup per CU whatever its key-tile count per wave (162 VGPRs at two): with
    // three tiles per wave a 7^3 window's 22 key tiles are ONE workgroup instead of two that each stage every query row
    const int ktiles_all = d->Nkp / 16 - ((dk && dv) ? 0 : d->Nqp / 16);
    const bool three_tiles = ktiles_all > 16 && ktiles_all <= 24;
#define DKV_ARGS d, q, k, v, kp, vp, qa, ka, tok_rid, d_o, lse, delta, dk, dv, dkp_part, dvp_part, dtok_part, dka_part, st
#define DKV_PICK3(AUGV, DROPV, MV)                                              \
    do {                                                                        \
        if (dks == 1) return launch_dkv<1, 1, 2, AUGV, DROPV, MV>(DKV_ARGS);    \
        if (dks == 2 && three_tiles) return launch_dkv<2, 2, (MV && !AUGV && !DROPV) ? 3 : 2, AUGV, DROPV, MV>(DKV_ARGS); \
        if (dks == 2) return launch_dkv<2, 2, 2, AUGV, DROPV, MV>(DKV_ARGS);    \
        if (three_tiles && !MV && !AUGV && !DROPV) return launch_dkv<3, 3, (!MV && !AUGV && !DROPV) ? 3 : 1, AUGV, DROPV, MV>(DKV_ARGS); \
        if (ktiles_all > 8 && MV && !AUGV && !DROPV) return launch_dkv<3, 3, (MV && !AUGV && !DROPV) ? 2 : 1, AUGV, DROPV, MV>(DKV_ARGS); \
        return launch_dkv<3, 3, 1, AUGV, DROPV, MV>(DKV_ARGS);                  \
    } while (0)
#define DKV_PICK(AUGV, DROPV) do { if (d->has_mask) DKV_PICK3(AUGV, DROPV, true); else DKV_PICK3(AUGV, DROPV, false); } while (0)
    if (dka_part) { if (d->attn_drop_thr) DKV_PICK(true, true); else DKV_PICK(true, false); }
    if (d->attn_drop_thr) DKV_PICK(false, true);
    DKV_PICK(false, false);
#undef DKV_PICK3
#undef DKV_PICK
#undef DKV_ARGS
}

extern "C" int mivp_swin_qkv_bwd(const MivpSwinDesc* d, const void* dq, const void* dk, const void* dv, const void* x,
                                 const int32_t* tok_src, const float* ln_w, const float* ln_b, const void* wqkv_t,
                                 const void* d_t1, void* dx, void* dn_out, mivp_stream_t stream) {
    int rc = bwd_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(dq && dk && dv && x && tok_src && ln_w && ln_b && wqkv_t && d_t1 && dx);
    const long T = (long)d->B * d->P * d->Nqp;
    const unsigned grid = (unsigned)((T + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    if (mivp_tok_wide_supported(d))                           // C = 96 / 192 / 384: column-split form (swin_tok_wide.hip)
        return mivp_tok_wide_qkv_bwd(d, dq, dk, dv, x, tok_src, ln_w, wqkv_t, d_t1, dx, dn_out, st);
#define L_QB(K) hipLaunchKernelGGL((k_swin_qkv_bwd<K>), dim3(grid), dim3(256), 0, st, *d, (const bf16_t*)dq, (const bf16_t*)dk, \
                                    (const bf16_t*)dv, (const bf16_t*)x, tok_src, ln_w, ln_b, (const bf16_t*)wqkv_t,             \
                                    (const bf16_t*)d_t1, (bf16_t*)dx, (bf16_t*)dn_out)
    CT_SWITCH((d->C + 15) / 16, L_QB)
#undef L_QB
    return mivp_check_launch("swin_qkv_bwd");
}

extern "C" int mivp_prompt_kv_bwd(const MivpSwinDesc* d, const float* dkp, const float* dvp, const float* prompt,
                                  const float* ln_w, const float* ln_b, const void* wqkv, float* dprompt,
                                  void* wg_a, void* wg_n, float* wg_ln, mivp_stream_t stream) {
    int rc = bwd_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(d->Np > 0 && dkp && dvp && prompt && ln_w && ln_b && wqkv && dprompt);
    MIVP_REQUIRE((wg_a == nullptr) == (wg_n == nullptr) && (wg_a == nullptr) == (wg_ln == nullptr));
    MIVP_REQUIRE(d->C <= 1024);
    hipLaunchKernelGGL(k_prompt_kv_bwd, dim3(d->Np), dim3(1024), (3 * d->C + 16 + 16 * 64) * sizeof(float), (hipStream_t)stream, *d,
                       dkp, dvp, prompt, ln_w, ln_b, (const bf16_t*)wqkv, dprompt, (bf16_t*)wg_a, (bf16_t*)wg_n, wg_ln);
    return mivp_check_launch("prompt_kv_bwd");
}

// test hook: materialise the keep masks exactly as the kernels derive them
__global__ __launch_bounds__(256) void k_dropout_masks(MivpSwinDesc d, uint8_t* __restrict__ attn_keep, uint8_t* __restrict__ proj_keep) {
    // dropout keys of this call (common.hpp drop_seed: uniform, scalar-ALU work; unused without dropout)
    const uint32_t attn_key = drop_seed(d.attn_seed, d.seed_epoch), proj_key = drop_seed(d.proj_seed, d.seed_epoch);
    (void)attn_key; (void)proj_key;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    if (attn_keep) {
        const long total = (long)d.B * d.P * d.heads * d.Nqp * d.Nkp;
        for (long e = gtid; e < total; e += stride) {
            const int k = (int)(e % d.Nkp);
            const long rest = e / d.Nkp;
            const int q = (int)(rest % d.Nqp);
            const long bph = rest / d.Nqp;
            const uint32_t h = drop_hash(attn_pair(attn_row(bph, q, d.Nqp, d.Nkp), k), attn_key);
            attn_keep[e] = (d.attn_drop_thr == 0 || drop_keep(h, k & 1, d.attn_drop_thr)) ? 1 : 0;
        }
    }
    if (proj_keep) {
        const long total = (long)d.B * d.P * d.Nqp * d.C;
        for (long e = gtid; e < total; e += stride) {
            const uint32_t h = drop_hash((uint32_t)(e >> 1), proj_key);
            proj_keep[e] = (d.proj_drop_thr == 0 || drop_keep(h, (int)(e & 1), d.proj_drop_thr)) ? 1 : 0;
        }
    }
}

extern "C" int mivp_dropout_masks(const MivpSwinDesc* d, uint8_t* attn_keep, uint8_t* proj_keep, mivp_stream_t stream) {
    int rc = bwd_checks(d);
    if (rc) return rc;
    hipLaunchKernelGGL(k_dropout_masks, dim3(1024), dim3(256), 0, (hipStream_t)stream, *d, attn_keep, proj_keep);
    return mivp_check_launch("dropout_masks");
}

extern "C" int mivp_sizeof_desc(int which) {
    switch (which) {
        case 0: return (int)sizeof(MivpSwinDesc);
        case 1: return (int)sizeof(MivpMergeDesc);
        case 2: return (int)sizeof(MivpConvDesc);
        case 3: return (int)sizeof(MivpEmbedDesc);
        case 4: return (int)sizeof(MivpUpcatDesc);
        case 5: return (int)sizeof(MivpOperandDesc);
        case 6: return (int)sizeof(MivpGemmTnDesc);
        default: return -1;
    }
}
