// Window-attention backward in ONE pass (head_dim <= 16: every encoder stage and the last decoder stage).
// Reference semantics: multi_head_attention/window_attention.py:49-61 (autograd of it), swin_block.py:187-225
// (prompt keys, multiplicative shift mask).  Replaces the dq-owner + dkv-owner pair of swin_bwd.hip for these shapes:
// S, dP and the exponentials are formed ONCE per (query tile, key tile).
//
// One workgroup (8 waves) owns one (window, head): ALL its queries and ALL its keys.
//   * wave w owns key tiles w, w+8, w+16, (w+24): K' / V fragments live in registers for the whole kernel, dK^T / dV^T
//     accumulate in registers (no cross-wave sum);
//   * the workgroup walks the query tiles t = 0, 1, ... together.  Per tile every wave forms, for each of its key tiles,
//         S  = Q' K'^T - lse        (16x16x32 MFMA, key on the lane, accumulator starts at -lse * log2 e)
//         dP = dO V^T - delta       (16x16x16 MFMA, accumulator starts at -delta)
//         P  = exp2(S), dS = P * dP
//     feeds P / dS -- already in B-operand position -- to dV^T += dO^T P and dK^T += Q^T dS (16x16x16 MFMAs whose A
//     operands are TRANSPOSING reads, ds_read_b64_tr_b16, of the same Q' / dO row images the S / dP products read by rows:
//     no transposed copies are staged);
//   * dQ^T = K^T dS^T needs dS with the QUERY on the lane: the wave parks its dS tiles (bf16) in a private LDS slot, reads
//     them back transposed and forms the partial dQ^T of the query tile over ITS keys (one 16x16x16 MFMA per key tile);
//     the eight partials meet in LDS ([wave][query][head dim] f32, double buffered) and after the tile's barrier ONE wave
//     (rotating) adds them in wave order and stores dq -- one barrier per query tile, nothing serial in between.
//     (First form of this kernel: one wave formed dQ of a whole query tile from a shared dS image, a 13-step dependent
//     LDS-read -> MFMA chain that every other wave waited for at the next barrier: 2.3x slower.)
// Prompt keys (key tiles beyond Nqp) produce per-window f32 partials of dKp / dVp and the column sums of dS (the gradient
// of the prompt-token bias), as the two-pass kernels did.  delta = sum_j dO * O is computed while staging dO.
//
// LDS (7^3 window + 64 prompt keys: 77 KB -> two workgroups per CU):
//   Qimg  [nq][64 B]   Q' rows (head dims | bias one-hots), 16-byte chunks XOR-swizzled (common.hpp OperandRows<32>)
//   Oimg  [nq][32 B]   dO rows (16 columns), the two 16-byte halves swapped in rows 8..15 of every 16
//   Kt    [16][Nkp+8]  K^T (head dims x keys) for the dQ product
//   exch  [Nkp/16][512 B]  dS of the current query tile, one slot per key tile (private to the owning wave): element
//         (key r, query quad p) at 128 p + 8 (r ^ 8 (p >> 1)) -- conflict-free for the 8-byte writes and the transposed reads
//   dqp   [2][8 waves][16 queries][16] f32  partial dQ^T tiles
//   lse_s, del_s [nq] f32, ridq [nq] bytes
// Every reduction has a fixed order: results are bit-reproducible.
#include "common.hpp"
#include <cstdlib>

int mivp_attn_tile_config(const MivpSwinDesc* d, int* dks, int* nt);

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

MIVP_DEV f32x4 mfma16k16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
// f32x4 -> four bf16 as exactly two v_cvt_pk_bf16_f32.  The value feeds BOTH an LDS store and the short-typed operand of
// the K = 16 MFMA builtin; converted element by element that second use became four single conversions plus two v_perm
// per pack (10 + 4 VALU instructions per tile instead of 4, in a VALU-issue-bound loop).  Converting PAIRS and carrying
// the result as two dwords keeps it at two instructions.  (Not inline asm: hipcc pads no VALU -> MFMA-operand wait states
// around an asm statement, and a first asm form of this returned garbage on the large shapes.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
MIVP_DEV u32x2 pack4_pk(f32x4 v) {
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    u32x2 u;
    u[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
    u[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2));
    return u;
}
MIVP_DEV f32x4 mfma16k16(bf16x4 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
// 4 rows x 16 columns of bf16 per 16-lane group, delivered column-major: lane 4q+p of the group passes the address of row
// q, columns 4p..4p+3; lane i receives column i of the four rows (EXEC must be full: callers keep the wave converged)
MIVP_DEV bf16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
}

// LDS-DMA staging (DMA form of the fused kernel): constant source words and one 4-byte global -> LDS transfer per lane
// (LDS address = wave-uniform base + 4 * lane), as in the forward kernel (swin_fwd.hip)
__device__ __attribute__((aligned(16))) unsigned int g_bwd_zero[4] = {0u, 0u, 0u, 0u};
MIVP_DEV void glds4(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

constexpr int FUSED_KPW = 4;                                   // key tiles per wave (8 waves: up to 512 keys)
using KR = OperandRows<32>;

MIVP_DEV int oimg_off(int row, int byte) { return row * 32 + (byte ^ (((row >> 3) & 1) << 4)); }
MIVP_DEV int exch_off(int tile, int key, int quad) { return tile * 512 + 128 * quad + 8 * (key ^ ((quad >> 1) << 3)); }

// ---- staging of the query-side images (shared by the two kernels of this file) ----
// A workgroup of these kernels runs ~10-15 us and a dependent global load -> LDS store round trip costs ~1 us at two
// workgroups per CU.  Written as plain loops (one conditional load and one store per iteration, every conditional load in
// its own basic block) the staging was a chain of ~12 such round trips: most of the kernel.  Here every load is
// UNCONDITIONAL (clamped address, value replaced by zero afterwards), all of a thread's loads of a batch are issued before
// its first LDS store, and the first batches of the Q' and dO / O images go out back to back.
struct QSide {
    const bf16_t *qb, *qa, *dob, *ob;
    int Nqp, hd, hd4, A, a4, C;
};
// piece e of the Q' image: row e >> 3, 4-column group e & 7 = [head dims | bias one-hots | zero]
MIVP_DEV bf16x4 q_piece(const QSide& s, int e) {
    const int lrow = e >> 3, c4 = e & 7;
    const int row = min(lrow, s.Nqp - 1);
    const bool fq = c4 < s.hd4;
    const int ca = min(c4 - s.hd4, s.a4 - 1);
    const long to_qa = s.qa - s.qb;                            // (uniform) the two sources as one base + offset
    const long off = sel(fq, (long)(row * s.hd + 4 * c4), to_qa + (long)(row * s.A + 4 * ca));
    return keep_if(ld4(s.qb + off), lrow < s.Nqp && c4 < s.hd4 + s.a4);
}
// piece e of the dO image (row e >> 2, 4-column group e & 3) and the matching piece of O (for delta)
MIVP_DEV void o_piece(const QSide& s, int e, bf16x4& gv, bf16x4& ov) {
    const int lrow = e >> 2, c4 = e & 3;
    const int row = min(lrow, s.Nqp - 1);
    const int cc = min(c4, s.hd4 - 1);
    const uint32_t off = (uint32_t)row * s.C + 4 * cc;
    const bool ok = lrow < s.Nqp && c4 < s.hd4;
    gv = keep_if(ld4(s.dob + off), ok);
    ov = keep_if(ld4(s.ob + off), ok);
}
// four consecutive lanes share a query row; callers keep whole waves converged (the element count is a multiple of 64)
MIVP_DEV void o_store(char* Oimg, float* del_s, int e, bf16x4 gv, bf16x4 ov) {
    const int lrow = e >> 2, c4 = e & 3;
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) part += (float)gv[i] * (float)ov[i];
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    *reinterpret_cast<bf16x4*>(Oimg + oimg_off(lrow, 8 * c4)) = gv;
    if (c4 == 0) del_s[lrow] = -part;                          // delta = sum_j dO * O, negated: the dP accumulators start from it
}
// One batch: UQ pieces of Q' and UO pieces of dO / O per thread, all loads before the first LDS store.  Indices beyond the
// images are CLAMPED to the last piece for the loads AND the stores (whole waves at a time: the piece counts are multiples
// of 64): a conditional store would pull its load into the conditional block, behind its own s_waitcnt vmcnt(0).
template <int NTHR, int UQ, int UO>
MIVP_DEV void stage_batch(const QSide& s, char* Qimg, char* Oimg, float* del_s, int q_elems, int o_elems, int eq0, int eo0) {
    bf16x4 qv[UQ], gv[UO], ov[UO];
#pragma unroll
    for (int u = 0; u < UQ; ++u) qv[u] = q_piece(s, min(eq0 + NTHR * u, q_elems - 1));
#pragma unroll
    for (int u = 0; u < UO; ++u) o_piece(s, min(eo0 + NTHR * u, o_elems - 1), gv[u], ov[u]);
    __builtin_amdgcn_sched_barrier(0);                         // every load of the batch is out before the first store waits
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
        const int e = min(eq0 + NTHR * u, q_elems - 1);
        *reinterpret_cast<bf16x4*>(Qimg + KR::off(e >> 3, 4 * (e & 7))) = qv[u];
    }
#pragma unroll
    for (int u = 0; u < UO; ++u) o_store(Oimg, del_s, min(eo0 + NTHR * u, o_elems - 1), gv[u], ov[u]);
}
template <int NTHR>
MIVP_DEV void stage_query_side(const QSide& s, char* Qimg, char* Oimg, float* del_s, int nq, int tid) {
    constexpr int UQ = 6, UO = 3;                              // 7^3 windows (352 rows): the first batch is all there is
    const int q_elems = nq * 8, o_elems = nq * 4;
    stage_batch<NTHR, UQ, UO>(s, Qimg, Oimg, del_s, q_elems, o_elems, tid, tid);       // straight-line: joins the caller's loads
    for (int b = 1; b * NTHR * UQ < q_elems; ++b)
        stage_batch<NTHR, UQ, UO>(s, Qimg, Oimg, del_s, q_elems, o_elems, tid + b * NTHR * UQ, tid + b * NTHR * UO);
}

}  // namespace

// ABL: timing ablations (results are WRONG for ABL != 0; selected by MIVP_ATTN_BWD_ABL for cost breakdowns, tools/ab_attn_bwd.sh):
//   1 no dQ reduce / store   2 no barrier in the tile loop   3 no dS exchange and dQ products   4 at most three key tiles per wave
//   5 no exponentials / products / conversions (P = dS = S bits)
template <int NW, bool DROP, bool MASKED, bool DMA = false, int ABL = 0>
__global__ __launch_bounds__(64 * NW, 4) void k_win_attn_bwd_fused(
    MivpSwinDesc d, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ kp, const bf16_t* __restrict__ vp, const bf16_t* __restrict__ qa,
    const bf16_t* __restrict__ ka, const int* __restrict__ tok_rid, const bf16_t* __restrict__ o,
    const bf16_t* __restrict__ d_o, const float* __restrict__ lse, bf16_t* __restrict__ dq, bf16_t* __restrict__ dk,
    bf16_t* __restrict__ dv, float* __restrict__ dkp_part, float* __restrict__ dvp_part, float* __restrict__ dtok_part) {
    // dropout keys of this call (common.hpp drop_seed: uniform, scalar-ALU work; unused without dropout)
    const uint32_t attn_key = drop_seed(d.attn_seed, d.seed_epoch), proj_key = drop_seed(d.proj_seed, d.seed_epoch);
    (void)attn_key; (void)proj_key;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a SCALAR: everything derived from it (key-tile ownership, LDS slots) stays in SGPRs and the
    // per-tile "does this wave own a fourth key tile" tests are scalar branches, not exec-masked regions
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, heads = d.heads, hd = C / heads, Nqp = d.Nqp, Nkp = d.Nkp, A = d.augp;
    const int hd4 = hd / 4, a4 = A / 4;
    const long bph = blockIdx.x;
    const int head = (int)(bph % heads);
    const long bp = bph / heads;
    const int pw = (int)(bp % d.P);
    const int nt = Nkp / 16;                                  // key tiles
    const int nqt = (Nqp + 31) / 32 * 2;                      // query tiles, rounded to an even count (phantom tile: P = 0)
    const int nq = nqt * 16;
    const int KTROW = (Nkp + 8) * 2;

    char* Qimg = smem;
    char* Oimg = Qimg + (size_t)nq * 64;
    // K^T image [hd][Nkp + 8] of the classic form; DMA form: K ROW-major [Nkp][16] (32-byte rows), the dQ product's A operand
    // comes out of transposing reads
    char* Kt = Oimg + (size_t)nq * 32;
    char* exch = Kt + (DMA ? (size_t)Nkp * 32 : (size_t)hd * KTROW);
    float* dqp = reinterpret_cast<float*>(exch + (size_t)nt * 512);
    float* lse_s = dqp + 2 * NW * 256;
    float* del_s = lse_s + nq;
    uint8_t* ridq = reinterpret_cast<uint8_t*>(del_s + nq);

    const bf16_t* qb = q + bph * (long)Nqp * hd;             // uniform per-(window, head) bases, 32-bit offsets
    const bf16_t* kb = k + bph * (long)Nqp * hd;
    const bf16_t* vb = v + bph * (long)Nqp * hd;
    const bf16_t* kpb = d.Np > 0 ? kp + (long)head * d.Npp * hd : kb;
    const bf16_t* vpb = d.Np > 0 ? vp + (long)head * d.Npp * hd : vb;
    const bf16_t* kab = ka + (long)head * Nkp * A;
    const bf16_t* dob = d_o + bp * (long)Nqp * C + head * hd;
    const bf16_t* ob = o + bp * (long)Nqp * C + head * hd;
    const float* lseb = lse + bph * (long)Nqp;
    const int n_prompt_rows = d.Np > 0 ? d.Npp : 0;

    // ---------------- staging (see stage_query_side: unconditional loads, issued in batches) ----------------
    const int max_prow = n_prompt_rows > 0 ? n_prompt_rows - 1 : 0;
    // a row of [K ; Kp] / [V ; Vp] (clamped: rows beyond the prompts re-read the last one and are zeroed by the caller)
    const long to_kp = kpb - kb, to_vp = vpb - vb, to_ka = kab - kb;       // (uniform) several sources as one base + offset
    auto kv_off = [&](int row, int c4, long to_prompt) -> long {
        const int pr = min(max(row - Nqp, 0), max_prow);
        return sel(row < Nqp, (long)(row * hd + 4 * c4), to_prompt + (long)(pr * hd + 4 * c4));
    };
    // this wave's key tiles: K' / V fragments (B operands) from global, once -- issued first, used after the staging
    bf16x8 kf[FUSED_KPW];
    bf16x4 vf[FUSED_KPW];
    // key classes of this lane's key in each owned tile, one BYTE per tile (region id; 255: prompt / padding key, never
    // masked): live = (class == 255) | (class == query class).  One register instead of one per tile: the masked instantiation
    // sits at the 128-VGPR limit of two workgroups per CU
    uint32_t kk4 = 0u;
    f32x4 dkacc[FUSED_KPW], dvacc[FUSED_KPW];
    // column sums of dS (gradient of the prompt-token bias), two partial sums per lane, of this wave's LAST key tile only: prompt
    // tiles are the highest-numbered ones and there are at most eight of them, so a wave owns at most one and it is its last
    f32x2 dsum = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FUSED_KPW; ++i) {
        const int kt = wave + NW * i;
        const int krow = (kt < nt ? kt : 0) * 16 + r;
        const bool staged = krow < Nqp + n_prompt_rows;
        bf16x4 piece[2];
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int c4 = 2 * g + hlf;
            const int ca = min(max(c4 - hd4, 0), a4 - 1);
            const long off = sel(c4 < hd4, kv_off(krow, min(c4, hd4 - 1), to_kp), to_ka + (long)(krow * A + 4 * ca));
            piece[hlf] = keep_if(ld4(kb + off), sel(c4 < hd4, (int)staged, (int)(c4 < hd4 + a4)) != 0);
        }
        kf[i] = cat44(piece[0], piece[1]);
        vf[i] = keep_if(ld4(vb + kv_off(krow, min(g, hd4 - 1), to_vp)), g < hd4 && staged);
        // content key: its region id; prompt and padding keys are never masked (padding keys are excluded by their bias)
        const bool content = krow < d.Nq;
        const uint32_t cls = MASKED ? (uint32_t)tok_rid[pw * Nqp + min(krow, d.Nq - 1)] : 0u;
        kk4 |= (uint32_t)sel(content, (int)(cls & 0xFFu), 255) << (8 * i);
        dkacc[i] = fzero4();
        dvacc[i] = fzero4();
    }
    // lse (negated, log2 units: the S accumulators start from it; padding query rows: P = 0) and the query classes
    const int m0 = min(tid, Nqp - 1);
    const float lse0 = lseb[m0];
    const int rid0 = MASKED ? tok_rid[pw * Nqp + m0] : 0;
    const QSide qs{qb, qa, dob, ob, Nqp, hd, hd4, A, a4, C};
    if constexpr (DMA) {
        // ---- LDS-DMA staging (global_load_lds_dword: a lane names its own 4-byte source, a wave-instruction fills 256
        //      consecutive LDS bytes; no VGPR destination, no ds_write, no register transposes).  Row groups go round-robin
        //      over the waves, so a lane's swizzle, source kind and column are constants and a pointer only advances by a
        //      per-lane stride.  The register-path staging was a third of this kernel's vector instructions.
        static_assert(!DMA || NW == 8, "group strides below assume eight waves");
        const char* zsrc = reinterpret_cast<const char*>(g_bwd_zero);
        const int hd2 = hd >> 1, a2 = A >> 1;
        {   // Q' image: 4 rows (64 B, chunk-swizzled: OperandRows<32>) per instruction; slot sl holds logical dword ld
            const int sl = lane & 15, rowin = lane >> 4;
            const int ld = 4 * ((sl >> 2) ^ ((0 - wave) & 3)) + (sl & 3);
            const int kind = ld < hd2 ? 0 : (ld < hd2 + a2 ? 1 : 2);
            const int row0 = 4 * wave + rowin;
            const long stride = kind == 0 ? 64L * hd : (kind == 1 ? 64L * A : 0L);
            const char* src = kind == 0 ? reinterpret_cast<const char*>(qb) + ((long)row0 * hd + 2 * ld) * 2
                            : kind == 1 ? reinterpret_cast<const char*>(qa) + ((long)row0 * A + 2 * (ld - hd2)) * 2 : zsrc;
            int gi = wave;
            for (; gi < Nqp / 4; gi += NW) { glds4(src, Qimg + gi * 256); src += stride; }
            for (; gi < nq / 4; gi += NW) glds4(zsrc, Qimg + gi * 256);                  // phantom rows
        }
        {   // dO image: 8 rows (32 B, halves swapped in rows 8..15 of every 16: oimg_off) per instruction
            const int sl = lane & 7, rowin = lane >> 3;
            const int ld = sl ^ ((wave & 1) << 2);
            const bool fo = ld < hd2;
            const long stride = fo ? 128L * C : 0L;                                      // bytes per 64 rows
            const char* src = fo ? reinterpret_cast<const char*>(dob) + ((long)(8 * wave + rowin) * C + 2 * ld) * 2 : zsrc;
            int gi = wave;
            for (; gi < Nqp / 8; gi += NW) { glds4(src, Oimg + gi * 256); src += stride; }
            for (; gi < nq / 8; gi += NW) glds4(zsrc, Oimg + gi * 256);
        }
        {   // K rows [Nkp][16] (head dims | zero): 8 rows per instruction; prompt rows from kp, padding rows zero
            const int sl = lane & 7, rowin = lane >> 3;
            const bool fk = sl < hd2;
            const long stride = fk ? 128L * hd : 0L;
            const char* src = fk ? reinterpret_cast<const char*>(kb) + ((long)(8 * wave + rowin) * hd + 2 * sl) * 2 : zsrc;
            int gi = wave;
            for (; gi < Nqp / 8; gi += NW) { glds4(src, Kt + gi * 256); src += stride; }
            if (fk) src = reinterpret_cast<const char*>(kpb) + ((long)(8 * gi + rowin - Nqp) * hd + 2 * sl) * 2;
            for (; gi < Nkp / 8; gi += NW) {
                const bool live = 8 * gi < Nqp + n_prompt_rows;
                glds4((fk && !live) ? zsrc : src, Kt + gi * 256);
                src += stride;
            }
        }
        // delta = sum_j dO * O per query row (negated): plain loads of the two head slices, four lanes per row
        {
            const int o_elems = nq * 4;
            for (int e0 = tid; e0 < o_elems; e0 += 3 * 64 * NW) {
                bf16x4 gv[3], ov[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) o_piece(qs, min(e0 + 64 * NW * u, o_elems - 1), gv[u], ov[u]);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int e = min(e0 + 64 * NW * u, o_elems - 1);
                    float part = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) part += (float)gv[u][i] * (float)ov[u][i];
                    part += __shfl_xor(part, 1);
                    part += __shfl_xor(part, 2);
                    if ((e & 3) == 0) del_s[e >> 2] = -part;
                }
            }
        }
    } else {
    // K^T (head dims x keys): four consecutive keys of one 4-channel group, transposed 4x4 in registers.  The image has hd
    // rows: lanes that would read rows hd..15 of the A operand re-read row 0 (row j of A only reaches row j of dQ^T, and
    // rows >= hd are never stored).  Loads here, stores after the query side's loads are out.
    const int kt_elems = (Nkp / 4) * hd4;
    auto kt_load = [&](int e, bf16x4 (&in)[4]) {
        const int c4 = e % hd4, k4 = e / hd4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * k4 + i;
            in[i] = keep_if(ld4(kb + kv_off(row, c4, to_kp)), row < Nqp + n_prompt_rows);
        }
    };
    auto kt_store = [&](int e, const bf16x4 (&in)[4]) {
        const int c4 = e % hd4, k4 = e / hd4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 outv;
            outv[0] = in[0][j]; outv[1] = in[1][j]; outv[2] = in[2][j]; outv[3] = in[3][j];
            *reinterpret_cast<bf16x4*>(Kt + (size_t)(4 * c4 + j) * KTROW + 2 * (4 * k4)) = outv;
        }
    };
    bf16x4 kt_in[4];
    kt_load(min(tid, kt_elems - 1), kt_in);
    stage_query_side<64 * NW>(qs, Qimg, Oimg, del_s, nq, tid);
    if (tid < kt_elems) kt_store(tid, kt_in);
    for (int e = tid + 64 * NW; e < kt_elems; e += 64 * NW) { kt_load(e, kt_in); kt_store(e, kt_in); }
    }
    if (tid < nq) {
        const bool ok = tid < Nqp;
        lse_s[tid] = ok ? -lse0 * MIVP_LOG2E : -INFINITY;
        ridq[tid] = (uint8_t)((ok && tid < d.Nq) ? rid0 : 255);
    }
    for (int m = tid + 64 * NW; m < nq; m += 64 * NW) {
        const bool ok = m < Nqp;
        lse_s[m] = ok ? -lseb[m] * MIVP_LOG2E : -INFINITY;
        ridq[m] = (uint8_t)((ok && m < d.Nq) ? (MASKED ? tok_rid[pw * Nqp + m] : 0) : 255);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs have landed; the barrier publishes them
    __syncthreads();
    // a shifted block's window that the volume boundary does not cut has ONE region id: its mask is a no-op
    bool cut = false;
    if (MASKED) {
        const uint8_t first = ridq[0];
        int differs = 0;
        for (int m = tid; m < d.Nq; m += 64 * NW) differs |= ridq[m] != first;
        cut = __syncthreads_or(differs) != 0;
    }

    const uint32_t dbase = DROP ? attn_row(bph, 0, Nqp, Nkp) : 0u;
    const char* ktrow = Kt + (size_t)(r < hd ? r : 0) * KTROW;  // this lane's row of the dQ product's A operand

    // dQ of query tile t: the eight waves' partial tiles (buffer t & 1) added in wave order
    auto dq_store = [&](int t) {
        // lane (r, g) adds chunk gs = g ^ f(r) of query row r: linear 64-byte rows read with ds_read_b128 in the natural lane
        // order take twice the passes (tools/ubench/lds_patterns: 0.87 against 0.75 of LDS-active cycles), the chunk rotation of
        // OperandRows<32> is the conflict-free b128 read order; which lane holds which four head dims is free here
        const int gs = g ^ ((0 - (r >> 2)) & 3);
        const float* src = dqp + (size_t)(t & 1) * NW * 256 + r * 16 + 4 * gs;
        f32x4 acc = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int w = 1; w < NW; ++w) acc = acc + *reinterpret_cast<const f32x4*>(src + w * 256);
        const int qrow = 16 * t + r;
        if (qrow < Nqp && 4 * gs < hd)
            st4(dq + ((bph * Nqp + qrow) * (long)hd + 4 * gs), pack4(acc * MIVP_LN2));     // K carries log2(e)
    };

    // lane-constant parts of every LDS address of the tile loop (the t-dependent part is a multiple of the tile stride:
    // the row swizzles of Qimg / Oimg repeat every 16 rows)
    const char* q_rd = Qimg + KR::off(r, 8 * g);                                   // + 1024 t : S operand row of this lane
    const char* q_tr = Qimg + KR::off(4 * g + (r >> 2), 4 * (r & 3));              // + 1024 t : Q'^T block for dK^T
    const char* o_rd = Oimg + oimg_off(r, 8 * g);                                  // +  512 t : dP operand row
    const char* o_tr = Oimg + oimg_off(4 * g + (r >> 2), 8 * (r & 3));             // +  512 t : dO^T block for dV^T
    char* ex_wr = exch + exch_off(wave, r, g);                                     // + 4096 i : this wave's slot, as written
    const char* ex_tr = exch + exch_off(wave, 4 * g + (r >> 2), r & 3);            // + 4096 i : ... as read back transposed
    // K^T columns of key tile i (+ 256 i classic; DMA: + 4096 i, a transposing read of the tile's rows 4g .. 4g+3)
    const char* kt_rd = DMA ? Kt + (16 * wave + 4 * g + (r >> 2)) * 32 + 8 * (r & 3) : ktrow + (16 * wave + 4 * g) * 2;
    float* dqp_wr = dqp + wave * 256 + r * 16 + 4 * g;
    static_assert(NW == 8, "slot strides below assume eight waves");

    // The wave that adds the partials of tile t rotates among the waves that own the FEWEST key tiles (waves nt % NW .. NW-1):
    // with 26 key tiles the two waves that own four would otherwise also take a turn and every barrier would wait for them
    const int light0 = nt % NW, nlight = NW - light0;
    auto reducer_of = [&](int t) { return light0 + t % nlight; };
    // The tile loop is instantiated per (number of key tiles this wave owns, shift mask in effect): its body is then free
    // of branches, so the scheduler interleaves the tiles' MFMA -> exp -> multiply -> convert chains (with a scalar
    // "do I own a fourth tile" test per tile every tile sat in its own basic block behind two s_nop 7).
    auto walk = [&](auto nt_c, auto cut_c) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr bool CUT = decltype(cut_c)::value;
        for (int t = 0; t < nqt; ++t) {
            if (ABL != 1 && t > 0 && wave == reducer_of(t - 1)) dq_store(t - 1);
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * t + 4 * g);
            const f32x4 n4 = *reinterpret_cast<const f32x4*>(del_s + 16 * t + 4 * g);
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(q_rd + 1024 * t);
            const bf16x4 of = *reinterpret_cast<const bf16x4*>(o_rd + 512 * t);
            // A operands of the dK^T / dV^T products: Q'^T and dO^T of this tile by transposing reads (rows 4g..4g+3)
            const bf16x4 qt = tr_read(q_tr + 1024 * t);
            const bf16x4 ot = tr_read(o_tr + 512 * t);
            uint32_t rqs[4] = {0u, 0u, 0u, 0u};
            if (CUT) {
                const uint32_t rq4 = *reinterpret_cast<const uint32_t*>(ridq + 16 * t + 4 * g);
                rqs[0] = rq4 & 0xFFu; rqs[1] = (rq4 >> 8) & 0xFFu; rqs[2] = (rq4 >> 16) & 0xFFu; rqs[3] = rq4 >> 24;
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const f32x4 s = mfma16(qf, kf[i], l4);
                const f32x4 dp = mfma16k16(of, vf[i], DROP ? fzero4() : n4);
                f32x4 pv, ds;
                const uint32_t kcls = (kk4 >> (8 * i)) & 0xFFu;
                const bool never = kcls == 255u;               // prompt / padding keys are never masked
                // Dropout hashes of elements (query 16t + 4g + j, key 16 kt + r), j = 0..3.  Keys 2m and 2m + 1 share a hash and sit on
                // ADJACENT lanes here, so the pair splits the four queries: the even lane hashes j = 0, 1, the odd lane j = 2, 3, and
                // quad-permute moves (one vector instruction each, no LDS) hand every hash to both -- 2 hashes + 4 moves per tile
                // instead of 4 hashes (8 instructions apiece)
                uint32_t hj[4] = {0u, 0u, 0u, 0u};
                if (DROP) {
                    const int krow = 16 * (wave + NW * i) + r;
                    const uint32_t row0 = dbase + (uint32_t)(16 * t + 4 * g + 2 * (r & 1)) * (uint32_t)(Nkp >> 1);
                    const uint32_t h0 = drop_hash(attn_pair(row0, krow), attn_key);
                    const uint32_t h1 = drop_hash(attn_pair(row0 + (uint32_t)(Nkp >> 1), krow), attn_key);
                    hj[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0xA0, 0xF, 0xF, false);      // quad_perm [0,0,2,2]: the even lane's
                    hj[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0xA0, 0xF, 0xF, false);
                    hj[2] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0xF5, 0xF, 0xF, false);      // quad_perm [1,1,3,3]: the odd lane's
                    hj[3] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0xF5, 0xF, 0xF, false);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float keep = 1.f;
                    if (DROP) keep = drop_keep(hj[j], r & 1, d.attn_drop_thr) ? d.attn_drop_scale : 0.f;
                    const float dpe = DROP ? dp[j] * keep + n4[j] : dp[j];
                    float pe, dsv;
                    if (ABL == 5) {
                        pe = s[j]; dsv = dpe;
                    } else if (CUT) {
                        // a masked logit is the constant 0 (accumulator value l4[j]): it keeps its P, carries no gradient
                        const bool live = never | (rqs[j] == kcls);
                        pe = __builtin_amdgcn_exp2f(live ? s[j] : l4[j]);
                        dsv = live ? pe * dpe : 0.f;
                    } else {
                        pe = __builtin_amdgcn_exp2f(s[j]);
                        dsv = pe * dpe;
                    }
                    pv[j] = DROP ? pe * keep : pe;
                    ds[j] = dsv;
                }
                if (i == NT - 1) dsum = dsum + (f32x2{ds[0], ds[1]} + f32x2{ds[2], ds[3]});   // (used if that tile holds prompt keys)
                const u32x2 pb = ABL == 5 ? u32x2{__builtin_bit_cast(unsigned, pv[0]), __builtin_bit_cast(unsigned, pv[1])} : pack4_pk(pv);
                const u32x2 sb = ABL == 5 ? u32x2{__builtin_bit_cast(unsigned, ds[0]), __builtin_bit_cast(unsigned, ds[1])} : pack4_pk(ds);
                if (ABL != 3) *reinterpret_cast<u32x2*>(ex_wr + 4096 * i) = sb;            // this wave's private slot
                dkacc[i] = mfma16k16(qt, sb, dkacc[i]);
                dvacc[i] = mfma16k16(ot, pb, dvacc[i]);
            }
            // partial dQ^T [head dim][query] over this wave's keys: B = dS^T by transposing reads of the slots just written
            // (same wave: ordered by the LDS queue), A = K^T columns of the key tile
            // Key tiles go in PAIRS through one K = 32 product: a 16x16x16 MFMA costs the matrix pipe and the vector port exactly
            // what a 16x16x32 one does (tools/ubench/valu_rates.hip), and the chain of dependent accumulations in front of the
            // barrier is half as long
            f32x4 dqa = fzero4();
            constexpr int NTQ = ABL == 3 ? 0 : NT;
#pragma unroll
            for (int i = 0; i + 1 < NTQ; i += 2) {
                const bf16x8 b = cat44(tr_read(ex_tr + 4096 * i), tr_read(ex_tr + 4096 * (i + 1)));
                const bf16x8 a = DMA ? cat44(tr_read(kt_rd + 4096 * i), tr_read(kt_rd + 4096 * (i + 1)))
                                     : cat44(*reinterpret_cast<const bf16x4*>(kt_rd + 256 * i), *reinterpret_cast<const bf16x4*>(kt_rd + 256 * (i + 1)));
                dqa = mfma16(a, b, dqa);
            }
            if (NTQ & 1) {
                constexpr int i = NTQ > 0 ? NTQ - 1 : 0;
                const bf16x4 b = tr_read(ex_tr + 4096 * i);
                const bf16x4 a = DMA ? tr_read(kt_rd + 4096 * i) : *reinterpret_cast<const bf16x4*>(kt_rd + 256 * i);
                dqa = mfma16k16(a, b, dqa);
            }
            *reinterpret_cast<f32x4*>(dqp_wr + (t & 1) * NW * 256) = dqa;
            if (ABL != 2) __syncthreads();
        }
        if (ABL != 1 && wave == reducer_of(nqt - 1)) dq_store(nqt - 1);      // (a phantom tile stores nothing)
    };
    int my_nt = wave < nt ? (nt - wave + NW - 1) / NW : 0;         // scalar
    if (ABL == 4 && my_nt > 3) my_nt = 3;
    auto walk_nt = [&](auto cut_c) {
        switch (my_nt) {
            case 0: walk(std::integral_constant<int, 0>{}, cut_c); break;
            case 1: walk(std::integral_constant<int, 1>{}, cut_c); break;
            case 2: walk(std::integral_constant<int, 2>{}, cut_c); break;
            case 3: walk(std::integral_constant<int, 3>{}, cut_c); break;
            default: walk(std::integral_constant<int, 4>{}, cut_c); break;
        }
    };
    if (MASKED && cut) walk_nt(std::true_type{});
    else walk_nt(std::false_type{});

    // ---------------- results of this wave's key tiles ----------------
#pragma unroll
    for (int i = 0; i < FUSED_KPW; ++i) {
        const int kt = wave + NW * i;
        if (kt >= nt) continue;
        const int krow = kt * 16 + r;
        if (krow < Nqp) {
            if (4 * g < hd) {
                st4(dk + ((bph * Nqp + krow) * (long)hd + 4 * g), pack4(dkacc[i]));
                st4(dv + ((bph * Nqp + krow) * (long)hd + 4 * g), pack4(dvacc[i]));
            }
        } else if (krow < Nqp + d.Npp) {
            const int tp = krow - Nqp;
            const float dt = col_sum(dsum[0] + dsum[1]);               // (i is this wave's last tile: see dsum)
            if (4 * g < hd) {
                *reinterpret_cast<f32x4*>(dkp_part + ((bph * d.Npp + tp) * (long)hd + 4 * g)) = dkacc[i];
                *reinterpret_cast<f32x4*>(dvp_part + ((bph * d.Npp + tp) * (long)hd + 4 * g)) = dvacc[i];
            }
            if (g == 0) dtok_part[bph * d.Npp + tp] = dt;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Prompt-only form for the FIRST prompted block behind a frozen stem (no data gradient is needed there: only the prompt
// keys' dKp / dVp partials and the prompt-bias column sums).  KP = Npp / 16 prompt key tiles (1, 2, 4 or 8): wave w owns
// prompt tile w % KP and the query tiles of part w / KP (8 / KP parts); no dS exchange, no dQ, no barrier in the loop.  The
// parts' accumulators meet in LDS at the end (fixed order).  Replaces mivp_win_attn_delta + the prompt-only mode of
// mivp_win_attn_bwd_dkv (which staged every query for four key tiles: 141 us -> see profiles/).
// ---------------------------------------------------------------------------------------------
template <int NW, bool DROP>
__global__ __launch_bounds__(64 * NW, DROP ? 4 : 8) void k_win_attn_bwd_prompt(
    MivpSwinDesc d, const bf16_t* __restrict__ q, const bf16_t* __restrict__ kp, const bf16_t* __restrict__ vp,
    const bf16_t* __restrict__ qa, const bf16_t* __restrict__ ka, const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o,
    const float* __restrict__ lse, float* __restrict__ dkp_part, float* __restrict__ dvp_part, float* __restrict__ dtok_part) {
    // dropout keys of this call (common.hpp drop_seed: uniform, scalar-ALU work; unused without dropout)
    const uint32_t attn_key = drop_seed(d.attn_seed, d.seed_epoch), proj_key = drop_seed(d.proj_seed, d.seed_epoch);
    (void)attn_key; (void)proj_key;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, heads = d.heads, hd = C / heads, Nqp = d.Nqp, Nkp = d.Nkp, A = d.augp;
    const int hd4 = hd / 4, a4 = A / 4;
    const long bph = blockIdx.x;
    const int head = (int)(bph % heads);
    const long bp = bph / heads;
    const int nqt = Nqp / 16;
    const int nq = nqt * 16;
    const int KP = d.Npp / 16, parts = NW / KP;
    char* Qimg = smem;
    char* Oimg = Qimg + (size_t)nq * 64;
    float* lse_s = reinterpret_cast<float*>(Oimg + (size_t)nq * 32);
    float* del_s = lse_s + nq;
    // [NW][2][256] + [NW][64] partial accumulators: over the Q' image once the tile loop is done (a workgroup lives ~10 us
    // behind its staging round trips: four workgroups per CU instead of two hide them)
    float* red = reinterpret_cast<float*>(smem);
    const bf16_t* qb = q + bph * (long)Nqp * hd;
    const bf16_t* kpb = kp + (long)head * d.Npp * hd;
    const bf16_t* vpb = vp + (long)head * d.Npp * hd;
    const bf16_t* kab = ka + (long)head * Nkp * A;
    const bf16_t* dob = d_o + bp * (long)Nqp * C + head * hd;
    const bf16_t* ob = o + bp * (long)Nqp * C + head * hd;
    const float* lseb = lse + bph * (long)Nqp;
    // this wave's prompt key tile (loads first, used after the staging)
    const int pt = wave % KP, part = wave / KP;
    const int trow = pt * 16 + r;                             // row of Kp / Vp
    bf16x4 piece[2];
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
        const int c4 = 2 * g + hlf;
        const int ca = min(max(c4 - hd4, 0), a4 - 1);
        const long to_ka = kab - kpb;
        const long off = sel(c4 < hd4, (long)(trow * hd + 4 * min(c4, hd4 - 1)), to_ka + (long)((Nqp + trow) * A + 4 * ca));
        piece[hlf] = keep_if(ld4(kpb + off), c4 < hd4 + a4);
    }
    const bf16x8 kf = cat44(piece[0], piece[1]);
    const bf16x4 vf = keep_if(ld4(vpb + ((uint32_t)trow * hd + 4 * min(g, hd4 - 1))), g < hd4);
    const int m0 = min(tid, nq - 1);
    const float lse0 = lseb[m0];
    const QSide qs{qb, qa, dob, ob, Nqp, hd, hd4, A, a4, C};
    stage_query_side<64 * NW>(qs, Qimg, Oimg, del_s, nq, tid);
    if (tid < nq) lse_s[tid] = -lse0 * MIVP_LOG2E;
    for (int m = tid + 64 * NW; m < nq; m += 64 * NW) lse_s[m] = -lseb[m] * MIVP_LOG2E;
    f32x4 dkacc = fzero4(), dvacc = fzero4();
    f32x2 dsum = {0.f, 0.f};
    __syncthreads();
    const uint32_t dbase = DROP ? attn_row(bph, 0, Nqp, Nkp) : 0u;
    const char* q_rd = Qimg + KR::off(r, 8 * g);
    const char* q_tr = Qimg + KR::off(4 * g + (r >> 2), 4 * (r & 3));
    const char* o_rd = Oimg + oimg_off(r, 8 * g);
    const char* o_tr = Oimg + oimg_off(4 * g + (r >> 2), 8 * (r & 3));
    const int t_begin = part * nqt / parts, t_end = (part + 1) * nqt / parts;
    for (int t = t_begin; t < t_end; ++t) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * t + 4 * g);
        const f32x4 n4 = *reinterpret_cast<const f32x4*>(del_s + 16 * t + 4 * g);
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(q_rd + 1024 * t);
        const bf16x4 of = *reinterpret_cast<const bf16x4*>(o_rd + 512 * t);
        const bf16x4 qt = tr_read(q_tr + 1024 * t);
        const bf16x4 ot = tr_read(o_tr + 512 * t);
        const f32x4 s = mfma16(qf, kf, l4);                   // prompt keys are never masked
        const f32x4 dp = mfma16k16(of, vf, DROP ? fzero4() : n4);
        f32x4 pv, ds;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float keep = 1.f;
            if (DROP) {
                const int krow = Nqp + trow;
                const uint32_t hsh = drop_hash(attn_pair(dbase + (uint32_t)(16 * t + 4 * g + j) * (uint32_t)(Nkp >> 1), krow), attn_key);
                keep = drop_keep(hsh, krow & 1, d.attn_drop_thr) ? d.attn_drop_scale : 0.f;
            }
            const float dpe = DROP ? dp[j] * keep + n4[j] : dp[j];
            const float pe = __builtin_amdgcn_exp2f(s[j]);
            pv[j] = DROP ? pe * keep : pe;
            ds[j] = pe * dpe;
        }
        dsum = dsum + (f32x2{ds[0], ds[1]} + f32x2{ds[2], ds[3]});
        dkacc = mfma16k16(qt, pack4_pk(ds), dkacc);
        dvacc = mfma16k16(ot, pack4_pk(pv), dvacc);
    }
    // add the parts (fixed order), store this (window, head)'s partials
    __syncthreads();                                          // every wave is done reading Qimg / Oimg
    float* mine = red + (size_t)wave * 576;
    *reinterpret_cast<f32x4*>(mine + r * 16 + 4 * g) = dkacc;
    *reinterpret_cast<f32x4*>(mine + 256 + r * 16 + 4 * g) = dvacc;
    mine[512 + lane] = dsum[0] + dsum[1];
    __syncthreads();
    if (part == 0) {
        f32x4 ak = fzero4(), av = fzero4();
        float at = 0.f;
        for (int pp = 0; pp < parts; ++pp) {
            const float* src = red + (size_t)(pt + KP * pp) * 576;
            ak = ak + *reinterpret_cast<const f32x4*>(src + r * 16 + 4 * g);
            av = av + *reinterpret_cast<const f32x4*>(src + 256 + r * 16 + 4 * g);
            at += src[512 + lane];
        }
        at = col_sum(at);
        if (4 * g < hd) {
            *reinterpret_cast<f32x4*>(dkp_part + ((bph * d.Npp + trow) * (long)hd + 4 * g)) = ak;
            *reinterpret_cast<f32x4*>(dvp_part + ((bph * d.Npp + trow) * (long)hd + 4 * g)) = av;
        }
        if (g == 0) dtok_part[bph * d.Npp + trow] = at;
    }
}

static size_t prompt_lds_bytes(const MivpSwinDesc* d) {
    const size_t images = (size_t)d->Nqp * 96, red = 8 * 576 * 4;                  // the partials reuse the images' space
    return (images > red ? images : red) + 2 * (size_t)d->Nqp * 4;
}

extern "C" int mivp_win_attn_bwd_prompt_supported(const MivpSwinDesc* d) {
    if (!d || d->heads <= 0 || d->C % d->heads || d->Np <= 0) return 0;
    const int hd = d->C / d->heads, kp = d->Npp / 16;
    int dks, nt;
    if (mivp_attn_tile_config(d, &dks, &nt) || dks != 1) return 0;
    if (hd > 16 || hd % 4 || d->augp < 4 || !(kp == 1 || kp == 2 || kp == 4 || kp == 8) || d->Nqp % 16) return 0;
    const size_t lds = prompt_lds_bytes(d);
    return lds <= 78 * 1024 ? 1 : 0;
}

extern "C" int mivp_win_attn_bwd_prompt(const MivpSwinDesc* d, const void* q, const void* kp, const void* vp, const void* qa,
                                        const void* ka, const void* o, const void* d_o, const float* lse, float* dkp_part,
                                        float* dvp_part, float* dtok_part, mivp_stream_t stream) {
    MIVP_REQUIRE(d && q && kp && vp && qa && ka && o && d_o && lse && dkp_part && dvp_part && dtok_part);
    if (!mivp_win_attn_bwd_prompt_supported(d)) { mivp_set_error("win_attn_bwd_prompt: shape outside the kernel's range"); return MIVP_EUNSUPPORTED; }
    const size_t lds = prompt_lds_bytes(d);
    constexpr int NW = 8;
    auto kern = d->attn_drop_thr ? k_win_attn_bwd_prompt<NW, true> : k_win_attn_bwd_prompt<NW, false>;
    MIVP_LDS_OPT_IN(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)d->B * d->P * d->heads)), dim3(64 * NW), lds, (hipStream_t)stream, *d,
                       (const bf16_t*)q, (const bf16_t*)kp, (const bf16_t*)vp, (const bf16_t*)qa, (const bf16_t*)ka,
                       (const bf16_t*)o, (const bf16_t*)d_o, lse, dkp_part, dvp_part, dtok_part);
    return mivp_check_launch("win_attn_bwd_prompt");
}

static size_t fused_lds_bytes(const MivpSwinDesc* d, bool dma) {
    const size_t nq = (size_t)((d->Nqp + 31) / 32 * 2) * 16, nt = d->Nkp / 16;
    const size_t hd = d->C / d->heads;
    const size_t kimg = dma ? (size_t)d->Nkp * 32 : hd * (size_t)(d->Nkp + 8) * 2;
    return nq * 64 + nq * 32 + kimg + nt * 512 + 2 * 8 * 256 * sizeof(float) + 2 * nq * sizeof(float) + ((nq + 15) & ~(size_t)15);
}
// LDS-DMA staged images unless MIVP_ATTN_BWD_REG_STAGING is set (A/B runs) or they would cost the second workgroup per CU
static bool fused_use_dma(const MivpSwinDesc* d) {
    static const bool reg_staging = getenv("MIVP_ATTN_BWD_REG_STAGING") != nullptr;
    if (reg_staging || d->attn_drop_thr) return false;
    const size_t a = fused_lds_bytes(d, true), b = fused_lds_bytes(d, false);
    return a <= 80 * 1024 || b > 80 * 1024;
}

/* 1 when mivp_win_attn_bwd_fused covers this shape (head_dim <= 16, head_dim + bias columns <= 32, <= 512 keys, LDS fits) */
extern "C" int mivp_win_attn_bwd_fused_supported(const MivpSwinDesc* d) {
    if (!d || d->heads <= 0 || d->C % d->heads) return 0;
    const int hd = d->C / d->heads;
    int dks, nt;
    if (mivp_attn_tile_config(d, &dks, &nt) || dks != 1) return 0;
    if (hd > 16 || hd % 4 || d->augp < 4 || d->Nq < 1 || d->Nkp / 16 > 8 * FUSED_KPW) return 0;
    return fused_lds_bytes(d, fused_use_dma(d)) <= 160 * 1024 ? 1 : 0;
}

extern "C" int mivp_win_attn_bwd_fused(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp,
                                       const void* vp, const void* qa, const void* ka, const int32_t* tok_rid,
                                       const void* o, const void* d_o, const float* lse, void* dq, void* dk, void* dv,
                                       float* dkp_part, float* dvp_part, float* dtok_part, mivp_stream_t stream) {
    MIVP_REQUIRE(d != nullptr);
    MIVP_REQUIRE(d->B > 0 && d->C > 0 && d->heads > 0 && d->P > 0);
    MIVP_REQUIRE(d->C % 8 == 0 && d->C % d->heads == 0 && (d->C / d->heads) % 4 == 0);
    MIVP_REQUIRE(d->Nqp % 16 == 0 && d->Npp % 16 == 0 && d->Nkp % 32 == 0 && d->augp % 4 == 0);
    MIVP_REQUIRE(d->Nkp >= d->Nqp + d->Npp && d->Nqp >= d->Nq);
    MIVP_REQUIRE((long)d->B * d->P * d->Nqp < (1L << 31));
    MIVP_REQUIRE(q && k && v && qa && ka && o && d_o && lse && dq && dk && dv);
    MIVP_REQUIRE(d->Np == 0 || (kp && vp && dkp_part && dvp_part && dtok_part));
    MIVP_REQUIRE(!d->has_mask || tok_rid);
    if (!mivp_win_attn_bwd_fused_supported(d)) { mivp_set_error("win_attn_bwd_fused: shape outside the fused kernel's range"); return MIVP_EUNSUPPORTED; }
    const bool dma = fused_use_dma(d);
    const size_t lds = fused_lds_bytes(d, dma);
    constexpr int NW = 8;
    static const int abl = getenv("MIVP_ATTN_BWD_ABL") ? atoi(getenv("MIVP_ATTN_BWD_ABL")) : 0;
    const bool msk = d->has_mask != 0;
    auto kern = d->attn_drop_thr ? (msk ? k_win_attn_bwd_fused<NW, true, true> : k_win_attn_bwd_fused<NW, true, false>)
              : (dma && abl == 1) ? (msk ? k_win_attn_bwd_fused<NW, false, true, true, 1> : k_win_attn_bwd_fused<NW, false, false, true, 1>)
              : (dma && abl == 2) ? (msk ? k_win_attn_bwd_fused<NW, false, true, true, 2> : k_win_attn_bwd_fused<NW, false, false, true, 2>)
              : (dma && abl == 3) ? (msk ? k_win_attn_bwd_fused<NW, false, true, true, 3> : k_win_attn_bwd_fused<NW, false, false, true, 3>)
              : (dma && abl == 4) ? (msk ? k_win_attn_bwd_fused<NW, false, true, true, 4> : k_win_attn_bwd_fused<NW, false, false, true, 4>)
              : (dma && abl == 5) ? (msk ? k_win_attn_bwd_fused<NW, false, true, true, 5> : k_win_attn_bwd_fused<NW, false, false, true, 5>)
              : dma ? (msk ? k_win_attn_bwd_fused<NW, false, true, true> : k_win_attn_bwd_fused<NW, false, false, true>)
                    : (msk ? k_win_attn_bwd_fused<NW, false, true> : k_win_attn_bwd_fused<NW, false, false>);
    MIVP_LDS_OPT_IN(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)d->B * d->P * d->heads)), dim3(64 * NW), lds, (hipStream_t)stream, *d,
                       (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)kp, (const bf16_t*)vp,
                       (const bf16_t*)qa, (const bf16_t*)ka, tok_rid, (const bf16_t*)o, (const bf16_t*)d_o, lse, (bf16_t*)dq,
                       (bf16_t*)dk, (bf16_t*)dv, dkp_part, dvp_part, dtok_part);
    return mivp_check_launch("win_attn_bwd_fused");
}
