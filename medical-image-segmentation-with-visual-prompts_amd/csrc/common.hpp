// Shared device helpers for the gfx950 (CDNA4) kernels.  Wave = 64 lanes.
//
// MFMA convention used everywhere ("swapped" orientation):
//   acc = mfma_f32_16x16x32_bf16(A, B, acc)   with  D[row][col] += sum_k A[row][k] * B[k][col]
//   lane l: r = l & 15, g = l >> 4
//     A operand: 8 bf16 = A[row r][k = 8g .. 8g+7]
//     B operand: 8 bf16 = B[k = 8g .. 8g+7][col r]
//     D        : 4 f32  = D[row 4g + j][col r], j = 0..3
//   We always put the WEIGHT tile on A (rows = output channels) and the TOKEN tile on B
//   (cols = tokens), so that after the MFMA every lane owns ONE token (r) and four
//   consecutive output channels (4g..4g+3): epilogues are 8-byte stores and per-token
//   reductions (LayerNorm, softmax) need only two cross-lane steps (xor 16, xor 32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/mivp.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MIVP_DEV __device__ __forceinline__

MIVP_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
MIVP_DEV bf16x8 zero8() { bf16x8 z; for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.0f; return z; }
MIVP_DEV bf16x4 zero4() { bf16x4 z; for (int i = 0; i < 4; ++i) z[i] = (bf16_t)0.0f; return z; }
MIVP_DEV f32x4 fzero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

MIVP_DEV bf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
MIVP_DEV bf16x4 ld4(const bf16_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
MIVP_DEV void st8(bf16_t* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
MIVP_DEV void st4(bf16_t* p, bf16x4 v) { *reinterpret_cast<bf16x4*>(p) = v; }

MIVP_DEV bf16x4 pack4(f32x4 v) {
    bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3]; return o;
}
MIVP_DEV bf16x8 cat44(bf16x4 lo, bf16x4 hi) {
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}
// sum / max over the 4 lanes that share a token column (lanes r, r+16, r+32, r+48)
MIVP_DEV float col_sum(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }
MIVP_DEV float col_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32)); return v; }
// IEEE-754-2019 maximum (gfx950: v_maximum3_f32): unlike fmaxf() it needs no canonicalisation of its operands -- fmaxf()
// on an MFMA result or a cross-lane read costs an extra `v_max x, x` per operand because the compiler cannot prove the
// value is not a signalling NaN; in the VALU-bound softmax loops that is one wasted op per logit.  max3_raw folds two
// inputs per instruction.
MIVP_DEV float max2_raw(float a, float b) { return __builtin_elementwise_maximum(a, b); }
MIVP_DEV float max3_raw(float a, float b, float c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }

// Attention logit classes (branch-free: these compile to v_cmp + v_cndmask, never to exec-mask branches).
//   key class  -1: excluded from the softmax (logit -> -inf); padding keys no longer use it (MIVP_PAD_KEY_BIAS below)
//              -2: prompt key or padding key: never masked
//            >= 0: region id; the logit survives only where it equals the query's region id, otherwise
//                  the reference's multiplicative mask forces it to 0 (window_attention.py:54-56)
// `live` is false where the logit is a constant (masked or excluded), i.e. where it carries no gradient.
// Logits live in log2 units: K' (head dims, prompt keys and the bias columns) is stored multiplied by log2(e), so
// S = K'Q'^T feeds v_exp_f32 directly, and the MFMA accumulator starts at `zero` = -(reference point) (the running max
// in the forward, the log-sum-exp in the backward): a logit forced to 0 by the mask is therefore the value `zero`.
template <int PAD, int PROMPT>
MIVP_DEV float classify_logit_c(float s, int kcls, int rq, bool& live, float zero) {
    const bool excl = kcls == PAD;
    const bool match = (kcls == PROMPT) | (kcls == rq);
    live = match & !excl;
    const float v = match ? s : zero;
    return excl ? -INFINITY : v;
}
MIVP_DEV float classify_logit(float s, int kcls, int rq, bool& live, float zero) {
    return classify_logit_c<-1, -2>(s, kcls, rq, live, zero);
}
constexpr float MIVP_LOG2E = 1.4426950408889634f;
// Padding keys (rows Nq..Nqp-1 and the unused prompt rows of K') carry this value in the i0 one-hot columns of the bias
// augmentation: every valid query then sees the logit -3e4 (x log2 e) there, P underflows to exactly 0 and no kernel
// spends VALU work on excluding them.  Their key class is the prompt class (always "matches"), so the shift mask --
// which forces a logit to 0, not to -inf -- never resurrects them.
constexpr float MIVP_PAD_KEY_BIAS = -30000.0f;
constexpr float MIVP_LN2 = 0.6931471805599453f;

// Counter-based dropout (no RNG state): one 32-bit hash serves the two elements of an index pair, 16 bits each.
// The attention kernels are bound by vector issue, so every operation here is paid for, and a 32-bit integer multiply
// (v_mul_lo_u32) is a QUARTER-rate instruction: round 2's murmur-style hash (three of them per hash) made the dropout
// forward 2.3x the dropout-free one.  This hash uses full-rate operations only: the pair counter plus a per-call key, two
// 24-bit multiplies (v_mul_u32_u24) with xor-shifts that carry the high bits back down, the key folded in a second time.
// The key is the fmix32 of (seed, epoch) -- uniform per launch, i.e. scalar-ALU work -- so masks of different seeds are not
// index-shifted copies of one another.  Measured on 4M consecutive counters (tools/dbg, numpy model): drop rates within 3e-4
// of p for both halves, |correlation| < 1.2e-3 at lags 1..1000 and between the halves, agreement between two seeds / epochs
// = keep^2 + drop^2 to 2e-4.
MIVP_DEV uint32_t drop_hash(uint32_t pair_idx, uint32_t key) {
    uint32_t t = pair_idx + key;
    t ^= t >> 16;
    t = __umul24(t, 0xB5AD4Fu);
    t ^= t >> 13;
    return __umul24(t, 0x6C62B9u) + (t >> 8) + ((key << 13) | (key >> 19));
}
// Key of a call's dropout hash: fmix32 of the descriptor's seed, advanced by the device-resident epoch word when the
// descriptor names one (MivpSwinDesc.seed_epoch, ABI 12).  A recorded HIP graph freezes the descriptor -- a kernel ARGUMENT --
// so a replay could only repeat one mask; the epoch word is device MEMORY that the graph itself increments at its start, and
// the forward and backward kernels of one step read the same value.  NULL (eager calls): the seed as given.  (Uniform
// operands: the compiler keeps this on the scalar ALU.)
MIVP_DEV uint32_t drop_seed(uint32_t seed, const uint32_t* __restrict__ epoch) {
    uint32_t s = epoch ? seed + epoch[0] * 0x9E3779B1u : seed;
    s ^= s >> 16; s *= 0x85EBCA6Bu;
    s ^= s >> 13; s *= 0xC2B2AE35u;
    return s ^ (s >> 16);
}
MIVP_DEV bool drop_keep(uint32_t h, int odd, uint32_t thr) { return ((odd ? (h >> 16) : (h & 0xFFFFu)) >= thr); }
// pair index of attention element (window-head bph, query q, key k): keys k and k^1 share a hash.  All in 32-bit
// arithmetic (wrap-around only re-uses counters between far-apart windows): attn_row() once per query row, then one add.
MIVP_DEV uint32_t attn_row(long bph, int q, int Nqp, int Nkp) {
    return ((uint32_t)bph * (uint32_t)Nqp + (uint32_t)q) * (uint32_t)(Nkp >> 1);
}
MIVP_DEV uint32_t attn_pair(uint32_t row, int k) { return row + (uint32_t)(k >> 1); }

// ---------------------------------------------------------------------------------------------
// Branch-free loads.  The token kernels run few waves per CU at the deep stages, so their run time is the NUMBER OF
// DEPENDENT global-load round trips (~1 us each), not bytes or flops.  A load under `if (in range)` compiles to its own
// basic block with an s_waitcnt vmcnt(0) right behind it -- one round trip per load (tools/isa_waits.py counts them).
// The idiom used instead: clamp the address into the allocation, load unconditionally, AND the value with a mask.
//   sel()     : the arms are by-value parameters, so clang emits a select (v_cndmask), never control flow
//   keep_if() : value or zero as a bitwise AND (a `cond ? v : 0` select is turned back into a branch around the load)
//   FastDiv   : n / d for n, d < 2^16 as one v_mul_hi (the per-piece `/ C`, `/ head_dim` were ~35 VALU instructions each)
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
MIVP_DEV int sel(bool c, int a, int b) { return c ? a : b; }
MIVP_DEV long sel(bool c, long a, long b) { return c ? a : b; }
MIVP_DEV float sel(bool c, float a, float b) { return c ? a : b; }
MIVP_DEV bf16x4 keep_if(bf16x4 v, bool ok) {
    u32x2 u = __builtin_bit_cast(u32x2, v);
    const unsigned m = 0u - (unsigned)ok;
    u[0] &= m; u[1] &= m;
    return __builtin_bit_cast(bf16x4, u);
}
MIVP_DEV bf16x8 keep_if(bf16x8 v, bool ok) {
    u32x4 u = __builtin_bit_cast(u32x4, v);
    const unsigned m = 0u - (unsigned)ok;
    u[0] &= m; u[1] &= m; u[2] &= m; u[3] &= m;
    return __builtin_bit_cast(bf16x8, u);
}
struct FastDiv {
    uint32_t m, d;
    MIVP_DEV explicit FastDiv(int div) : m(0xFFFFFFFFu / (uint32_t)div + 1u), d((uint32_t)div) {}      // (d == 1: m wraps to 0)
    MIVP_DEV int div(int n) const { return d == 1u ? n : (int)__umulhi((uint32_t)n, m); }   // exact for 0 <= n < 2^16, d < 2^16
    MIVP_DEV int mod(int n) const { return n - div(n) * (int)d; }
};

// LDS image of 16-byte-chunked operand rows read as MFMA fragments (lane = (row r, chunk g), ds_read_b128).
// DK == 32: rows are exactly 64 bytes and the chunk index is XOR-swizzled with {0,3,2,1}[(row >> 2) & 3], which
// makes every 16-lane service group of ds_read_b128 cover all sixteen 16-byte slots of the 256-byte bank row
// (no bank conflicts, no padding; same swizzle as the conv kernel).  Wider rows keep a +16-byte pad.
template <int DK>
struct OperandRows {
    static constexpr int ROW = DK == 32 ? 64 : (DK + 8) * 2;                       // bytes per row
    // byte offset of element `elem` (a multiple of 4) of row `row`
    static MIVP_DEV int off(int row, int elem) {
        if (DK == 32) return row * 64 + 16 * ((elem >> 3) ^ ((0 - (row >> 2)) & 3)) + 2 * (elem & 7);
        return row * ROW + 2 * elem;
    }
};

// Weight fragment images (host: pack_weight_frags / mivp.h): [row tile nt][k-step s][lane][8] bf16 -- the A fragment of
// (nt, s) is 1 KB contiguous, lane l's 16 bytes at 16 l.
MIVP_DEV bf16x8 wfrag(const bf16_t* __restrict__ wf, int ks, int nt, int s, int lane) {
    return ld8(wf + ((long)(nt * ks + s) * 64 + lane) * 8);
}
// Workgroup-shared weight slabs for the token GEMM kernels of the wide stages (each wave used to pull the whole matrix
// through L1 for its 16 tokens).  A slab is one row tile of a fragment image: KSL KB, copied as is by the 256 threads
// (linear, fully coalesced; the LDS reads are lane-linear too: no bank conflicts); two slabs are in flight (global ->
// registers one slab ahead, the other buffer was last read one iteration earlier, one barrier per slab).
template <int KSL>
struct WeightSlabs {
    static constexpr int PCS = (64 * KSL + 255) / 256;            // 16-byte pieces per thread
    static constexpr int BYTES = 2 * KSL * 1024;
    bf16x8 reg[PCS];
    MIVP_DEV void fetch(const bf16_t* __restrict__ wf, int nt) {   // (the ragged last piece repeats the final one)
#pragma unroll
        for (int u = 0; u < PCS; ++u) reg[u] = ld8(wf + (long)nt * KSL * 512 + 8 * min((int)threadIdx.x + 256 * u, 64 * KSL - 1));
    }
    MIVP_DEV void store(char* smem, int buf) const {
#pragma unroll
        for (int u = 0; u < PCS; ++u)
            *reinterpret_cast<bf16x8*>(smem + buf * KSL * 1024 + 16 * min((int)threadIdx.x + 256 * u, 64 * KSL - 1)) = reg[u];
    }
    static MIVP_DEV bf16x8 frag(const char* smem, int buf, int s, int lane) {
        return *reinterpret_cast<const bf16x8*>(smem + buf * KSL * 1024 + s * 1024 + 16 * lane);
    }
};

// The same for a ROW-MAJOR matrix (patch merging / expanding weights, merge_up.hip): a slab is 16 consecutive rows x 32*KSL
// columns staged as KSL sub-tiles of [16 rows][64 B] in the swizzled operand layout (OperandRows<32>).
template <int KSL>
struct WeightSlabsRM {
    static constexpr int PCS = (64 * KSL + 255) / 256;            // 16-byte pieces per thread
    static constexpr int BYTES = 2 * KSL * 1024;
    bf16x8 reg[PCS];
    // rows row0 .. row0+15 of w (leading dimension ld), zero beyond nrows / ncols
    MIVP_DEV void fetch(const bf16_t* __restrict__ w, int ld, int row0, int nrows, int ncols) {
#pragma unroll
        for (int u = 0; u < PCS; ++u) {
            const int p = threadIdx.x + 256 * u;
            const int sub = p >> 6, row = row0 + ((p & 63) >> 2), col = 32 * sub + 8 * (p & 3);
            reg[u] = (p < 64 * KSL && row < nrows && col < ncols) ? ld8(w + (long)row * ld + col) : zero8();
        }
    }
    MIVP_DEV void store(char* smem, int buf) const {
#pragma unroll
        for (int u = 0; u < PCS; ++u) {
            const int p = threadIdx.x + 256 * u;
            if (p < 64 * KSL)
                *reinterpret_cast<bf16x8*>(smem + buf * KSL * 1024 + (p >> 6) * 1024 + OperandRows<32>::off((p & 63) >> 2, 8 * (p & 3))) = reg[u];
        }
    }
    // fragment pieces of row r: 8 elements at column 32*sub + elem (elem multiple of 8) or 4 elements (elem multiple of 4)
    static MIVP_DEV bf16x8 frag8(const char* smem, int buf, int sub, int r, int elem) {
        return *reinterpret_cast<const bf16x8*>(smem + buf * KSL * 1024 + sub * 1024 + OperandRows<32>::off(r, elem));
    }
    static MIVP_DEV bf16x4 frag4(const char* smem, int buf, int sub, int r, int elem) {
        return *reinterpret_cast<const bf16x4*>(smem + buf * KSL * 1024 + sub * 1024 + OperandRows<32>::off(r, elem));
    }
};

// Deterministic block reduction of per-thread channel-group partials: thread t (global id gt) owns channel group
// gt % G; its 16 partial sums (8 channels x {s1, s2}) go to LDS and one thread per output channel adds the owners
// of that group in a fixed order.  (LDS float atomics would be shorter but their order changes run to run, and the
// BatchNorm statistics feed everything downstream.)  smem: 256 * 16 floats.
MIVP_DEV void block_reduce_groups(float* lds, const float (&s1)[8], const float (&s2)[8], int G, int C, long block_first_gtid,
                                  float* __restrict__ out2c) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) { lds[tid * 16 + i] = s1[i]; lds[tid * 16 + 8 + i] = s2[i]; }
    __syncthreads();
    for (int o = tid; o < 2 * C; o += 256) {
        const int which = o / C, c = o - which * C, cg = c >> 3, i = c & 7;
        int t0 = (int)(((long)cg - block_first_gtid % G + G) % G);      // first thread of this block that owns group cg
        float acc = 0.f;
        for (int t = t0; t < 256; t += G) acc += lds[t * 16 + which * 8 + i];
        out2c[o] = acc;
    }
}

// number of 256-thread blocks such that (blocks * 256) % G == 0 (every thread keeps one channel group)
static inline unsigned fixed_group_grid(long items, int G, long cap) {
    long want = (items + 255) / 256;
    if (want > cap) want = cap;
    long a = G, b = 256;
    while (b) { const long t = a % b; a = b; b = t; }       // a = gcd(G, 256)
    const long step = G / a;
    long blocks = want / step * step;
    if (blocks < step) blocks = step;
    return (unsigned)blocks;
}

// error plumbing shared by the C-ABI translation units
void mivp_set_error(const char* msg);
int mivp_check_launch(const char* what);

// opt a kernel into more than 64 KiB of dynamic LDS; a failure is reported like a failed launch
#define MIVP_LDS_OPT_IN(kern, bytes)                                                                                          \
    do {                                                                                                                      \
        if ((bytes) > 64 * 1024) {                                                                                            \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e_ != hipSuccess) { mivp_set_error(hipGetErrorString(e_)); return MIVP_ELAUNCH; }                             \
        }                                                                                                                     \
    } while (0)

#define MIVP_REQUIRE(cond) do { if (!(cond)) { mivp_set_error("contract violated: " #cond); return MIVP_EINVAL; } } while (0)
