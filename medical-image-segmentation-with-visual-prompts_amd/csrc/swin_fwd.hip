// Swin block forward: gather+LN+QKV, prompt K/V, bias augmentation, window attention,
// proj+MLP+scatter.  Reference semantics: swin_transformer/swin_block.py:145-255,
// multi_head_attention/window_attention.py:35-61, relative_positional_encoding.py:99-142.
#include "common.hpp"
#include <cstdlib>

// swin_tok_wide.hip: the column-split token kernels of the wide stages
int mivp_tok_wide_supported(const MivpSwinDesc* d);
int mivp_tok_rows_supported(const MivpSwinDesc* d);
long mivp_tok_natural_offset(int C);
int mivp_tok_wide_qkv_fwd(const MivpSwinDesc* d, const void* x, const int32_t* tok_src, const float* ln_w, const float* ln_b,
                          const void* wqkv, void* q, void* k, void* v, hipStream_t st);
int mivp_tok_wide_proj_mlp_fwd(const MivpSwinDesc* d, const void* o, const void* x, const int32_t* tok_src, const int32_t* tok_dst,
                               const void* wproj, const float* bproj, const float* ln_w, const float* ln_b, const void* wmlp,
                               const float* bmlp, void* t1_out, void* y, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// K1a  gather + LayerNorm + QKV
//   one wave = 32 tokens (two 16-token B tiles) so that every weight fragment fetched from L2
//   feeds two MFMAs; no LDS: the B operand (tokens) is loaded straight into its MFMA lane map
//   (lane r,g holds 8 consecutive channels of token r), LayerNorm statistics are per-lane
//   partials + two xor-shuffles.  gridDim.y splits the 3C output columns: the deep stages have
//   only ~22k tokens (700 waves for 1024 SIMDs), so several waves share a token tile and each
//   takes a slice of the output tiles (the LayerNorm is recomputed per slice, it is cheap).
// ---------------------------------------------------------------------------------------------
//   CH12 (C = 48, head_dim 12, one column split -- stage 0 of every configuration, the largest token count): the 8-byte
//   pieces of the MFMA lane map (token r, channels 4g..) are rows 24 bytes apart, i.e. 64 separate accesses per store for
//   the texture-address unit.  A (q|k|v, head) chunk of a 16-token tile is 384 CONTIGUOUS bytes of the
//   head-major layout (a tile never straddles a window: Nqp % 16 == 0), so the pieces are first moved into address
//   order across the lanes (ds_bpermute: lane l takes piece l = token l/3, part l%3; the crossbar, no LDS memory) and
//   leave as one 48-lane contiguous store per chunk.
template <int KS, bool CH12>
__global__ __launch_bounds__(256, 4) void k_swin_qkv_fwd(MivpSwinDesc d, const bf16_t* __restrict__ x,
                                                      const int* __restrict__ tok_src,
                                                      const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                      const bf16_t* __restrict__ wqkv,
                                                      bf16_t* __restrict__ q, bf16_t* __restrict__ k,
                                                      bf16_t* __restrict__ v) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, hd = C / d.heads;
    const long T = (long)d.B * d.P * d.Nqp;
    const long tile0 = ((long)blockIdx.x * 4 + wave) * 2;

    bf16x8 xb[2][KS];
    long bp[2];
    int slot[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long t = (tile0 + u) * 16 + r;
        live[u] = t < T;
        const unsigned tt = live[u] ? (unsigned)t : 0u;          // T = B*P*Nqp fits 32 bits (checked on the host)
        const unsigned bpu = tt / (unsigned)d.Nqp;
        bp[u] = bpu;
        slot[u] = (int)(tt - bpu * (unsigned)d.Nqp);
        const int pw = (int)(bpu % (unsigned)d.P);
        const long b = bpu / (unsigned)d.P;
        const int src = live[u] ? tok_src[pw * d.Nqp + slot[u]] : -2;
        float xs[KS][8];
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 32 * s + 8 * g;
            if (src >= 0 && c < C) {
                bf16x8 raw = ld8(x + ((b * d.vol_in + src) * (long)C + c));
#pragma unroll
                for (int i = 0; i < 8; ++i) { xs[s][i] = (float)raw[i]; sum += xs[s][i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) xs[s][i] = 0.f;
            }
        }
        const float mean = col_sum(sum) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (32 * s + 8 * g < C) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float dv = xs[s][i] - mean; var += dv * dv; }
            }
        }
        const float rstd = rsqrtf(col_sum(var) / (float)C + d.ln_eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 32 * s + 8 * g;
            bf16x8 y = zero8();
            if (src >= -1 && c < C) {          // -1: zero-pad token still goes through LN (-> beta)
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = (bf16_t)((xs[s][i] - mean) * rstd * ln_w[c + i] + ln_b[c + i]);
            }
            xb[u][s] = y;
        }
    }

    if constexpr (CH12) {
        const unsigned Nqp = (unsigned)d.Nqp;
        // T % 32 == 0 here (the launcher asks for Nqp % 32 == 0): a wave's two tiles are live or dead together, and a dead
        // wave leaves (no barrier in this instantiation)
        if (tile0 * 16 >= T) return;
        long chunk[2];                                          // element offset of head 0's chunk of each token tile
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned tt = (unsigned)((tile0 + u) * 16);
            // (the division is expanded on the vector ALU: pin its uniform result to a scalar register, or every chunk address
            // below is formed per lane in 64-bit vector arithmetic)
            const unsigned bpu = (unsigned)__builtin_amdgcn_readfirstlane((int)(tt / Nqp));
            chunk[u] = ((long)bpu * 4 * Nqp + (tt - bpu * Nqp)) * 12;
        }
        // lanes 48-63 repeat the pieces of lanes 32-47 (same bytes to the same addresses): no exec-mask region around each of
        // the 24 stores (three scalar instructions and a branch apiece in a kernel bound by instruction issue)
        const int dl = lane < 48 ? lane : lane - 16, rq = dl / 3, gq = dl - 3 * rq;
        int src_addr[4];
        bool from_b[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int ch = 12 * h + 4 * gq;                     // channel of this lane's piece inside the q|k|v third
            src_addr[h] = 4 * (rq + 16 * ((ch & 15) >> 2));
            from_b[h] = (ch >> 4) != ((12 * h) >> 4);
        }
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const float sc = sl == 0 ? d.q_scale : (sl == 1 ? MIVP_LOG2E : 1.0f);
            u32x2 pk[3][2];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                f32x4 acc0 = fzero4(), acc1 = fzero4();
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    // scalar fragment base + one 32-bit lane offset (the 64-bit per-lane form was two VALU instructions per load)
                    const char* fb = reinterpret_cast<const char*>(wqkv) + (size_t)(((3 * sl + t) * KS + s) * 1024);
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(fb + (unsigned)(16 * lane));
                    acc0 = mfma16(a, xb[0][s], acc0);
                    acc1 = mfma16(a, xb[1][s], acc1);
                }
                pk[t][0] = __builtin_bit_cast(u32x2, pack4(acc0 * sc));
                pk[t][1] = __builtin_bit_cast(u32x2, pack4(acc1 * sc));
            }
            bf16_t* base = sl == 0 ? q : (sl == 1 ? k : v);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int ta = (12 * h) >> 4, tb = (12 * h + 8) >> 4;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    u32x2 piece;
                    piece[0] = (unsigned)__builtin_amdgcn_ds_bpermute(src_addr[h], (int)pk[ta][u][0]);
                    piece[1] = (unsigned)__builtin_amdgcn_ds_bpermute(src_addr[h], (int)pk[ta][u][1]);
                    if (tb != ta) {
                        const unsigned b0 = (unsigned)__builtin_amdgcn_ds_bpermute(src_addr[h], (int)pk[tb][u][0]);
                        const unsigned b1 = (unsigned)__builtin_amdgcn_ds_bpermute(src_addr[h], (int)pk[tb][u][1]);
                        piece[0] = from_b[h] ? b0 : piece[0];
                        piece[1] = from_b[h] ? b1 : piece[1];
                    }
                    char* cb = reinterpret_cast<char*>(base + (chunk[u] + (long)h * Nqp * 12));     // uniform
                    *reinterpret_cast<u32x2*>(cb + (unsigned)(8 * dl)) = piece;
                }
            }
        }
        return;
    }
    const int n_out = 3 * C;
    const int n_tiles = (n_out + 15) / 16;
    const int per_split = (n_tiles + gridDim.y - 1) / gridDim.y;
    const int nt_begin = blockIdx.y * per_split;
    const int nt_end = nt_begin + per_split < n_tiles ? nt_begin + per_split : n_tiles;
    // wide stages (C >= 96): the four waves share each 16-row weight slab through LDS (see k_swin_proj_mlp_fwd)
    constexpr bool LDSW = KS >= 3;
    constexpr int PCS = (64 * KS + 255) / 256;
    __shared__ __attribute__((aligned(16))) char wsm[LDSW ? 2 * KS * 1024 : 16];
    using WR = OperandRows<32>;
    bf16x8 wreg[PCS];
    auto slab_fetch = [&](int nt) {
#pragma unroll
        for (int u = 0; u < PCS; ++u) {
            wreg[u] = ld8(wqkv + (long)nt * KS * 512 + 8 * min((int)threadIdx.x + 256 * u, 64 * KS - 1));    // fragment image: a linear copy
        }
    };
    auto slab_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PCS; ++u) {
            *reinterpret_cast<bf16x8*>(wsm + buf * KS * 1024 + 16 * min((int)threadIdx.x + 256 * u, 64 * KS - 1)) = wreg[u];
        }
    };
    if (LDSW && nt_begin < nt_end) { slab_fetch(nt_begin); slab_store(0); __syncthreads(); }
    int o_sel = (16 * nt_begin + 4 * g) / C;
    int o_head = ((16 * nt_begin + 4 * g) - o_sel * C) / hd;
    int o_j0 = (16 * nt_begin + 4 * g) - o_sel * C - o_head * hd;
    for (int nt = nt_begin; nt < nt_end; ++nt) {
        f32x4 acc0 = fzero4(), acc1 = fzero4();
        const int nrow = 16 * nt + r;
        if (LDSW) {
            const int cur = (nt - nt_begin) & 1;
            if (nt + 1 < nt_end) slab_fetch(nt + 1);
            const char* slab = wsm + cur * KS * 1024;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(slab + s * 1024 + 16 * lane);
                acc0 = mfma16(a, xb[0][s], acc0);
                acc1 = mfma16(a, xb[1][s], acc1);
            }
            if (nt + 1 < nt_end) slab_store(cur ^ 1);
            __syncthreads();
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 a = wfrag(wqkv, KS, nt, s, lane);
                acc0 = mfma16(a, xb[0][s], acc0);
                acc1 = mfma16(a, xb[1][s], acc1);
            }
        }
        const int n0 = 16 * nt + 4 * g;
        if (n0 < n_out) {
            // (q|k|v, head, offset in head) of output column n0, carried from tile to tile instead of divided out
            while (o_j0 >= hd) { o_j0 -= hd; ++o_head; }
            while (o_head >= d.heads) { o_head -= d.heads; ++o_sel; }
            const int sel = o_sel, head = o_head, j0 = o_j0;
            o_j0 += 16;
            bf16_t* base = sel == 0 ? q : (sel == 1 ? k : v);
            const float sc = sel == 0 ? d.q_scale : (sel == 1 ? MIVP_LOG2E : 1.0f);    // K carries log2(e): common.hpp
            if (live[0]) st4(base + ((bp[0] * d.heads + head) * d.Nqp + slot[0]) * (long)hd + j0, pack4(acc0 * sc));
            if (live[1]) st4(base + ((bp[1] * d.heads + head) * d.Nqp + slot[1]) * (long)hd + j0, pack4(acc1 * sc));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2  prompt tokens -> LN -> K_p, V_p (once per block; 64 rows: plain VALU)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prompt_kv_fwd(MivpSwinDesc d, const float* __restrict__ prompt,
                                                       const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                       const bf16_t* __restrict__ wqkv, bf16_t* __restrict__ kp,
                                                       bf16_t* __restrict__ vp, float* __restrict__ yln) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* y = reinterpret_cast<float*>(smem);             // [C]
    float* red = y + d.C;                                   // [8]
    const int t = blockIdx.x, C = d.C, hd = C / d.heads;
    const int tid = threadIdx.x;
    if (t >= d.Np) {                                        // zero the padding rows
        for (int n = tid; n < 2 * C; n += 256) {
            const int sel = n / C, cc = n - sel * C, head = cc / hd, j = cc - head * hd;
            (sel == 0 ? kp : vp)[((long)head * d.Npp + t) * hd + j] = (bf16_t)0.0f;
        }
        return;
    }
    float part = 0.f;
    for (int c = tid; c < C; c += 256) part += prompt[(long)t * C + c];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    __syncthreads();
    part = 0.f;
    for (int c = tid; c < C; c += 256) { const float dv = prompt[(long)t * C + c] - mean; part += dv * dv; }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = part;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)C + d.ln_eps);
    for (int c = tid; c < C; c += 256) {
        const float yy = (prompt[(long)t * C + c] - mean) * rstd * ln_w[c] + ln_b[c];
        if (yln) yln[(long)t * C + c] = yy;
        y[c] = (float)(bf16_t)yy;                           // same rounding point as the window tokens
    }
    __syncthreads();
    for (int n = tid; n < 2 * C; n += 256) {
        const int sel = n / C, cc = n - sel * C;
        const bf16_t* wrow = wqkv + (long)((sel + 1) * C + cc) * C;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += y[c] * (float)wrow[c];
        const int head = cc / hd, j = cc - head * hd;
        (sel == 0 ? kp : vp)[((long)head * d.Npp + t) * hd + j] = (bf16_t)(sel == 0 ? acc * MIVP_LOG2E : acc);
    }
}

// K2 for several blocks in one launch (blockIdx.y = block), which also writes the prompt-token bias ts into the i0 one-hot
// columns of the block's K'-augmentation image: with frozen content tables everything else in that image is constant, so
// a prompted block's per-step small forward work is this launch plus the (equally batched) token scores -- instead of
// token scores + mivp_relbias_aug + mivp_prompt_kv_fwd per block.
struct PromptKvJob {
    MivpSwinDesc d;
    const float* prompt; const float* ln_w; const float* ln_b; const bf16_t* wqkv; const float* ts;
    bf16_t* kp; bf16_t* vp; bf16_t* ka;
};
struct PromptKvJobs { PromptKvJob job[16]; };
__global__ __launch_bounds__(256) void k_prompt_kv_fwd_multi(PromptKvJobs jobs) {
    const PromptKvJob& jb = jobs.job[blockIdx.y];
    const MivpSwinDesc& d = jb.d;
    const int t = blockIdx.x;
    if (t >= d.Npp) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* y = reinterpret_cast<float*>(smem);             // [C]
    float* red = y + d.C;                                   // [8]
    const int C = d.C, hd = C / d.heads;
    const int tid = threadIdx.x;
    const float* __restrict__ prompt = jb.prompt;
    if (t >= d.Np) {                                        // zero the padding rows
        for (int n = tid; n < 2 * C; n += 256) {
            const int sel = n / C, cc = n - sel * C, head = cc / hd, j = cc - head * hd;
            (sel == 0 ? jb.kp : jb.vp)[((long)head * d.Npp + t) * hd + j] = (bf16_t)0.0f;
        }
        return;
    }
    // prompt-token bias: ka[head][Nqp + t][a < w0] = ts[head][t] * log2 e (mivp_relbias_aug's prompt rows)
    for (int n = tid; n < d.heads * d.win[0]; n += 256) {
        const int head = n / d.win[0], a = n - head * d.win[0];
        jb.ka[((long)head * d.Nkp + d.Nqp + t) * d.augp + a] = (bf16_t)(jb.ts[(long)head * d.Np + t] * MIVP_LOG2E);
    }
    float part = 0.f;
    for (int c = tid; c < C; c += 256) part += prompt[(long)t * C + c];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    __syncthreads();
    part = 0.f;
    for (int c = tid; c < C; c += 256) { const float dv = prompt[(long)t * C + c] - mean; part += dv * dv; }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = part;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)C + d.ln_eps);
    for (int c = tid; c < C; c += 256)
        y[c] = (float)(bf16_t)((prompt[(long)t * C + c] - mean) * rstd * jb.ln_w[c] + jb.ln_b[c]);   // same rounding point as the window tokens
    __syncthreads();
    for (int n = tid; n < 2 * C; n += 256) {
        const int sel = n / C, cc = n - sel * C;
        const bf16_t* wrow = jb.wqkv + (long)((sel + 1) * C + cc) * C;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += y[c] * (float)wrow[c];
        const int head = cc / hd, j = cc - head * hd;
        (sel == 0 ? jb.kp : jb.vp)[((long)head * d.Npp + t) * hd + j] = (bf16_t)(sel == 0 ? acc * MIVP_LOG2E : acc);
    }
}

// ---------------------------------------------------------------------------------------------
// K3  relative-position bias -> augmentation columns of Q' / K'
//   bias[n,m] = Th[k0-i0+w0-1] + Tw[k1-i1+w1-1] + Td[k2-i2+w2-1]   (tables pre-scaled by s/3)
//   query side: one-hot(i0) | one-hot(i1) | one-hot(i2) without its last entry
//   key side  : Th[.. for each i0] + c | Tw[..] | Td[..] - Td_last,  c = Td at i2 = w2-1
//   prompt key t: ts[t] in every i0 column (the i0 one-hot sums to one).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_relbias_aug(MivpSwinDesc d, const float* __restrict__ t_h,
                                                     const float* __restrict__ t_w, const float* __restrict__ t_d,
                                                     const float* __restrict__ ts, bf16_t* __restrict__ qa,
                                                     bf16_t* __restrict__ ka) {
    // grid = (heads + 1, row chunks): block row `heads` writes the head-independent query one-hots
    const int head = blockIdx.x;
    const int w0 = d.win[0], w1 = d.win[1], w2 = d.win[2];
    const int A = d.augp;
    const int chunk = blockIdx.y, nchunk = gridDim.y;
    if (head == d.heads) {
        for (int e = chunk * 256 + threadIdx.x; e < d.Nqp * A; e += nchunk * 256) {
            const int n = e / A, a = e - n * A;
            float val = 0.f;
            if (n < d.Nq) {
                const int i2 = n % w2, i1 = (n / w2) % w1, i0 = n / (w2 * w1);
                if (a < w0) val = (a == i0) ? 1.f : 0.f;
                else if (a < w0 + w1) val = (a - w0 == i1) ? 1.f : 0.f;
                else if (a < w0 + w1 + w2 - 1) val = (a - w0 - w1 == i2) ? 1.f : 0.f;
            }
            qa[e] = (bf16_t)val;
        }
        return;
    }
    const float* th = t_h + (long)head * (2 * w0 - 1);
    const float* tw = t_w + (long)head * (2 * w1 - 1);
    const float* td = t_d + (long)head * (2 * w2 - 1);
    for (int e = chunk * 256 + threadIdx.x; e < d.Nkp * A; e += nchunk * 256) {
        const int m = e / A, a = e - m * A;
        float val = 0.f;
        if (m < d.Nq) {
            const int k2 = m % w2, k1 = (m / w2) % w1, k0 = m / (w2 * w1);
            const float fold = td[k2 - (w2 - 1) + w2 - 1];          // query i2 = w2-1
            if (a < w0) val = th[k0 - a + w0 - 1] + fold;
            else if (a < w0 + w1) val = tw[k1 - (a - w0) + w1 - 1];
            else if (a < w0 + w1 + w2 - 1) val = td[k2 - (a - w0 - w1) + w2 - 1] - fold;
        } else if (m >= d.Nqp && m < d.Nqp + d.Np) {
            if (a < w0) val = ts[(long)head * d.Np + (m - d.Nqp)];
        } else {
            if (a < w0) val = MIVP_PAD_KEY_BIAS;                   // padding key: common.hpp
        }
        ka[((long)head * d.Nkp + m) * A + a] = (bf16_t)(val * MIVP_LOG2E);
    }
}

// ---------------------------------------------------------------------------------------------
// K1b  window attention forward, one workgroup (8 waves) per (window instance, head)
//   LDS: K' image [Nkp][DK+8] (rows = head dims | bias aug | zero pad), V^T image [16*DVT][Nkp+8],
//        key classes.  Q' fragments are assembled per query tile straight from global.
//   Each wave walks 16-query tiles; S^T = K' Q'^T puts ONE query on each lane, so the softmax row
//   reductions are in-lane + 2 shuffles and P feeds the PV MFMA with no lane movement (k index
//   permuted identically on both operands).  Keys are consumed 32 at a time with an online softmax
//   (running max / sum, O rescaled by exp(m_old - m_new)): only two score tiles are live, which
//   keeps the kernel under 128 VGPRs so that 16+ waves per CU hide the LDS / MFMA / exp latency
//   (this kernel is bound by VALU + transcendental issue, not by MFMA: DESIGN.md section 4).
// ---------------------------------------------------------------------------------------------
//   ZREF (forward-only calls: lse == NULL, nothing saved for a backward pass): the optimistic walk keeps the reference point
//   at ZERO -- P = exp2(s); attention logits in log2 units sit within a few tens of zero and f32 / bf16 carry 2^+-126 -- so
//   the first step is an ordinary step too: no per-lane maximum, no cross-lane agreement, no subtraction of the new
//   reference from two score tiles and the accumulator seed (70 fewer instructions per query tile, -3 % / -4.5 % per
//   stage-0 launch).  A row sum outside [2^-100, 2^100) sends the tile to the tested walk.  Calls that save for backward
//   keep the first-step maximum: their rounding pattern is the one the backward parity bars were measured with
//   (oracle/swin_ref.py models both: ``zero_ref``).
// LDS-DMA staging helpers of the DMA form (below): constant source words (zero | bf16 (1, 0)), one 4-byte global -> LDS
// transfer per lane (LDS address = wave-uniform base + 4 * lane), and the transposing LDS read (wgrad.hip)
__device__ __attribute__((aligned(16))) unsigned int g_attn_consts[4] = {0u, 0x00003F80u, 0u, 0u};
MIVP_DEV void glds4(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
typedef __attribute__((address_space(3))) bf16x4 attn_lds_bf16x4;
MIVP_DEV bf16x4 attn_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((attn_lds_bf16x4*)p);
}

// waves per SIMD asked of the register allocator: the one-k-step kernels hold four workgroups per CU (34 KB of LDS each at 7^3
// windows) when they fit 64 VGPRs
template <int DKS, bool DROP, bool MASKED, bool ZREF, bool DMA, bool BITS>
constexpr int attn_fwd_occupancy() {
    if (DKS != 1) return 2;
    if (DROP) return DMA ? 6 : 2;            // (dropout kernels: ~90-115 VGPRs left alone = two workgroups per CU; 80 = three)
    // (the masked kernels need 68-72 VGPRs, with byte classes and with mask words: capped at 64 they spill around every tile)
    if (DMA) return MASKED ? 6 : 8;
    return (MASKED && ZREF) ? 8 : 2;
}
//   BITS (shifted blocks, mivp.h "mask words"): the shift mask of a (query tile, key tile) pair comes as four 64-bit lane
//   masks from a per-geometry table (scalar loads -> v_cndmask on an SGPR pair: ONE vector instruction per logit) instead of
//   byte classes compared per logit (extract + compare + select); cut windows cost ~2x an uncut one with the compares.
template <int DKS, int DVT, int NW, int QT, bool DROP, bool ONES, bool MASKED, bool ZREF = false, bool DMA = false, bool BITS = false>
__global__ __launch_bounds__(64 * NW, (attn_fwd_occupancy<DKS, DROP, MASKED, ZREF, DMA, BITS>())) void k_win_attn_fwd(MivpSwinDesc d, const bf16_t* __restrict__ q,
                                                         const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                         const bf16_t* __restrict__ kp, const bf16_t* __restrict__ vp,
                                                         const bf16_t* __restrict__ qa, const bf16_t* __restrict__ ka,
                                                         const int* __restrict__ tok_rid, bf16_t* __restrict__ o,
                                                         float* __restrict__ lse, int xcd_remap,
                                                         const unsigned long long* __restrict__ mbits,
                                                         const unsigned char* __restrict__ cutw) {
    // dropout keys of this call (common.hpp drop_seed: uniform, scalar-ALU work; unused without dropout)
    const uint32_t attn_key = drop_seed(d.attn_seed, d.seed_epoch), proj_key = drop_seed(d.proj_seed, d.seed_epoch);
    (void)attn_key; (void)proj_key;
    static_assert(!BITS || MASKED, "mask words belong to the masked kernels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DK = 32 * DKS;
    using KR = OperandRows<DK>;
    constexpr int KROW = KR::ROW;                            // bytes
    // Softmax denominator from the matrix pipe: when the value tile has a spare row (head_dim < 16*DVT) and there is no
    // dropout, row head_dim of V^T is set to one, so O's row head_dim accumulates sum_k P -- of the same bf16-rounded P
    // the numerator uses -- and the eight adds per 32 keys leave the (binding) VALU stream.  (ONES is picked at launch.)
    const int Nkp = d.Nkp, Nqp = d.Nqp;
    // DMA form (DKS == DVT == 1): V stays ROW-major [Nkp][16] (32-byte rows) and the PV product's A operand comes out of
    // transposing reads; the classic form keeps a V^T image [16 DVT][Nkp + 8]
    static_assert(!DMA || (DKS == 1 && DVT == 1), "the DMA-staged images are laid out for one k-step / one value tile");
    const int VROW = DMA ? 32 : (Nkp + 8) * 2;               // bytes
    char* Kimg = smem;
    char* Vt = Kimg + (size_t)Nkp * KROW;
    // key classes as bytes (255 padding, 254 prompt, else region id): with 4-byte classes the image is 128 B over a
    // quarter of the CU's 160 KB at 7^3 windows, i.e. three workgroups per CU instead of four
    uint8_t* ridk = reinterpret_cast<uint8_t*>(Vt + (DMA ? (size_t)Nkp * 32 : (size_t)(16 * DVT) * VROW));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the query-tile loop and its addresses stay in SGPRs
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, heads = d.heads, hd = C / heads;
    // (window, head) item of this workgroup: (b*P + pw)*heads + head.  Workgroups go round-robin over the 8 XCDs; with the
    // remap consecutive items (the heads of a window, neighbouring windows) run on ONE XCD, i.e. meet in one L2
    long bph = blockIdx.x;
    if (xcd_remap) bph = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int head = (int)(bph % heads);
    const long bp = bph / heads;
    const int pw = (int)(bp % d.P);
    const int A = d.augp;
    const int hd4 = hd / 4, dk4 = DK / 4, a4 = A / 4;

    // Staging is VALU work in a VALU-bound kernel (it used to be more than a third of the kernel's vector instructions):
    // per-workgroup bases are uniform, a thread keeps ONE column of the image so its source, stride and LDS column are
    // loop invariants, and offsets are 32-bit.
    const bf16_t* kb = k + bph * (long)Nqp * hd;             // this (window, head)'s rows
    const bf16_t* vb = v + bph * (long)Nqp * hd;
    const bf16_t* kpb = d.Np > 0 ? kp + (long)head * d.Npp * hd : kb;
    const bf16_t* vpb = d.Np > 0 ? vp + (long)head * d.Npp * hd : vb;
    const bf16_t* kab = ka + (long)head * Nkp * A;
    const int n_prompt_rows = d.Np > 0 ? d.Npp : 0;
    if constexpr (DMA) {
        // ---- LDS-DMA staging (global_load_lds_dword): a lane names its own 4-byte source, a wave-instruction fills 256
        //      consecutive LDS bytes, no VGPR destination, no ds_write, no transposes: the K' and V images cost ~3 vector
        //      instructions per 256 bytes instead of ~25 (this kernel is bound by VALU issue and the register-path staging
        //      was a third of its vector instructions).
        //      K' image: one instruction = 4 rows (64-byte rows); dword slot `sl` of a row holds logical dword
        //      4 ((sl >> 2) ^ swz) + (sl & 3) -- the chunk swizzle of OperandRows<32> moved to the SOURCE address -- which
        //      comes from k / kp (head dims), ka (bias columns) or the zero word.  Row groups go round-robin over the waves:
        //      group & 3 == wave & 3, so a lane's swizzle, source kind and column are constants and a source pointer only
        //      advances by a per-lane stride.
        const char* zsrc = reinterpret_cast<const char*>(g_attn_consts);
        const int hd2 = hd >> 1, a2 = A >> 1;
        {
            const int sl = lane & 15, rowin = lane >> 4;
            const int ld = 4 * ((sl >> 2) ^ ((0 - wave) & 3)) + (sl & 3);
            const int kind = ld < hd2 ? 0 : (ld < hd2 + a2 ? 1 : 2);
            const int row0 = 4 * wave + rowin;
            const long stride = kind == 0 ? 64L * hd : (kind == 1 ? 64L * A : 0L);      // bytes per 32 rows
            const char* src = kind == 0 ? reinterpret_cast<const char*>(kb) + ((long)row0 * hd + 2 * ld) * 2
                            : kind == 1 ? reinterpret_cast<const char*>(kab) + ((long)row0 * A + 2 * (ld - hd2)) * 2 : zsrc;
            int gi = wave;
            for (; gi < Nqp / 4; gi += NW) { glds4(src, Kimg + gi * 256); src += stride; }
            // prompt rows, then padding rows (head dims: zero; their bias columns exclude them)
            if (kind == 0) src = reinterpret_cast<const char*>(kpb) + ((long)(4 * gi + rowin - Nqp) * hd + 2 * ld) * 2;
            for (; gi < Nkp / 4; gi += NW) {
                const bool live = 4 * gi < Nqp + n_prompt_rows;                          // wave-uniform
                glds4((kind == 0 && !live) ? zsrc : src, Kimg + gi * 256);
                src += stride;
            }
        }
        // V image: one instruction = 8 rows of [16 dv] (32-byte rows): dwords 0 .. hd/2-1 from v / vp, dword hd/2 = (1, 0)
        // when the softmax denominator rides on the PV product (ONES), the rest zero
        {
            const int sl = lane & 7, rowin = lane >> 3;
            const int row0 = 8 * wave + rowin;
            const bool fv = sl < hd2;
            const char* csrc = (ONES && sl == hd2) ? zsrc + 4 : zsrc;
            const long stride = fv ? 128L * hd : 0L;                                     // bytes per 64 rows
            const char* src = fv ? reinterpret_cast<const char*>(vb) + ((long)row0 * hd + 2 * sl) * 2 : csrc;
            int gi = wave;
            for (; gi < Nqp / 8; gi += NW) { glds4(src, Vt + gi * 256); src += stride; }
            if (fv) src = reinterpret_cast<const char*>(vpb) + ((long)(8 * gi + rowin - Nqp) * hd + 2 * sl) * 2;
            for (; gi < Nkp / 8; gi += NW) {
                const bool live = 8 * gi < Nqp + n_prompt_rows;
                glds4((fv && !live) ? zsrc : src, Vt + gi * 256);
                src += stride;
            }
        }
    } else {
    // ---- stage K' ----  (loads of four pieces in flight per thread before the first LDS write)
    {
        constexpr int RPP = 64 * NW / dk4;                   // rows per pass
        const int c4 = tid % dk4, row0 = tid / dk4;
        const bool from_k = c4 < hd4, from_a = !from_k && c4 < hd4 + a4;
        const bf16_t* src = from_k ? kb : kab;
        const uint32_t stride = from_k ? hd : A, coff = from_k ? 4 * c4 : 4 * (c4 - hd4);
        const int rows_main = from_k ? Nqp : (from_a ? Nkp : 0);
        for (int rowb = row0; rowb < Nkp; rowb += 4 * RPP) {
            bf16x4 vals[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rowb + i * RPP;
                bf16x4 val = zero4();
                if (row < rows_main) val = ld4(src + ((uint32_t)row * stride + coff));
                else if (from_k && row < Nqp + n_prompt_rows) val = ld4(kpb + ((uint32_t)(row - Nqp) * hd + coff));
                vals[i] = val;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rowb + i * RPP;
                if (row < Nkp) *reinterpret_cast<bf16x4*>(Kimg + KR::off(row, 4 * c4)) = vals[i];
            }
        }
    }
    // ---- stage V^T (zero rows dv >= hd, zero key columns beyond the staged rows): a thread takes four consecutive keys
    //      of one 4-channel group, transposes them in registers and writes four 8-byte row pieces ----
    {
        constexpr int CPR = 4 * DVT, RPP = 64 * NW / CPR;
        const int c4 = tid % CPR;
        const bool from_v = c4 < hd4;
        for (int rq4 = tid / CPR; rq4 < Nkp / 4; rq4 += RPP) {
            bf16x4 in[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * rq4 + i;
                bf16x4 val = zero4();
                if (from_v) {
                    if (row < Nqp) val = ld4(vb + ((uint32_t)row * hd + 4 * c4));
                    else if (row < Nqp + n_prompt_rows) val = ld4(vpb + ((uint32_t)(row - Nqp) * hd + 4 * c4));
                }
                if (ONES && c4 == hd4) val[0] = (bf16_t)1.0f; // V^T row hd = 1: the PV product then also returns sum_k P
                in[i] = val;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 outv;
                outv[0] = in[0][j]; outv[1] = in[1][j]; outv[2] = in[2][j]; outv[3] = in[3][j];
                *reinterpret_cast<bf16x4*>(Vt + (size_t)(4 * c4 + j) * VROW + 2 * (4 * rq4)) = outv;
            }
        }
    }
    }
    // ---- key classes (classify_logit in common.hpp) ----
    if (MASKED && !BITS) {                                   // (only the masked steps of the class-compare form read them)
        for (int m = tid; m < Nkp; m += 64 * NW) {
            // content key: region id; prompt and padding keys: 254 = never masked (padding keys are excluded by their bias)
            ridk[m] = (uint8_t)(m < d.Nq ? tok_rid[pw * Nqp + m] : 254);
        }
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs have landed; the barrier publishes them
    __syncthreads();
    // Most windows of a shifted block are not cut by the volume boundary: all their content tokens share one region id
    // and the mask is a no-op.  Those windows take the unmasked steps (workgroup-uniform choice).
    bool cut = false;
    if (MASKED && BITS) {
        cut = cutw[pw] != 0;                                 // per-window flag of the mask table (uniform: a scalar load)
    } else if (MASKED) {
        int differs = 0;
        for (int m = tid; m < d.Nq; m += 64 * NW) differs |= ridk[m] != ridk[0];
        cut = __syncthreads_or(differs) != 0;
    }

    const int npairs = Nkp / 32;                             // key tiles come in pairs (Nkp % 32 == 0)
    const int nt_full = d.Nq / 16;                           // tiles made of valid content keys only
    const int nkt_c = Nqp / 16;                              // key tiles that hold content rows (the mask table covers these)
    constexpr float RESCALE_LOG2 = 8.f;

    // A wave carries QT query tiles through the key loop at once: every K' / V^T fragment read from LDS serves QT tiles
    // (the LDS pipe is the co-bottleneck of this kernel) and the QT softmax chains are independent instruction streams.
    const int nqt = Nqp / 16;
    const bf16_t* qb = q + bph * (long)Nqp * hd;
    bf16_t* ob = o + bp * (long)Nqp * C + head * hd;
    // Q' fragments (+ region ids) of a query tile: unconditional loads (common.hpp "Branch-free loads"), issued ONE TILE AHEAD:
    // written as conditional loads in the tile's prologue they were 2-3 dependent memory round trips in front of every
    // ~1 us key loop of the wave.
    // DMA form: a lane's piece source (q, the bias one-hots or -- beyond head_dim + bias columns -- the zero word) is a lane
    // constant: a running pointer per piece that advances by a per-lane stride from one tile of this wave to the next (64-bit
    // multiply-adds and keep-masks per tile were ~10 of the ~60 vector instructions a tile spends outside its key loop).
    // Classic form (kept for the masked forward-only kernel, which fits 64 VGPRs -- four workgroups per CU -- only with the
    // round-2 addressing): per-lane piece offsets of query tile 0 and per-tile strides, one multiply-add per piece and tile.
    const char* qsrc[DKS][2];
    int q_tile[DKS][2];                                      // bytes from one query tile to the next (0: constant source)
    const long to_qa = qa - qb;
    long q_off0[DKS][2];
    int q_step[DKS][2];
    bool q_keep[DKS][2];
#pragma unroll
    for (int s = 0; s < DKS; ++s)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int c4 = 8 * s + 2 * g + hlf;
            const bool fq = c4 < hd4, fa = !fq && c4 < hd4 + a4;
            const int row0 = QT * wave * 16 + r;
            qsrc[s][hlf] = fq ? reinterpret_cast<const char*>(qb) + (row0 * hd + 4 * c4) * 2
                         : fa ? reinterpret_cast<const char*>(qa) + (row0 * A + 4 * (c4 - hd4)) * 2
                              : reinterpret_cast<const char*>(g_attn_consts) + 8;      // (two zero words)
            q_tile[s][hlf] = fq ? 32 * hd : (fa ? 32 * A : 0);
            const int ca = min(max(c4 - hd4, 0), a4 - 1);
            q_off0[s][hlf] = sel(c4 < hd4, (long)(r * hd + 4 * min(c4, hd4 - 1)), to_qa + (long)(r * A + 4 * ca));
            q_step[s][hlf] = sel(c4 < hd4, 16 * hd, 16 * A);
            q_keep[s][hlf] = c4 < hd4 + a4;
        }
    // loads the Q' fragments of tiles qt_first .. ; DMA form: the pointers stand there and step to this wave's next turn
    auto load_q = [&](int qt_first, bf16x8 (&qfo)[QT][DKS], uint32_t (&rqo)[QT]) {
        if constexpr (DMA) {
#pragma unroll
            for (int a = 0; a < QT; ++a) {
                const int ao = (qt_first + a < nqt) ? a : 0;  // an odd tile count: the spare slot shadows the first tile
                const int row = (qt_first + ao) * 16 + r;
                rqo[a] = (MASKED && !BITS) ? (uint32_t)sel(row < d.Nq, (int)ridk[min(row, d.Nq - 1)], 0) : 0u;   // (the key classes hold the same ids)
#pragma unroll
                for (int s = 0; s < DKS; ++s)
                    qfo[a][s] = cat44(*reinterpret_cast<const bf16x4*>(qsrc[s][0] + ao * q_tile[s][0]),
                                      *reinterpret_cast<const bf16x4*>(qsrc[s][1] + ao * q_tile[s][1]));
            }
#pragma unroll
            for (int s = 0; s < DKS; ++s)
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) qsrc[s][hlf] += QT * NW * q_tile[s][hlf];
        } else {
#pragma unroll
            for (int a = 0; a < QT; ++a) {
                const int qt = (qt_first + a < nqt) ? qt_first + a : (qt_first < nqt ? qt_first : 0);
                const int row = qt * 16 + r;
                rqo[a] = (MASKED && !BITS) ? (uint32_t)sel(row < d.Nq, tok_rid[pw * Nqp + min(row, d.Nq - 1)], 0) : 0u;
#pragma unroll
                for (int s = 0; s < DKS; ++s) {
                    bf16x4 piece[2];
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf)
                        piece[hlf] = keep_if(ld4(qb + (q_off0[s][hlf] + (long)(qt * q_step[s][hlf]))), q_keep[s][hlf]);
                    qfo[a][s] = cat44(piece[0], piece[1]);
                }
            }
        }
    };
    bf16x8 qf_next[QT][DKS];
    uint32_t rq_next[QT];
    if (!DMA || QT * wave < nqt) load_q(QT * wave, qf_next, rq_next);
    for (int qt0 = QT * wave; qt0 < nqt; qt0 += QT * NW) {
        int qrow[QT];
        const unsigned long long* mrow[QT];                  // (BITS) mask words of this query tile: [key tile][4]
        uint32_t rq[QT];
        bf16x8 qf[QT][DKS];
        f32x4 oacc[QT][DVT], negm[QT];                       // negm = -(reference point): the accumulator the S MFMA starts from
        float mrun[QT], lsum[QT];                            // reference point of P in log2 units (set by the first step)
        uint32_t drow[QT];
#pragma unroll
        for (int a = 0; a < QT; ++a) {
            const int qt = (qt0 + a < nqt) ? qt0 + a : qt0;  // an odd tile count: the spare slot shadows tile qt0, nothing stored
            qrow[a] = qt * 16 + r;
            if (BITS) mrow[a] = mbits + ((long)pw * nqt + qt) * (long)(4 * nkt_c);
            rq[a] = rq_next[a];
#pragma unroll
            for (int s = 0; s < DKS; ++s) qf[a][s] = qf_next[a][s];
        }
        if (!DMA || qt0 + QT * NW < nqt) load_q(qt0 + QT * NW, qf_next, rq_next);   // the next tile's operands travel under this tile's key loop
        bool first = true;
        auto reset = [&]() {
#pragma unroll
            for (int a = 0; a < QT; ++a) {
#pragma unroll
                for (int dd = 0; dd < DVT; ++dd) oacc[a][dd] = fzero4();
                negm[a] = fzero4();
                mrun[a] = 0.f;
                lsum[a] = 0.f;
                drow[a] = DROP ? attn_row(bph, qrow[a], Nqp, Nkp) : 0u;
            }
            first = true;
        };
        reset();

        // One step = 32 keys.  TAIL steps hold padding / prompt keys and classify every logit; the others only apply
        // the shift mask (MASK).  Logits arrive in log2 units relative to the reference point (common.hpp).  The
        // reference is refreshed lazily: only when some lane of the wave sees a logit more than RESCALE_LOG2 above it
        // are O (and the sum) rescaled, otherwise P is formed against the older reference (P <= 2^RESCALE_LOG2, exact
        // after the final division) -- the cross-lane max, the exp, the subtract and the O multiplies leave the VALU /
        // LDS streams for almost every step.  max3 / max2 are v_maximum3_f32: fmaxf() makes the compiler canonicalise
        // each MFMA result first (one extra VALU op per logit).
        // OPT (optimistic) steps skip the per-lane maximum and the wave vote altogether: after the first step has set the
        // reference point, P = exp2(s - ref) cannot overflow unless a later logit exceeds it by ~100 (log2 units), and bf16 / f32
        // keep full relative precision at any scale.  The tile's row sums are checked at the end; a tile that did overflow
        // is redone with the tested steps (never seen on real attention logits; tests force it).  That removes 4
        // v_maximum3 + compare + vote from the 30 vector instructions of a step in a kernel bound by VALU issue.
        auto step = [&](int u, auto tail_c, auto mask_c, auto opt_c) {
            constexpr bool TAIL = decltype(tail_c)::value;
            constexpr bool MASK = decltype(mask_c)::value;
            constexpr bool OPT = decltype(opt_c)::value;
            bf16x8 kfr[2][DKS];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int s = 0; s < DKS; ++s)
                    kfr[hh][s] = *reinterpret_cast<const bf16x8*>(Kimg + KR::off(16 * (2 * u + hh) + r, 32 * s + 8 * g));
            uint32_t kcl[2] = {0u, 0u};
            if (MASK && !BITS) {
                kcl[0] = *reinterpret_cast<const uint32_t*>(ridk + 16 * (2 * u) + 4 * g);
                kcl[1] = *reinterpret_cast<const uint32_t*>(ridk + 16 * (2 * u + 1) + 4 * g);
            }
            f32x4 sv[QT][2];                                 // log2-unit logits minus the reference point
            float pmax[QT];
            bool grow = first;
#pragma unroll
            for (int a = 0; a < QT; ++a) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int t = 2 * u + hh;
                    // ZREF optimistic steps: the reference point is the constant zero (an inline-constant accumulator seed)
                    constexpr bool ZSEED = ZREF && OPT;
                    const float masked_logit = ZSEED ? 0.f : negm[a][0];
                    f32x4 acc = ZSEED ? fzero4() : negm[a];
#pragma unroll
                    for (int s = 0; s < DKS; ++s) acc = mfma16(kfr[hh][s], qf[a][s], acc);
                    if (MASK && BITS) {
                        // four lane masks of this (query tile, key tile): bit 16 g + r = "logit (query r, key 4 g + j) survives";
                        // tiles beyond the content rows (prompt keys) are never masked and have no words
                        if (!TAIL || t < nkt_c) {
                            const unsigned long long* mw = mrow[a] + 4 * t;
                            acc[0] = __builtin_amdgcn_inverse_ballot_w64(mw[0]) ? acc[0] : masked_logit;
                            acc[1] = __builtin_amdgcn_inverse_ballot_w64(mw[1]) ? acc[1] : masked_logit;
                            acc[2] = __builtin_amdgcn_inverse_ballot_w64(mw[2]) ? acc[2] : masked_logit;
                            acc[3] = __builtin_amdgcn_inverse_ballot_w64(mw[3]) ? acc[3] : masked_logit;
                        }
                    } else if (MASK) {
                        const uint32_t kr = kcl[hh];
                        if (!TAIL || t < nt_full) {
                            acc[0] = ((kr & 0xFFu) == rq[a]) ? acc[0] : masked_logit;
                            acc[1] = (((kr >> 8) & 0xFFu) == rq[a]) ? acc[1] : masked_logit;
                            acc[2] = (((kr >> 16) & 0xFFu) == rq[a]) ? acc[2] : masked_logit;
                            acc[3] = ((kr >> 24) == rq[a]) ? acc[3] : masked_logit;
                        } else {                             // prompt / padding keys (254) are never masked
                            const uint32_t k0 = kr & 0xFFu, k1 = (kr >> 8) & 0xFFu, k2 = (kr >> 16) & 0xFFu, k3 = kr >> 24;
                            acc[0] = (k0 == rq[a] || k0 == 254u) ? acc[0] : masked_logit;
                            acc[1] = (k1 == rq[a] || k1 == 254u) ? acc[1] : masked_logit;
                            acc[2] = (k2 == rq[a] || k2 == 254u) ? acc[2] : masked_logit;
                            acc[3] = (k3 == rq[a] || k3 == 254u) ? acc[3] : masked_logit;
                        }
                    }
                    sv[a][hh] = acc;
                }
                if (!OPT) {
                    float pm = max3_raw(sv[a][0][0], sv[a][0][1], sv[a][0][2]);
                    pm = max3_raw(pm, sv[a][0][3], sv[a][1][0]);
                    pm = max3_raw(pm, sv[a][1][1], sv[a][1][2]);
                    pmax[a] = max2_raw(pm, sv[a][1][3]);     // this lane's 8 keys only: enough for the test
                    grow = grow || (pmax[a] > RESCALE_LOG2);
                }
            }
            if (!OPT && (first || __any(grow))) {            // wave-uniform
                asm volatile("" ::: "memory");               // keep it a branch: if-converted it costs 12 VALU ops per step
#pragma unroll
                for (int a = 0; a < QT; ++a) {
                    float pm = pmax[a];
                    pm = max2_raw(pm, __shfl_xor(pm, 16));   // the query's four lanes must agree on the new point
                    pm = max2_raw(pm, __shfl_xor(pm, 32));
                    // the first pair holds valid content keys (finite max) and always sets the reference point
                    const float up = first ? pm : max2_raw(pm, 0.f);
                    const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-up);
                    mrun[a] += up;
                    negm[a] = negm[a] - up;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) sv[a][hh] = sv[a][hh] - up;
                    if (!ONES) lsum[a] *= alpha;
#pragma unroll
                    for (int dd = 0; dd < DVT; ++dd) oacc[a][dd] = oacc[a][dd] * alpha;
                }
                first = false;
            }
            bf16x8 vfr[DVT];
            if constexpr (DMA) {
                // V^T fragment by transposing reads of the row-major image: the 16-lane group g passes the addresses of keys
                // 4g .. 4g+3 (lane 4q+p: row q, dv 4p ..), lane r receives dv = r of those four keys
                const char* vblk = Vt + (32 * u + 4 * g + (r >> 2)) * 32 + 8 * (r & 3);
                vfr[0] = cat44(attn_tr_read(vblk), attn_tr_read(vblk + 512));
            } else {
#pragma unroll
                for (int dd = 0; dd < DVT; ++dd) {
                    const char* vrow = Vt + (size_t)(16 * dd + r) * VROW;
                    vfr[dd] = cat44(*reinterpret_cast<const bf16x4*>(vrow + (32 * u + 4 * g) * 2),
                                    *reinterpret_cast<const bf16x4*>(vrow + (32 * u + 16 + 4 * g) * 2));
                }
            }
#pragma unroll
            for (int a = 0; a < QT; ++a) {
                float psum = 0.f;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float pe = __builtin_amdgcn_exp2f(sv[a][hh][j]); sv[a][hh][j] = pe; if (!ONES) psum += pe; }
                if (!ONES) lsum[a] += psum;
                if (DROP) {                                  // attention dropout acts on P after the softmax sum
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int key0 = 16 * (2 * u + hh) + 4 * g;
                        const uint32_t pi = attn_pair(drow[a], key0);
                        const uint32_t h0 = drop_hash(pi, attn_key), h1 = drop_hash(pi + 1, attn_key);
                        sv[a][hh][0] = drop_keep(h0, 0, d.attn_drop_thr) ? sv[a][hh][0] : 0.f;
                        sv[a][hh][1] = drop_keep(h0, 1, d.attn_drop_thr) ? sv[a][hh][1] : 0.f;
                        sv[a][hh][2] = drop_keep(h1, 0, d.attn_drop_thr) ? sv[a][hh][2] : 0.f;
                        sv[a][hh][3] = drop_keep(h1, 1, d.attn_drop_thr) ? sv[a][hh][3] : 0.f;
                    }
                }
                bf16x8 pb;
                if (DROP) {
                    // pairs converted with ONE v_cvt_pk_bf16_f32 each, AFTER the selects: left to itself the compiler converts
                    // the eight values one by one, selects on the halves and re-packs them with v_perm (20 instructions for 12)
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    u32x4 pk;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f32x2_t two = {sv[a][hh][2 * e], sv[a][hh][2 * e + 1]};
                            pk[2 * hh + e] = __builtin_bit_cast(unsigned, __builtin_convertvector(two, bf16x2_t));
                        }
                    pb = __builtin_bit_cast(bf16x8, pk);
                } else {
                    pb = cat44(pack4(sv[a][0]), pack4(sv[a][1]));
                }
#pragma unroll
                for (int dd = 0; dd < DVT; ++dd) oacc[a][dd] = mfma16(vfr[dd], pb, oacc[a][dd]);
            }
        };
        const int nfull = nt_full / 2;
        auto walk = [&](auto mask_c, auto opt_c) {
            // the first step computes the maximum (it sets the reference point) -- except in the ZREF optimistic walk
            using first_opt = std::integral_constant<bool, ZREF && decltype(opt_c)::value>;
            if (nfull > 0) step(0, std::false_type{}, mask_c, first_opt{});
            else step(0, std::true_type{}, mask_c, first_opt{});
            // the optimistic steps two at a time (by hand: the pragma is refused around the tested steps' branch): LDS
            // addresses become immediate offsets, one pointer bump and one loop test per 64 keys
            int u = 1;
            if (decltype(opt_c)::value && !decltype(mask_c)::value) {
                for (; u + 1 < nfull; u += 2) { step(u, std::false_type{}, mask_c, opt_c); step(u + 1, std::false_type{}, mask_c, opt_c); }
            }
            for (; u < nfull; ++u) step(u, std::false_type{}, mask_c, opt_c);
            for (u = nfull > 1 ? nfull : 1; u < npairs; ++u) step(u, std::true_type{}, mask_c, opt_c);
        };
        auto row_sum_of = [&](int a) -> float {
            float ls = lsum[a];
            if (ONES) {                                      // sum_k P sits in O's row hd: lane (r, g = (hd%16)/4), element hd%4
                const int dd1 = hd >> 4, e1 = hd & 3, g1 = (hd & 15) >> 2;
                float pick = 0.f;
#pragma unroll
                for (int dd = 0; dd < DVT; ++dd)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pick = (dd == dd1 && e == e1) ? oacc[a][dd][e] : pick;
                ls = __shfl(pick, r + 16 * g1);
            } else {
                ls = col_sum(ls);
            }
            return ls;
        };
        if (MASKED && cut) walk(std::true_type{}, std::true_type{});
        else walk(std::false_type{}, std::true_type{});
        float ls_of[QT];
        {
            bool bad = false;
#pragma unroll
            // 2^100 (also catches NaN); ZREF: and 2^-100 -- the zero reference was too far above this row's logits
            for (int a = 0; a < QT; ++a) {
                ls_of[a] = row_sum_of(a);
                bad = bad || !(ls_of[a] < 1.2676506e30f) || (ZREF && !(ls_of[a] > 7.8886091e-31f));
            }
            if (__any(bad)) {                                // overflow of the optimistic steps: redo the tile with the tested ones
                reset();
                if (MASKED && cut) walk(std::true_type{}, std::false_type{});
                else walk(std::false_type{}, std::false_type{});
#pragma unroll
                for (int a = 0; a < QT; ++a) ls_of[a] = row_sum_of(a);
            }
        }
#pragma unroll
        for (int a = 0; a < QT; ++a) {
            if (a > 0 && qt0 + a >= nqt) continue;
            const float ls = ls_of[a];
            const float inv = (DROP ? d.attn_drop_scale : 1.0f) * __builtin_amdgcn_rcpf(ls);
#pragma unroll
            for (int dd = 0; dd < DVT; ++dd) {
                const int j0 = 16 * dd + 4 * g;
                if (j0 < hd) st4(ob + ((uint32_t)qrow[a] * C + j0), pack4(oacc[a][dd] * inv));
            }
            // lse feeds the backward passes only: a forward-only call (frozen block without prompts, evaluation) passes NULL
            if (!ZREF && lse && g == 0) lse[bph * Nqp + qrow[a]] = (mrun[a] + __builtin_amdgcn_logf(ls)) * MIVP_LN2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1c  proj + residual -> LayerNorm -> Linear + residual -> scatter/crop
//   two chained GEMMs per token without LDS: the accumulator of GEMM1 (lane r,g owns channels
//   16*t+4g..+3 of token r) becomes the B operand of GEMM2 by pairing tiles (2s, 2s+1) into one
//   32-deep k-step and reading the weight row in the same permuted k order.
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256, CT >= 12 ? 2 : (CT >= 6 ? 4 : 5)) void k_swin_proj_mlp_fwd(MivpSwinDesc d, const bf16_t* __restrict__ o,
                                                           const bf16_t* __restrict__ x,
                                                           const int* __restrict__ tok_src, const int* __restrict__ tok_dst,
                                                           const bf16_t* __restrict__ wproj, const float* __restrict__ bproj,
                                                           const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                           const bf16_t* __restrict__ wmlp, const float* __restrict__ bmlp,
                                                           bf16_t* __restrict__ t1_out, bf16_t* __restrict__ y) {
    // dropout keys of this call (common.hpp drop_seed: uniform, scalar-ALU work; unused without dropout)
    const uint32_t attn_key = drop_seed(d.attn_seed, d.seed_epoch), proj_key = drop_seed(d.proj_seed, d.seed_epoch);
    (void)attn_key; (void)proj_key;
    constexpr int KS = (CT + 1) / 2;
    // Wide stages (C >= 96) are bound by fetching the weights: every wave used to pull the whole [C][C] matrix through
    // L1 for its 16 tokens.  There the workgroup's four waves share each 16-row weight slab through LDS (KS sub-tiles of
    // [16 rows][64 B] in the swizzled operand layout, two slabs in flight: global -> registers one slab ahead).
    constexpr bool LDSW = CT >= 6;
    constexpr int PCS = (64 * KS + 255) / 256;                      // 16-byte pieces of a slab per thread
    __shared__ __attribute__((aligned(16))) char wsm[LDSW ? 2 * KS * 1024 : 16];
    using WR = OperandRows<32>;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C;
    const long T = (long)d.B * d.P * d.Nqp;
    const long t = ((long)blockIdx.x * 4 + wave) * 16 + r;
    const bool live = t < T;
    const unsigned tt = live ? (unsigned)t : 0u;             // T = B*P*Nqp fits 32 bits (checked on the host): 32-bit divisions
    const unsigned bpu = tt / (unsigned)d.Nqp;
    const long bp = bpu;
    const int slot = (int)(tt - bpu * (unsigned)d.Nqp);
    const int pw = (int)(bpu % (unsigned)d.P);
    const long b = bpu / (unsigned)d.P;
    const int src = sel(live, tok_src[pw * d.Nqp + slot], -2);       // (a dead lane decodes as token 0: the address is valid)
    const int dst = sel(live, tok_dst[pw * d.Nqp + slot], -1);
    // Loads are unconditional (clamped address, masked value) and batched: common.hpp "Branch-free loads".
    bf16x8 wreg[PCS];
    auto slab_fetch = [&](const bf16_t* __restrict__ w, int nt) {
#pragma unroll
        for (int u = 0; u < PCS; ++u) {
            wreg[u] = ld8(w + (long)nt * KS * 512 + 8 * min((int)threadIdx.x + 256 * u, 64 * KS - 1));        // fragment image: a linear copy
        }
    };
    auto slab_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PCS; ++u) {
            *reinterpret_cast<bf16x8*>(wsm + buf * KS * 1024 + 16 * min((int)threadIdx.x + 256 * u, 64 * KS - 1)) = wreg[u];
        }
    };

    bf16x8 ob[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * g;
        ob[s] = keep_if(ld8(o + (tt * (long)C + min(c, C - 8))), live && c < C);
    }
    bf16x4 xraw[CT];                                             // the shortcut row (one round trip behind tok_src)
    const long xrow = (b * d.vol_in + max(src, 0)) * (long)C;
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) xraw[nt] = ld4(x + xrow + min(16 * nt + 4 * g, C - 4));
    // GEMM1: t1 = o Wproj^T + b + shortcut
    f32x4 t1[CT];
    float sum = 0.f;
    if (LDSW) { slab_fetch(wproj, 0); slab_store(0); __syncthreads(); }
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) {
        f32x4 acc = fzero4();
        const int nrow = 16 * nt + r;
        if (LDSW) {
            if (nt + 1 < CT) slab_fetch(wproj, nt + 1);
            const char* slab = wsm + (nt & 1) * KS * 1024;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc = mfma16(*reinterpret_cast<const bf16x8*>(slab + s * 1024 + 16 * lane), ob[s], acc);
            if (nt + 1 < CT) slab_store((nt + 1) & 1);        // that buffer was last read in iteration nt-1
            __syncthreads();
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = mfma16(wfrag(wproj, KS, nt, s, lane), ob[s], acc);
            }
        }
        const int n0 = 16 * nt + 4 * g;
        const f32x4 bp4 = *reinterpret_cast<const f32x4*>(bproj + min(n0, C - 4));
        if (n0 < C) {
            f32x4 sc;
            { const bf16x4 xv = keep_if(xraw[nt], src >= 0); for (int j = 0; j < 4; ++j) sc[j] = (float)xv[j]; }
            f32x4 keep = {1.f, 1.f, 1.f, 1.f};
            if (d.proj_drop_thr) {                           // proj dropout: on proj(o) + b, before the residual
                const uint32_t pi = (uint32_t)((tt * C + n0) >> 1);
                const uint32_t h0 = drop_hash(pi, proj_key), h1 = drop_hash(pi + 1, proj_key);
                keep[0] = drop_keep(h0, 0, d.proj_drop_thr) ? d.proj_drop_scale : 0.f;
                keep[1] = drop_keep(h0, 1, d.proj_drop_thr) ? d.proj_drop_scale : 0.f;
                keep[2] = drop_keep(h1, 0, d.proj_drop_thr) ? d.proj_drop_scale : 0.f;
                keep[3] = drop_keep(h1, 1, d.proj_drop_thr) ? d.proj_drop_scale : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float val = (float)(bf16_t)((acc[j] + bp4[j]) * keep[j] + sc[j]);   // t1 lives as bf16
                acc[j] = val;
                sum += val;
            }
            if (t1_out && live) st4(t1_out + (tt * (long)C + n0), pack4(acc));
        } else {
            acc = fzero4();
        }
        t1[nt] = acc;
    }
    const float mean = col_sum(sum) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) {
        if (16 * nt + 4 * g < C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dv = t1[nt][j] - mean; var += dv * dv; }
        }
    }
    const float rstd = rsqrtf(col_sum(var) / (float)C + d.ln_eps);
    bf16x4 y2[2 * KS];
#pragma unroll
    for (int nt = 0; nt < 2 * KS; ++nt) {
        bf16x4 yy = zero4();
        if (nt < CT) {
            const int n0 = 16 * nt + 4 * g;
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(ln_w + min(n0, C - 4));
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(ln_b + min(n0, C - 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) yy[j] = (bf16_t)((t1[nt][j] - mean) * rstd * w4[j] + b4[j]);
            yy = keep_if(yy, n0 < C);
        }
        y2[nt] = yy;
    }
    // GEMM2: t2 = t1 + y2 Wmlp^T + b ; k-step s covers channels 32s..32s+31 in the order
    // kappa = 8g+j' -> channel 32s + 16*(j'>>2) + 4g + (j'&3)
    if (LDSW) { slab_fetch(wmlp, 0); slab_store(0); __syncthreads(); }     // slab 0 was last read before the final barrier of GEMM1
#pragma unroll
    for (int mt = 0; mt < CT; ++mt) {
        f32x4 acc = fzero4();
        const int nrow = 16 * mt + r;
        if (LDSW) {
            if (mt + 1 < CT) slab_fetch(wmlp, mt + 1);
            const char* slab = wsm + (mt & 1) * KS * 1024;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                // (paired image: the fragment is already lo | hi)
                acc = mfma16(*reinterpret_cast<const bf16x8*>(slab + s * 1024 + 16 * lane), cat44(y2[2 * s], y2[2 * s + 1]), acc);
            }
            if (mt + 1 < CT) slab_store((mt + 1) & 1);
            __syncthreads();
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = mfma16(wfrag(wmlp, KS, mt, s, lane), cat44(y2[2 * s], y2[2 * s + 1]), acc);
            }
        }
        const int n0 = 16 * mt + 4 * g;
        const f32x4 bm4 = *reinterpret_cast<const f32x4*>(bmlp + min(n0, C - 4));
        if (n0 < C && dst >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += t1[mt][j] + bm4[j];
            st4(y + ((b * d.vol_out + dst) * (long)C + n0), pack4(acc));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int swin_common_checks(const MivpSwinDesc* d) {
    MIVP_REQUIRE(d != nullptr);
    MIVP_REQUIRE(d->B > 0 && d->C > 0 && d->heads > 0 && d->P > 0);
    MIVP_REQUIRE(d->C % 8 == 0 && d->C % d->heads == 0 && (d->C / d->heads) % 4 == 0);
    MIVP_REQUIRE(d->Nqp % 16 == 0 && d->Nqp >= d->Nq && d->Nqp - d->Nq < 16);
    MIVP_REQUIRE((long)d->B * d->P * d->Nqp < (1L << 31));      // kernels decode token indices in 32 bits
    MIVP_REQUIRE(d->Npp % 16 == 0 && d->Npp >= d->Np);
    MIVP_REQUIRE(d->Nkp % 32 == 0 && d->Nkp >= d->Nqp + d->Npp);
    MIVP_REQUIRE(d->augp % 4 == 0 && d->augp >= d->aug);
    MIVP_REQUIRE(d->aug == d->win[0] + d->win[1] + d->win[2] - 1);
    MIVP_REQUIRE(d->Nq == d->win[0] * d->win[1] * d->win[2]);
    return MIVP_OK;
}

extern "C" int mivp_swin_qkv_fwd(const MivpSwinDesc* d, const void* x, const int32_t* tok_src, const float* ln_w,
                                 const float* ln_b, const void* wqkv, void* q, void* k, void* v,
                                 mivp_stream_t stream) {
    int rc = swin_common_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(x && tok_src && ln_w && ln_b && wqkv && q && k && v);
    if (mivp_tok_wide_supported(d)) return mivp_tok_wide_qkv_fwd(d, x, tok_src, ln_w, ln_b, wqkv, q, k, v, (hipStream_t)stream);
    const long T = (long)d->B * d->P * d->Nqp;
    const unsigned gx = (unsigned)((T + 127) / 128);
    const int KS = (d->C + 31) / 32;
    // enough waves for ~4 per SIMD: split the output columns when the token count alone cannot provide them
    const int n_tiles = (3 * d->C + 15) / 16;
    int nsplit = (int)((4096 + 4L * gx - 1) / (4L * gx));
    if (nsplit > n_tiles / 3) nsplit = n_tiles / 3;
    if (nsplit < 1) nsplit = 1;
    const dim3 grid(gx, (unsigned)nsplit);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_QKV2(K, CH) hipLaunchKernelGGL((k_swin_qkv_fwd<K, CH>), grid, dim3(256), 0, st, *d, (const bf16_t*)x, tok_src, \
                                              ln_w, ln_b, (const bf16_t*)wqkv, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v)
#define LAUNCH_QKV(K) LAUNCH_QKV2(K, false)
    // contiguous chunk stores (kernel header): C = 48 with four heads of 12, all output columns in one workgroup
    const bool ch12 = d->C == 48 && d->heads == 4 && nsplit == 1 && d->Nqp % 32 == 0;
    switch (KS) {
        case 1: LAUNCH_QKV(1); break;
        case 2: if (ch12) LAUNCH_QKV2(2, true); else LAUNCH_QKV(2); break;
        case 3: LAUNCH_QKV(3); break;
        case 4: LAUNCH_QKV(4); break;
        case 6: LAUNCH_QKV(6); break;
        default: mivp_set_error("swin_qkv_fwd: C/32 not in {1,2,3,4,6}"); return MIVP_EUNSUPPORTED;
    }
#undef LAUNCH_QKV
    return mivp_check_launch("swin_qkv_fwd");
}

extern "C" int mivp_prompt_kv_fwd(const MivpSwinDesc* d, const float* prompt, const float* ln_w, const float* ln_b,
                                  const void* wqkv, void* kp, void* vp, float* yln, mivp_stream_t stream) {
    int rc = swin_common_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(d->Np > 0 && prompt && ln_w && ln_b && wqkv && kp && vp);
    hipLaunchKernelGGL(k_prompt_kv_fwd, dim3(d->Npp), dim3(256), (d->C + 8) * sizeof(float), (hipStream_t)stream, *d,
                       prompt, ln_w, ln_b, (const bf16_t*)wqkv, (bf16_t*)kp, (bf16_t*)vp, yln);
    return mivp_check_launch("prompt_kv_fwd");
}

/* n <= 16 blocks, arrays of n entries each (host memory); descriptors need C, heads, Np, Npp, Nqp, Nkp, augp, win, ln_eps.
 * ka[i]: the block's K'-augmentation image as mivp_relbias_aug wrote it for ts = 0 (its prompt rows are rewritten here). */
extern "C" int mivp_prompt_kv_fwd_multi(int32_t n, const MivpSwinDesc* d, const float* const* prompt, const float* const* ln_w,
                                        const float* const* ln_b, const void* const* wqkv, const float* const* ts,
                                        void* const* kp, void* const* vp, void* const* ka, mivp_stream_t stream) {
    MIVP_REQUIRE(n > 0 && n <= 16 && d && prompt && ln_w && ln_b && wqkv && ts && kp && vp && ka);
    PromptKvJobs jobs;
    int rows = 0;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        int rc = swin_common_checks(&d[i]);
        if (rc) return rc;
        MIVP_REQUIRE(d[i].Np > 0 && prompt[i] && ln_w[i] && ln_b[i] && wqkv[i] && ts[i] && kp[i] && vp[i] && ka[i]);
        jobs.job[i] = PromptKvJob{d[i], prompt[i], ln_w[i], ln_b[i], (const bf16_t*)wqkv[i], ts[i], (bf16_t*)kp[i], (bf16_t*)vp[i],
                                  (bf16_t*)ka[i]};
        rows = d[i].Npp > rows ? d[i].Npp : rows;
        const size_t need = (d[i].C + 8) * sizeof(float);
        lds = need > lds ? need : lds;
    }
    hipLaunchKernelGGL(k_prompt_kv_fwd_multi, dim3(rows, n), dim3(256), lds, (hipStream_t)stream, jobs);
    return mivp_check_launch("prompt_kv_fwd_multi");
}

extern "C" int mivp_relbias_aug(const MivpSwinDesc* d, const float* t_h, const float* t_w, const float* t_d,
                                const float* ts, void* qa, void* ka, mivp_stream_t stream) {
    int rc = swin_common_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(t_h && t_w && t_d && qa && ka);
    MIVP_REQUIRE(d->Np == 0 || ts != nullptr);
    hipLaunchKernelGGL(k_relbias_aug, dim3(d->heads + 1, 8), dim3(256), 0, (hipStream_t)stream, *d, t_h, t_w, t_d, ts,
                       (bf16_t*)qa, (bf16_t*)ka);
    return mivp_check_launch("relbias_aug");
}

template <int DKS, int DVT, int NW, int QT, bool DMA, bool BITS>
static int launch_attn_fwd_cfg(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp,
                               const void* vp, const void* qa, const void* ka, const int32_t* tok_rid, void* o, float* lse,
                               const unsigned long long* mask_words, const unsigned char* cut_flags, hipStream_t st) {
    const size_t krow = OperandRows<32 * DKS>::ROW, vrow = (d->Nkp + 8) * 2;
    const size_t lds = (size_t)d->Nkp * krow + (DMA ? (size_t)d->Nkp * 32 : (size_t)16 * DVT * vrow) + (size_t)d->Nkp;
    if (lds > 160 * 1024) { mivp_set_error("win_attn_fwd: LDS image exceeds 160 KiB"); return MIVP_EUNSUPPORTED; }
    const bool ones = !d->attn_drop_thr && (d->C / d->heads) < 16 * DVT;
    const bool msk = d->has_mask != 0;
    const bool zref = lse == nullptr && !d->attn_drop_thr;      // forward only (kernel header: ZREF)
    // (BITS only ever instantiates masked kernels: the un-masked arms below pass false)
    auto kern = d->attn_drop_thr
        ? (msk ? k_win_attn_fwd<DKS, DVT, NW, QT, true, false, true, false, DMA, BITS> : k_win_attn_fwd<DKS, DVT, NW, QT, true, false, false, false, DMA, false>)
        : ones ? (zref ? (msk ? k_win_attn_fwd<DKS, DVT, NW, QT, false, true, true, true, DMA, BITS> : k_win_attn_fwd<DKS, DVT, NW, QT, false, true, false, true, DMA, false>)
                       : (msk ? k_win_attn_fwd<DKS, DVT, NW, QT, false, true, true, false, DMA, BITS> : k_win_attn_fwd<DKS, DVT, NW, QT, false, true, false, false, DMA, false>))
               : (zref ? (msk ? k_win_attn_fwd<DKS, DVT, NW, QT, false, false, true, true, DMA, BITS> : k_win_attn_fwd<DKS, DVT, NW, QT, false, false, false, true, DMA, false>)
                       : (msk ? k_win_attn_fwd<DKS, DVT, NW, QT, false, false, true, false, DMA, BITS> : k_win_attn_fwd<DKS, DVT, NW, QT, false, false, false, false, DMA, false>));
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { mivp_set_error(hipGetErrorString(e)); return MIVP_ELAUNCH; }
    }
    const unsigned grid = (unsigned)((long)d->B * d->P * d->heads);
    static const bool no_remap = getenv("MIVP_ATTN_NO_XCD_REMAP") != nullptr;
    const int xcd_remap = (!no_remap && grid % 8 == 0 && grid >= 64) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, *d, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       (const bf16_t*)kp, (const bf16_t*)vp, (const bf16_t*)qa, (const bf16_t*)ka, tok_rid, (bf16_t*)o, lse, xcd_remap,
                       mask_words, cut_flags);
    return mivp_check_launch("win_attn_fwd");
}

template <int DKS, int DVT>
static int launch_attn_fwd(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp,
                           const void* vp, const void* qa, const void* ka, const int32_t* tok_rid, void* o, float* lse,
                           const unsigned long long* mw, const unsigned char* cf, hipStream_t st) {
    // (NW, QT) = (4, 2) -- two query tiles per wave sharing every K' / V^T fragment -- measures the same as (8, 1) at 7^3
    // windows (87.5 vs 87.2 us per stage-1 block): the kernel is bound by VALU issue, not by the LDS pipe
    static const bool no_bits = getenv("MIVP_ATTN_MASK_CLASSES") != nullptr;       // A/B: byte classes instead of mask words
    const bool bits = d->has_mask && mw != nullptr && cf != nullptr && !no_bits;
    // one k-step / one value tile (head_dim <= 16: the encoder stages and the last decoder stage): LDS-DMA staged images
    // (MIVP_ATTN_FWD_REG_STAGING=1 keeps the register-path staging for A/B runs)
    if constexpr (DKS == 1 && DVT == 1) {
        static const bool reg_staging = getenv("MIVP_ATTN_FWD_REG_STAGING") != nullptr;
        // (the masked kernels need ~72 VGPRs in this form with byte classes: three workgroups per CU)
        if (!reg_staging && d->Nqp % 8 == 0) {
            if (bits) return launch_attn_fwd_cfg<DKS, DVT, 8, 1, true, true>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, lse, mw, cf, st);
            return launch_attn_fwd_cfg<DKS, DVT, 8, 1, true, false>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, lse, mw, cf, st);
        }
    }
    if (bits) return launch_attn_fwd_cfg<DKS, DVT, 8, 1, false, true>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, lse, mw, cf, st);
    return launch_attn_fwd_cfg<DKS, DVT, 8, 1, false, false>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, lse, mw, cf, st);
}

// shared by forward and backward dispatch: which (NT, DKS=DVT) instantiation covers this shape
int mivp_attn_tile_config(const MivpSwinDesc* d, int* dks, int* nt) {
    const int hd = d->C / d->heads;
    const int dk = hd + d->augp;
    int ks = (dk + 31) / 32;
    const int vt = (hd + 15) / 16;
    if (vt > ks) ks = vt;                   // instantiations are diagonal: (1,1) (2,2) (3,3)
    *dks = ks;
    *nt = d->Nkp / 16;
    if (ks > 3) return MIVP_EUNSUPPORTED;
    return MIVP_OK;
}

extern "C" int mivp_win_attn_fwd(const MivpSwinDesc* d, const void* q, const void* k, const void* v, const void* kp,
                                 const void* vp, const void* qa, const void* ka, const int32_t* tok_rid, void* o,
                                 float* lse, const uint64_t* mask_words, const uint8_t* cut_flags, mivp_stream_t stream) {
    int rc = swin_common_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(q && k && v && qa && ka && o);             // lse may be NULL (ABI 11): forward only, nothing saved
    MIVP_REQUIRE(d->Np == 0 || (kp && vp));
    MIVP_REQUIRE(!d->has_mask || tok_rid);
    int dks, nt;
    if (mivp_attn_tile_config(d, &dks, &nt)) { mivp_set_error("win_attn_fwd: head_dim / key count outside the instantiated set"); return MIVP_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long* mw = reinterpret_cast<const unsigned long long*>(mask_words);
    if (dks == 1) return launch_attn_fwd<1, 1>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, lse, mw, cut_flags, st);
    if (dks == 2) return launch_attn_fwd<2, 2>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, lse, mw, cut_flags, st);
    return launch_attn_fwd<3, 3>(d, q, k, v, kp, vp, qa, ka, tok_rid, o, lse, mw, cut_flags, st);
}

extern "C" int mivp_swin_proj_mlp_fwd(const MivpSwinDesc* d, const void* o, const void* x, const int32_t* tok_src,
                                      const int32_t* tok_dst, const void* wproj, const float* bproj, const float* ln_w,
                                      const float* ln_b, const void* wmlp, const float* bmlp, void* t1, void* y,
                                      mivp_stream_t stream) {
    int rc = swin_common_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(o && x && tok_src && tok_dst && wproj && bproj && ln_w && ln_b && wmlp && bmlp && y);
    const long T = (long)d->B * d->P * d->Nqp;
    const unsigned grid = (unsigned)((T + 63) / 64);
    const int CT = (d->C + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    // C = 48 / 96 / 192 / 384: row-image kernel (proj dropout included, round 3), natural-order image of wmlp (it follows the paired one)
    if (mivp_tok_rows_supported(d))
        return mivp_tok_wide_proj_mlp_fwd(d, o, x, tok_src, tok_dst, wproj, bproj, ln_w, ln_b,
                                          (const bf16_t*)wmlp + mivp_tok_natural_offset(d->C), bmlp, t1, y, st);
#define LAUNCH_PM(K) hipLaunchKernelGGL((k_swin_proj_mlp_fwd<K>), dim3(grid), dim3(256), 0, st, *d, (const bf16_t*)o, \
                                         (const bf16_t*)x, tok_src, tok_dst, (const bf16_t*)wproj, bproj, ln_w, ln_b,   \
                                         (const bf16_t*)wmlp, bmlp, (bf16_t*)t1, (bf16_t*)y)
    switch (CT) {
        case 1: LAUNCH_PM(1); break;
        case 2: LAUNCH_PM(2); break;
        case 3: LAUNCH_PM(3); break;
        case 4: LAUNCH_PM(4); break;
        case 6: LAUNCH_PM(6); break;
        case 8: LAUNCH_PM(8); break;
        case 12: LAUNCH_PM(12); break;
        default: mivp_set_error("swin_proj_mlp_fwd: C/16 not in {1,2,3,4,6,8,12}"); return MIVP_EUNSUPPORTED;
    }
#undef LAUNCH_PM
    return mivp_check_launch("swin_proj_mlp_fwd");
}
