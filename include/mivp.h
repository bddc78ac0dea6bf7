/*
 * mivp.h -- C ABI of the MI355X-native (gfx950) Swin-UNETR hot path.
 *
 * Drop-in boundary: the reference has no FFI of its own -- its
 * hot path is eager PyTorch inside `SwinUnetR.forward` -- so these entry points
 * are what a binding from the reference's Python would call (ctypes stub in
 * INTEGRATION.md).  Each entry cites the reference lines it replaces (paths
 * relative to /root/reference/src/modules).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless it is a descriptor struct;
 *  - activations are bf16, channels-last `[B, H, W, D, C]` ("NDHWC");
 *  - parameters arrive as bf16 (GEMM weights, `[out][in]` row-major like
 *    nn.Linear.weight) or fp32 (norm scale/shift, biases);
 *  - no allocation, no host sync, no global state inside any call: the caller
 *    owns every buffer (torch allocations in the Python host) and passes the
 *    HIP stream to launch on; calls are thread-compatible;
 *  - return 0 on success, negative MIVP_E* on error; nothing throws.
 *
 * Window index tables (`tok_src`, `tok_dst`, `tok_rid`) are immutable per
 * (dims, window, shift) and are built by the host exactly as SURVEY Appendix
 * A.1 states (swin_transformer/swin_block.py:145-178,247-253,292-364):
 *    tok_src[P*Nqp]  voxel offset (h*W+w)*D+d inside one volume of x that feeds
 *                    token (window, slot); -1 = zero-pad token; -2 = slot >= Nq
 *    tok_dst[P*Nqp]  voxel offset in the block OUTPUT that the token writes; -1 = cropped
 *    tok_rid[P*Nqp]  int32 region id of the token for the shift mask
 * with Nqp = Nq rounded up to 16.
 */
#ifndef MIVP_H
#define MIVP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mivp_stream_t;     /* hipStream_t */

#define MIVP_OK            0
#define MIVP_EINVAL       -1     /* bad argument / shape contract violated          */
#define MIVP_EUNSUPPORTED -2     /* shape outside the instantiated kernel set       */
#define MIVP_ELAUNCH      -3     /* hipLaunch reported an error                     */

/* library identity: returns the ABI version (bumped on any signature change) */
int mivp_abi_version(void);
/* last HIP error string for MIVP_ELAUNCH (static storage, never NULL) */
const char* mivp_last_error(void);

/* ------------------------------------------------------------------------ */
/* Swin block (swin_block.py:145-255, window_attention.py:35-61)             */
/* ------------------------------------------------------------------------ */
typedef struct MivpSwinDesc {
    int32_t B;          /* batch                                              */
    int32_t C;          /* channels, multiple of 8                            */
    int32_t heads;      /* C % heads == 0, (C/heads) % 4 == 0                  */
    int32_t vol_in;     /* H*W*D of the block input  (voxels per volume)       */
    int32_t vol_out;    /* H*W*D of the block output (== vol_in)               */
    int32_t P;          /* windows per volume                                  */
    int32_t Nq;         /* tokens per window                                   */
    int32_t Nqp;        /* Nq rounded up to 16                                 */
    int32_t Np;         /* prompt tokens (0 = none)                            */
    int32_t Npp;        /* rows of the prompt K/V buffers (>= Np, multiple of 16) */
    int32_t Nkp;        /* key rows the attention kernel sees: multiple of 32, >= Nqp+Npp */
    int32_t aug;        /* bias augmentation dims  w0 + w1 + (w2-1)            */
    int32_t augp;       /* aug rounded up to 4                                 */
    int32_t has_mask;   /* 1 for a shifted block                               */
    int32_t win[3];     /* window size per axis                                */
    float   q_scale;    /* head_dim ** -0.5                                    */
    float   ln_eps;     /* 1e-6                                                */
    /* dropout of the training forward (window_attention.py:30,33,57,60): an element is dropped when the   */
    /* 16-bit counter hash of its index under `seed` is below `thr` (= round(p * 65536), 0 = no dropout);   */
    /* kept elements are scaled by `scale` = 65536 / (65536 - thr).  Backward re-derives the same mask.     */
    uint32_t attn_drop_thr;   float attn_drop_scale;   uint32_t attn_seed;
    uint32_t proj_drop_thr;   float proj_drop_scale;   uint32_t proj_seed;
    /* ABI 12: optional DEVICE pointer to one 32-bit epoch word (NULL = none).  Every dropout kernel uses the seeds      */
    /* seed + epoch[0] * 0x9E3779B1: a recorded HIP graph freezes this descriptor (a kernel argument), so a replay draws */
    /* a fresh mask by incrementing the word in device memory at the start of the graph; forward and backward kernels of */
    /* one step read the same value.                                                                                     */
    const uint32_t* seed_epoch;
} MivpSwinDesc;

/* Weight fragment images (ABI 9).  The four token kernels below read their GEMM weights as MFMA-fragment-major images:
 *   [ceil(rows / 16) row tiles][k_steps][64 lanes][8] bf16, zero padded; lane 16 g + r of (row tile nt, k-step s) holds
 *   W[16 nt + r][32 s + 8 g .. + 8]  ("natural"), or W[..][32 s + 4 g .. + 4] | W[..][32 s + 16 + 4 g .. + 4] ("paired": the k
 *   order of a GEMM whose B operand is the previous GEMM's accumulator tile pair).  A wave's A fragment is then one
 *   contiguous 1 KB load.  k_steps = ceil(cols / 32), except wqkv_t: ceil(3 * 16 * ceil(C / 16) / 32).
 *   mivp_pack_weight_frags builds one image from a row-major [rows][cols] bf16 matrix (out: ceil(rows/16) * k_steps * 512
 *   elements).  Images per entry:
 *     mivp_swin_qkv_fwd       wqkv    = natural image of [3C][C]
 *     mivp_swin_proj_mlp_fwd  wproj   = natural image of [C][C];  wmlp = paired image of [C][C], and for C in {48, 96, 192,
 *                                       384} the natural image directly behind it (the row-image kernels of those widths
 *                                       feed the second GEMM from LDS in natural k order; dropout calls use the paired one)
 *     mivp_swin_proj_mlp_bwd  wmlp_t  = natural image of Wmlp^T;  wproj_t = paired (+ natural, as above) image of Wproj^T
 *     mivp_swin_qkv_bwd       wqkv_t  = natural image of Wqkv^T [C][3C]
 *   (mivp_prompt_kv_fwd / _bwd keep the row-major wqkv.)                                                                  */
int mivp_pack_weight_frags(const void* w, int32_t rows, int32_t cols, int32_t k_steps, int32_t paired, void* out,
                           mivp_stream_t stream);
/* Every image of one Swin block in one launch, from the f32 [C][C] masters of to_q, to_k, to_v, proj and mlp
 * (swin_block.py:60-80).  Outputs bf16, any may be NULL: wqkv_rm [3C][C] row-major (for mivp_prompt_kv_fwd / _bwd),
 * wqkv_f, wproj_f, wmlp_f, wqkv_t, wmlp_t, wproj_t exactly as the table above; with_natural != 0 appends the natural images
 * behind the paired ones of wmlp_f / wproj_t (element offset ceil(C/16) * ceil(C/32) * 512).                            */
int mivp_pack_block_weights(int32_t C, const float* wq, const float* wk, const float* wv, const float* wproj,
                            const float* wmlp, int32_t with_natural, void* wqkv_rm, void* wqkv_f, void* wproj_f,
                            void* wmlp_f, void* wqkv_t, void* wmlp_t, void* wproj_t, mivp_stream_t stream);

/* gather + LayerNorm + q/k/v projections  (swin_block.py:205-214, window_attention.py:42-47)
 *   x [B, vol_in, C] bf16;  ln_w, ln_b [C] f32;  wqkv = fragment image of [3C][C] (to_q, to_k, to_v stacked), see above
 *   q, k, v [B*P][heads][Nqp][hd] bf16 ; q is pre-multiplied by q_scale ; slots >= Nq are zero.
 *   k is stored multiplied by log2(e): the attention kernels keep logits in log2 units so that S = K'Q'^T feeds
 *   v_exp_f32 directly (the gradient entries below still take / return dk w.r.t. the un-scaled k).           */
int mivp_swin_qkv_fwd(const MivpSwinDesc* d, const void* x, const int32_t* tok_src,
                      const float* ln_w, const float* ln_b, const void* wqkv,
                      void* q, void* k, void* v, mivp_stream_t stream);

/* prompt tokens -> LayerNorm -> to_k / to_v, once per block
 *   prompt [Np][C] f32 (the nn.Parameter) -> kp, vp [heads][Npp][hd] bf16 (rows >= Np zero; kp x log2(e) like k)
 *   yln [Np][C] f32: the normalised prompt, saved for backward                            */
int mivp_prompt_kv_fwd(const MivpSwinDesc* d, const float* prompt, const float* ln_w, const float* ln_b,
                       const void* wqkv, void* kp, void* vp, float* yln, mivp_stream_t stream);

/* prompt-token bias scores and their gradients (relative_positional_encoding.py:128-135):
 *   ts[h][t] = scale * sum_k W[h][k] * E[t][k]   (W = weights_token [heads][e], E = enc_token rows [Np][e], all f32)
 *   dW[h][k] = scale * sum_t dts[h][t] E[t][k],  dE[t][k] = scale * sum_h dts[h][t] W[h][k]                       */
int mivp_token_scores_fwd(const float* W, const float* E, int32_t heads, int32_t np, int32_t e, float scale, float* ts,
                          mivp_stream_t stream);
int mivp_token_scores_bwd(const float* dts, const float* W, const float* E, int32_t heads, int32_t np, int32_t e, float scale,
                          float* dW, float* dE, mivp_stream_t stream);
/* Batched forms for prompt tuning (n <= 16 blocks per call, arrays of n host-side entries): every one of these kernels lasts
 * 5-10 us whatever it computes, and a step runs one per prompted block.  mivp_prompt_kv_fwd_multi also writes the token bias
 * ts into the prompt rows of each block's K'-augmentation image ka[i] (an image mivp_relbias_aug produced with ts = 0: with
 * frozen content tables the rest of it is constant), so mivp_relbias_aug leaves the per-step path of those blocks. */
int mivp_token_scores_fwd_multi(int32_t n, const float* const* W, const float* const* E, const int32_t* heads, const int32_t* np,
                                int32_t e, const float* scale, float* const* ts, mivp_stream_t stream);
int mivp_token_scores_bwd_multi(int32_t n, const float* const* dts /* entries may be NULL */, const float* const* W,
                                const float* const* E, const int32_t* heads, const int32_t* np, int32_t e, const float* scale,
                                float* const* dW, float* const* dE, mivp_stream_t stream);
int mivp_prompt_kv_fwd_multi(int32_t n, const MivpSwinDesc* d, const float* const* prompt, const float* const* ln_w,
                             const float* const* ln_b, const void* const* wqkv, const float* const* ts, void* const* kp,
                             void* const* vp, void* const* ka, mivp_stream_t stream);

/* relative-position bias as MFMA augmentation dims (relative_positional_encoding.py:99-142)
 *   t_h [heads][2*w0-1], t_w, t_d: per-axis relative tables  T_a[h][j-i+w_a-1] = s/3 * W_a[h] . E_a[...]
 *   (scale embed_dim**-0.5 and the /3 already applied by the host), ts [heads][Np] f32 prompt-token
 *   scores (scaled), or NULL when Np == 0.
 *   Because the bias is a sum of three terms that each depend on ONE slot coordinate of the query, it
 *   equals <onehot(query coords), table values at the key coords>: the kernel emits that as extra K
 *   columns of the QK^T MFMA:   qa [Nqp][augp] bf16 (query one-hots),  ka [heads][Nkp][augp] bf16 (x log2(e)).
 *   Padding key rows (Nq..Nqp-1 and the unused prompt rows) carry -30000 in the w0 query-h columns: every valid query
 *   sees that logit there, P underflows to exactly 0 and the attention kernels never test for padding keys.      */
int mivp_relbias_aug(const MivpSwinDesc* d, const float* t_h, const float* t_w, const float* t_d,
                     const float* ts, void* qa, void* ka, mivp_stream_t stream);

/* softmax((q k^T + bias) * mask) v per (window, head)  (window_attention.py:49-59)
 *   o [B*P][Nqp][C] bf16 (heads merged, channel = head*hd + j) ; lse [B*P][heads][Nqp] f32 (natural log; may be NULL
 *   when no backward pass will follow -- ABI 11)
 *   tok_rid [P][Nqp] int32: shift-mask region id of every window slot, 0 <= id < 254 (27 regions in 3D)
 *   mask_words / cut_flags (ABI 12, shifted blocks only; both may be NULL: the kernel then compares tok_rid classes per
 *   logit): the shift mask of swin_block.py:187-200,312-364 precomputed per window geometry as lane masks.
 *   mask_words [P][Nqp/16 query tiles][Nqp/16 key tiles][4] uint64: bit 16 g + r of word j of (window pw, query tile qt, key
 *   tile kt) is set when the logit of query slot 16 qt + r and key slot 16 kt + 4 g + j SURVIVES the multiplicative mask
 *   (same region id; key rows >= Nq always survive -- they are excluded by their bias); query rows >= Nq carry class 0.
 *   cut_flags [P] uint8: 1 when the window's content slots hold more than one region id (the others skip the mask).
 *   The kernels load the words with scalar loads and apply each as ONE v_cndmask per logit.                         */
int mivp_win_attn_fwd(const MivpSwinDesc* d, const void* q, const void* k, const void* v,
                      const void* kp, const void* vp, const void* qa, const void* ka,
                      const int32_t* tok_rid, void* o, float* lse, const uint64_t* mask_words,
                      const uint8_t* cut_flags, mivp_stream_t stream);

/* proj + residual, drop prompts, LayerNorm + Linear + residual, scatter + crop
 * (window_attention.py:60, swin_block.py:221-253)
 *   t1 [B*P][Nqp][C] bf16 (saved for backward, may be NULL) ; y [B, vol_out, C] bf16
 *   wproj, wmlp: weight fragment images (see "Weight fragment images" above) */
int mivp_swin_proj_mlp_fwd(const MivpSwinDesc* d, const void* o, const void* x,
                           const int32_t* tok_src, const int32_t* tok_dst,
                           const void* wproj, const float* bproj, const float* ln_w, const float* ln_b,
                           const void* wmlp, const float* bmlp, void* t1, void* y, mivp_stream_t stream);

/* ---- backward of the block: data gradients + prompt / token-bias gradients ---- */
/* dy [B, vol_out, C] bf16 -> dO [B*P][Nqp][C] bf16 and dt1 [B*P][Nqp][C] bf16
 *   wmlp_t, wproj_t = fragment images of mlp.weight^T, proj.weight^T  ([in][out] -> rows are input channels)
 *   weight-gradient mode (both or neither, else NULL): dn_out [B*P][Nqp][C] bf16 = gradient w.r.t. the
 *   mlp_norm output, dyw [B*P][Nqp][C] bf16 = dy in window order (zero rows where the token was cropped) */
int mivp_swin_proj_mlp_bwd(const MivpSwinDesc* d, const void* dy, const int32_t* tok_dst, const void* t1,
                           const float* ln_w, const float* ln_b, const void* wmlp_t, const void* wproj_t,
                           void* d_o, void* d_t1, void* dn_out, void* dyw, void* d_pj, mivp_stream_t stream);
/*   d_pj (NULL unless proj dropout is on in weight-gradient mode): dt1 with the proj-dropout mask applied,
 *   i.e. the gradient w.r.t. the proj output (A operand of the proj weight gradient) */

/* delta[bp][head][n] = sum_j dO * O  (flash-attention backward row term) */
int mivp_win_attn_delta(const MivpSwinDesc* d, const void* o, const void* d_o, float* delta,
                        mivp_stream_t stream);

/* EXPERIMENTAL E4M3 variant of mivp_win_attn_fwd for head_dim < 16 without dropout (BASELINE.json configs[4]): same
 * operands and outputs; Q', K', V and P are converted to fp8 inside the kernel (csrc/swin_fwd_fp8.hip).  Measured against the
 * bf16 kernel in profiles/r02_fp8_attention.json; not used by the model unless mivp_amd.swin_ops.USE_FP8_ATTN_FWD is set. */
int mivp_win_attn_fwd_fp8(const MivpSwinDesc* d, const void* q, const void* k, const void* v,
                          const void* kp, const void* vp, const void* qa, const void* ka,
                          const int32_t* tok_rid, void* o, float* lse, mivp_stream_t stream);

/* query-owner pass: dq [B*P][heads][Nqp][hd] bf16 (w.r.t. the stored, pre-scaled q).
 * Also WRITES delta [B*P][heads][Nqp] (= mivp_win_attn_delta) for the key-owner pass that follows. */
int mivp_win_attn_bwd_dq(const MivpSwinDesc* d, const void* q, const void* k, const void* v,
                         const void* kp, const void* vp, const void* qa, const void* ka,
                         const int32_t* tok_rid, const void* o, const void* d_o, const float* lse, float* delta,
                         void* dq, mivp_stream_t stream);

/* ONE-PASS attention backward for head_dim <= 16 (every encoder stage, the last decoder stage): dq, dk, dv and the prompt
 * partials of the two passes above from a single launch that forms S, dP and the exponentials once per (query, key) pair
 * (csrc/swin_bwd_fused.hip; autograd of window_attention.py:49-61 with the prompt keys of swin_block.py:187-225).
 * Same operand layouts as mivp_win_attn_bwd_dq / _dkv; delta is computed internally.  mivp_win_attn_bwd_fused_supported
 * returns 1 when the shape is covered (otherwise use the two-pass entries). */
int mivp_win_attn_bwd_fused_supported(const MivpSwinDesc* d);
int mivp_win_attn_bwd_fused(const MivpSwinDesc* d, const void* q, const void* k, const void* v,
                            const void* kp, const void* vp, const void* qa, const void* ka,
                            const int32_t* tok_rid, const void* o, const void* d_o, const float* lse,
                            void* dq, void* dk, void* dv, float* dkp_part, float* dvp_part, float* dtok_part,
                            mivp_stream_t stream);

/* Prompt-only attention backward for the first prompted block behind a frozen stem (no data gradient there): the prompt
 * keys' per-window partials dkp_part / dvp_part / dtok_part exactly as mivp_win_attn_bwd_dkv writes them, from one launch
 * that also forms delta (head_dim <= 16, Npp / 16 in {1, 2, 4, 8}; mivp_win_attn_bwd_prompt_supported tells). */
int mivp_win_attn_bwd_prompt_supported(const MivpSwinDesc* d);
int mivp_win_attn_bwd_prompt(const MivpSwinDesc* d, const void* q, const void* kp, const void* vp, const void* qa,
                             const void* ka, const void* o, const void* d_o, const float* lse, float* dkp_part,
                             float* dvp_part, float* dtok_part, mivp_stream_t stream);

/* key-owner pass: dk, dv [B*P][heads][Nqp][hd] bf16 for window keys;
 *   dkp_part, dvp_part [B*P][heads][Npp][hd] f32 and dtok_part [B*P][heads][Npp] f32:
 *   per-window partial sums for the prompt keys (reduced by mivp_reduce_rows)            */
int mivp_win_attn_bwd_dkv(const MivpSwinDesc* d, const void* q, const void* k, const void* v,
                          const void* kp, const void* vp, const void* qa, const void* ka,
                          const int32_t* tok_rid, const void* d_o, const float* lse, const float* delta,
                          void* dk, void* dv, float* dkp_part, float* dvp_part, float* dtok_part,
                          float* dka_part, mivp_stream_t stream);
/*   dka_part (NULL or f32 [B*P][heads][Nkp][32]): per-window gradient of the key-side bias augmentation
 *   columns (mivp_relbias_aug's ka); summed over windows and folded by mivp_relbias_grad into the
 *   gradients of the three relative-position tables (relative_positional_encoding.py:99-142)       */
int mivp_relbias_grad(const MivpSwinDesc* d, const float* dka /* [heads][Nkp][32] */, float* d_th, float* d_tw,
                      float* d_td, mivp_stream_t stream);

/* dq,dk,dv -> (x W^T backward) -> LayerNorm backward -> + dt1 -> scatter to dx [B, vol_in, C] bf16
 *   wqkv_t = fragment image of the stacked weight transposed [C][3C]; q part is multiplied by q_scale inside
 *   dn_out (NULL or [B*P][Nqp][C] bf16): gradient w.r.t. the attn_norm output (weight-gradient mode) */
int mivp_swin_qkv_bwd(const MivpSwinDesc* d, const void* dq, const void* dk, const void* dv,
                      const void* x, const int32_t* tok_src, const float* ln_w, const float* ln_b,
                      const void* wqkv_t, const void* d_t1, void* dx, void* dn_out, mivp_stream_t stream);

/* prompt K/V gradients -> through to_k/to_v and LayerNorm -> dprompt [Np][C] f32
 *   dkp, dvp [heads][Npp][hd] f32 (already reduced over windows) ; wqkv [3C][C] bf16
 *   weight-gradient mode (all three or NULL): wg_a [2][Np][C] bf16 = dK rows then dV rows (head-merged),
 *   wg_n [Np][C] bf16 = attn_norm(prompt), wg_ln [2][Np][C] f32 = per-row dbeta terms then dgamma terms */
int mivp_prompt_kv_bwd(const MivpSwinDesc* d, const float* dkp, const float* dvp, const float* prompt,
                       const float* ln_w, const float* ln_b, const void* wqkv, float* dprompt,
                       void* wg_a, void* wg_n, float* wg_ln, mivp_stream_t stream);

/* test hook: the keep masks the kernels derive from the descriptor's dropout fields, 1 = kept.
 *   attn_keep u8 [B*P*heads][Nqp][Nkp] (NULL to skip), proj_keep u8 [B*P*Nqp][C] (NULL to skip) */
int mivp_dropout_masks(const MivpSwinDesc* d, uint8_t* attn_keep, uint8_t* proj_keep, mivp_stream_t stream);

/* out[r] = sum_i in[i][r]  for i < n, r < rows  (deterministic two-level tree) */
int mivp_reduce_rows(const float* in, int64_t n, int64_t rows, float* out, mivp_stream_t stream);
/* the same for up to four arrays that share n (the prompt-gradient partials of one block), one launch:
 * out[i][r] = sum_k in[i][k*rows[i] + r];  in / rows / out are HOST arrays of nseg entries */
int mivp_reduce_rows_multi(int32_t nseg, const float* const* in, const int64_t* rows, float* const* out, int64_t n,
                           mivp_stream_t stream);


/* ------------------------------------------------------------------------ */
/* Patch merging (swin_transformer/down.py:21-53)                           */
/* ------------------------------------------------------------------------ */
typedef struct MivpMergeDesc {
    int32_t B, C;               /* input channels (multiple of 8)                      */
    int32_t dims[3];            /* input H, W, D                                       */
    int32_t odims[3];           /* output dims                                         */
    int32_t merge_last;         /* 1: 2x2x2 -> 8C ; 0: 2x2x1 -> 4C                     */
    int32_t Cout;               /* output channels                                     */
    float   ln_eps;
} MivpMergeDesc;
/* x [B,H,W,D,C] bf16 -> y [B,oh,ow,od,Cout] bf16 ; w [Cout][kC] bf16 ; ln over kC */
int mivp_patch_merge_fwd(const MivpMergeDesc* d, const void* x, const float* ln_w, const float* ln_b,
                         const void* w, void* y, mivp_stream_t stream);
/* dy -> dx ; w_t [kC][Cout] bf16; yfwd = the forward output [T][Cout] bf16; wgam[n] = sum_c W[n][c]*gamma[c] and
 * wbet[n] = sum_c W[n][c]*beta[c] (f32 [Cout], W = the bf16 weight): the LayerNorm-backward row sums follow from
 * dy, yfwd and these two vectors without a second GEMM.
 *   weight-gradient mode (both or NULL): wg_dn, wg_x [T][kC] bf16 = gradient w.r.t. the LayerNorm output and the
 *   gathered LayerNorm input rows */
int mivp_patch_merge_bwd(const MivpMergeDesc* d, const void* dy, const void* x, const void* yfwd, const float* ln_w,
                         const float* ln_b, const float* wgam, const float* wbet, const void* w_t, void* dx,
                         void* wg_dn, void* wg_x, mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* 3x3x3 stride-1 pad-1 convolution as implicit GEMM                         */
/* (swin_unetr.py:87,229-237,260-266; unet_blocks.py:46-56,74)              */
/* ------------------------------------------------------------------------ */
typedef struct MivpConvDesc {
    int32_t B;
    int32_t dims[3];            /* H, W, D (same for input and output)                 */
    int32_t Cin;                /* multiple of 8                                       */
    int32_t Cout;               /* real output channels                                */
    int32_t Kp;                 /* 27*Cin rounded up to 32 = row length of w           */
    int32_t pro_affine;         /* 1: x' = x*scale[ci] + shift[ci] on in-bounds voxels */
    int32_t pro_lrelu;          /* 1: then LeakyReLU(0.01)                             */
    int32_t add_residual;       /* 1: y += residual (bf16, same shape as y)            */
    int32_t out_f32;            /* 1: y is fp32 [B,vol,Cout], else bf16                */
} MivpConvDesc;
/* x [B,H,W,D,Cin] bf16 ; w [Cout_p][Kp] bf16 with k = tap*Cin + ci, tap = (kh*3+kw)*3+kd,
 * rows >= Cout and columns >= 27*Cin zero (Cout_p = Cout rounded up to 16) ; bias [Cout] f32 or NULL */
int mivp_conv3d_fwd(const MivpConvDesc* d, const void* x, const void* w, const float* bias,
                    const float* scale, const float* shift, const void* residual, void* y,
                    void* workspace, size_t ws_bytes, mivp_stream_t stream);
/* bytes of f32 split-K workspace the call above wants for this shape (0 = none).  Few voxels with a very
 * long K (bottleneck, first decoder stages) cannot fill 256 CUs by voxel tiles alone: K is cut into slices
 * that write f32 partials, summed in a fixed order by an epilogue kernel.  Passing NULL runs unsplit. */
size_t mivp_conv3d_fwd_ws(const MivpConvDesc* d);

/* The same convolution for LARGE volumes with few output channels (conv_concat of the last decoder stage,
 * unet_blocks.py:46-56,74): a workgroup stages the 6x10x18 (6x6x18) input halo of a 4x8x16 (4x4x16) output brick once per
 * 16-channel chunk in LDS instead of fetching every voxel once per tap.
 *   supported: bf16 output, Cin % 16 == 0, Cout % 4 == 0, Cout <= 48 or a multiple of 48
 *   wh: bf16 [groups][Cin/16][14][BN][32]: groups of 48 output channels (BN = 48, or Cout rounded to 16 when there is
 *       one group), k-step j = taps (2j, 2j+1) x 16 channels of the chunk, tap 27 = zeros */
int mivp_conv3d_halo_supported(const MivpConvDesc* d);
int mivp_conv3d_halo_fwd(const MivpConvDesc* d, const void* x, const void* wh, const float* bias,
                         const float* scale, const float* shift /* prologue, NULL unless d->pro_affine */,
                         const void* residual /* NULL unless d->add_residual */, void* y, int32_t brick_w /* 8: 4x8x16 bricks, 8 waves; 4: 4x4x16 bricks, 4 waves; 6: 6x6x16 bricks, 12 waves; 66: 6x6x8 and 36: 3x6x8 bricks of 2x8-voxel tiles (6 / 3 waves) */, mivp_stream_t stream);
/* Segmentation-head forward (swin_unetr.py:229-237): y = conv3x3x3(x * scale + shift) + bias for 27*Cout <= 64
 * (Cout <= 2), Cin + 1 <= 64.  x [B,H,W,D,Cin] bf16 (pre-BatchNorm), w f32 [Cout][Cin][3][3][3] (the nn.Conv3d
 * weight as stored), scale/shift f32 [Cin] (BatchNorm as an affine), y f32 [B,H,W,D,Cout].
 * Per-voxel GEMM to the 27*Cout tap outputs on MFMA + 27-point gather through LDS (csrc/head.hip). */
size_t mivp_head_conv_ws(void);            /* bytes of workspace (the folded bf16 weight tile) */
int mivp_head_conv_fwd(const MivpConvDesc* d, const void* x, const float* w, const float* bias,
                       const float* scale, const float* shift, void* workspace, float* y, mivp_stream_t stream);

/* weight + bias gradient for small Cout (segmentation heads, Cout <= 8):
 *   dwdb [Cout*27*Cin + Cout] f32 : dw[co][tap][ci] = sum_v dy[v][co] * x'[v+tap][ci] followed by db[co]
 *   (x' = the operand the forward conv saw, i.e. after the fused scale/shift/activation);
 *   dy [B,vol,dy_stride] bf16 with dy_stride >= Cout channels per voxel ; per-workgroup partial sums
 *   land in part[] (mivp_conv3d_wgrad_small_ws(d) floats) and are reduced deterministically.        */
size_t mivp_conv3d_wgrad_small_ws(const MivpConvDesc* d);
int mivp_conv3d_wgrad_small(const MivpConvDesc* d, const void* x, const float* scale, const float* shift,
                            const void* dy, int32_t dy_stride, float* part, float* dwdb,
                            mivp_stream_t stream);

/* MFMA form of the same gradient (what the product uses): gs [3][80][64] f32 with, for the tap
 * (sh, sw, sd) in {-1,0,1}^3, nb = (sh+1)*3 + (sw+1):
 *   gs[sd+1][nb*8 + co][c]   = sum_u dy[u - tap][co] * x[u][c]          (raw x, no affine)   for c < Cin
 *   gs[sd+1][nb*8 + co][Cin] = sum_{u in bounds} dy[u - tap][co]
 * dy carries 8 channels (dy_stride == 8, zero padded), Cout <= 8, Cin % 8 == 0, Cin < 64.  The conv weight
 * gradient, its bias gradient and the gradients of a BatchNorm fused in front of the conv are linear
 * functions of gs.  `_ws` = number of f32 partial values the call needs in `part`. */
size_t mivp_conv3d_wgrad_rows_ws(const MivpConvDesc* d);
int mivp_conv3d_wgrad_rows(const MivpConvDesc* d, const void* x, const void* dy, int32_t dy_stride,
                           float* part, float* gs, mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Patch embedding + BatchNorm statistics (swin_unetr.py:148-158)            */
/* ------------------------------------------------------------------------ */
typedef struct MivpEmbedDesc {
    int32_t B, Cin;
    int32_t dims[3];            /* input H, W, D (fp32, channels-first [B,Cin,H,W,D])  */
    int32_t C;                  /* output channels, multiple of 8                      */
    int32_t nblk;               /* stats blocks: rows of `part`                        */
} MivpEmbedDesc;
/* mode 0: accumulate per-channel sum / sum-of-squares of conv(x)+bias into part [nblk][2C] f32
 * mode 1: y = (conv(x)+bias) * scale + shift -> bf16 [B,h,w,d,C]                        */
int mivp_patch_embed(const MivpEmbedDesc* d, int mode, const float* x, const float* w, const float* bias,
                     const float* scale, const float* shift, float* part, void* y, mivp_stream_t stream);
/* bf16 im2col of the 2x2x2 patches, p [B*(H/2)*(W/2)*(D/2)][Cin*8] with columns in nn.Conv3d.weight order:
 * the input-side operand of mivp_gemm_tn for the patch-embedding weight gradient */
int mivp_patch_im2col(const MivpEmbedDesc* d, const float* x, void* p, mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* BatchNorm3d, training mode (batch statistics), channels-last bf16         */
/* ------------------------------------------------------------------------ */
/* per-channel partial sums of a bf16 [n_vox][C] tensor -> part [nblk][2C] f32 */
int mivp_bn_stats(const void* x, int64_t n_vox, int32_t C, int32_t nblk, float* part, mivp_stream_t stream);
/* part -> mean/rstd -> scale = w*rstd, shift = b - mean*scale ; running stats updated in place
 * (momentum, unbiased variance) ; mean_rstd [2C] f32 saved for backward                 */
int mivp_bn_finalize(const float* part, int32_t nblk, int32_t C, double count, const float* w, const float* b,
                     float eps, float momentum, float* running_mean, float* running_var,
                     float* scale, float* shift, float* mean_rstd, mivp_stream_t stream);
/* y = act(x*scale + shift), bf16 -> bf16 (used where the consumer cannot fuse the affine) */
int mivp_affine_act(const void* x, int64_t n_vox, int32_t C, const float* scale, const float* shift,
                    int32_t lrelu, void* y, mivp_stream_t stream);
/* backward reductions: part [nblk][2C] : sum(dy), sum(dy * xhat) over voxels, where
 * dy is taken AFTER the activation derivative when lrelu=1 (z = x*scale+shift, dy *= z>0 ? 1 : 0.01) */
int mivp_bn_bwd_stats(const void* x, const void* dy, int64_t n_vox, int32_t C, const float* scale,
                      const float* shift, const float* mean_rstd, int32_t lrelu, int32_t nblk, float* part,
                      mivp_stream_t stream);
/* dx = scale * (dyz - sum_dy/n - xhat * sum_dy_xhat/n) ; sums [2C] f32 (already reduced) */
int mivp_bn_bwd_apply(const void* x, const void* dy, int64_t n_vox, int32_t C, const float* scale,
                      const float* shift, const float* mean_rstd, const float* sums, int32_t lrelu,
                      void* dx, mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Trilinear upsample (align_corners=False) + crop + concat                  */
/* (unet_blocks.py:31-35,72-73; swin_unetr.py:351-355)                       */
/* ------------------------------------------------------------------------ */
typedef struct MivpUpcatDesc {
    int32_t B;
    int32_t idims[3];           /* low-res input dims                                  */
    int32_t odims[3];           /* output dims (= skip dims; <= scale*idims)            */
    int32_t scale[3];           /* 1 or 2 per axis                                     */
    int32_t Cx;                 /* channels of the low-res tensor                      */
    int32_t Cs;                 /* channels of the skip tensor (0 = no concat)         */
    int32_t align_corners;      /* 0: half-pixel centres (output_layer, SwinUpBlock); 1: align_corners=True  */
                                /*    (the reconstruction head's nn.Upsample, swin_unetr.py:199-201)          */
} MivpUpcatDesc;
/* x [B,ih,iw,id,Cx] bf16, skip [B,oh,ow,od,Cs] bf16 -> y [B,oh,ow,od,Cx+Cs] bf16 */
int mivp_upcat_fwd(const MivpUpcatDesc* d, const void* x, const void* skip, void* y, mivp_stream_t stream);
/* ABI 11: the concat tensor of SwinUpBlock (unet_blocks.py:72-75: up -> cat -> norm_concat -> act -> conv_concat) is
 * never stored un-normalised when nothing upstream needs a gradient:
 *   mivp_upcat_stats       per-channel sum / sum of squares of the bf16-rounded upsample + concat values ->
 *                          part [nblk][2*(Cx+Cs)] f32 in mivp_bn_stats' layout (mivp_bn_finalize reduces the nblk rows);
 *                          nblk = nx*B*oh with 1 <= nx <= ow (nx workgroups walk one (b, oh) line of output rows), Cx/8 <= 192 (256 without skip), Cs/8 <= 64, id*Cx/8 <= 2048
 *   mivp_upcat_affine_fwd  y = act(scale[c] * v + shift[c]) of the bf16-rounded concat value v (lrelu: slope 0.01),
 *                          bit-identical to mivp_upcat_fwd followed by mivp_affine_act                                  */
int mivp_upcat_stats(const MivpUpcatDesc* d, const void* x, const void* skip, int32_t nblk, float* part,
                     mivp_stream_t stream);
int mivp_upcat_affine_fwd(const MivpUpcatDesc* d, const void* x, const void* skip, const float* scale,
                          const float* shift, int32_t lrelu, void* y, mivp_stream_t stream);
/* dy [B,oh,ow,od,Cx+Cs] bf16 -> dx [B,ih,iw,id,Cx] bf16 (transposed stencil), dskip (slice copy; may be NULL) */
int mivp_upcat_bwd(const MivpUpcatDesc* d, const void* dy, void* dx, void* dskip, mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Dice + focal loss of the downstream step (modules/segmentation.py:44-50)  */
/* ------------------------------------------------------------------------ */
/* logits f32 [B, vol, C] channels-last (C <= 8), target f32 [B, vol] class indices.
 * loss[0] = mean_bc dice_bc + mean focal (MONAI DiceFocalLoss(include_background, to_onehot_y, softmax,
 * gamma) as documented; class 0 dropped when include_background == 0); dlogits f32 [B, vol, C] = d loss / d logits.
 * workspace: mivp_dice_focal_ws(B, vol) floats.
 * dlogits may be NULL (ABI 10): the value pass alone; the gradient pass then runs when autograd asks for it,
 * mivp_dice_focal_grad on the SAME workspace (it holds the per-sample Dice sums), scaled by gscale[0] (device pointer to
 * the incoming d total / d loss, NULL = 1) inside the pass instead of a second sweep over the 28 MB gradient.
 * gamma < 0 (ABI 12): the Dice term ALONE -- MONAI DiceLoss(include_background, to_onehot_y, softmax) of the students /
 * teacher trainer's supervised modes (modules/students_teacher.py:96-100,190-197): loss[0] = mean_bc dice_bc, no sigmoid /
 * focal work in either pass.                                                                                          */
size_t mivp_dice_focal_ws(int32_t B, int64_t vol);
int mivp_dice_focal(const float* logits, const float* target, int32_t B, int64_t vol, int32_t C,
                    int32_t include_background, float gamma, float* workspace, float* loss, float* dlogits,
                    mivp_stream_t stream);
int mivp_dice_focal_grad(const float* logits, const float* target, int32_t B, int64_t vol, int32_t C,
                         int32_t include_background, float gamma, float* workspace, const float* gscale, float* dlogits,
                         mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* downstream head on the LOW-resolution decoder output                      */
/* (swin_unetr.py:351-355 nn.Upsample x2 trilinear, then :229-237 BatchNorm3d */
/*  -> Conv3d 3^3): upsample, statistics, conv and the head's parameter       */
/*  gradients are all linear in x [B,h,w,d,C] bf16, so the C-channel          */
/*  full-resolution tensor is never formed.  C % 8 == 0, C < 64, Cout <= 4.   */
/* ------------------------------------------------------------------------ */
/* BatchNorm statistics of upsample(x): part [mivp_uphead_nblk(...)][2C] f32 partial (sum | sum of squares),   */
/* reduce with mivp_bn_finalize exactly like mivp_bn_stats' output (count = 8*B*h*w*d)                         */
int mivp_uphead_nblk(int32_t B, int32_t h, int32_t w, int32_t d, int32_t C);
/* gx (may be NULL): f32 [B*h*w*d][C] = U^T U x, the 27-point Gram stencil of the upsample applied to x -- the   */
/* per-cell term mivp_uphead_dx needs for the BatchNorm backward; keeping it saves that kernel a 27-point gather  */
int mivp_uphead_stats(const void* x, int32_t B, int32_t h, int32_t w, int32_t d, int32_t C, float* part, float* gx,
                      mivp_stream_t stream);
/* y [B,2h,2w,2d,Cout] f32 = bias + conv3x3x3(affine(upsample(x))).  wf: bf16 [16*ceil(27*Cout/16)][64], row
 * tap*Cout + co = (weight[co][:, tap] * scale | sum_c weight[co][c, tap] * shift[c] | 0...): the BatchNorm affine
 * folded into the weights, column C multiplying a constant-one channel.  workspace: mivp_uphead_fwd_ws bytes. */
size_t mivp_uphead_fwd_ws(int32_t B, int32_t h, int32_t w, int32_t d, int32_t Cout);
int mivp_uphead_fwd(const void* x, const void* wf, const float* bias, int32_t B, int32_t h, int32_t w, int32_t d,
                    int32_t C, int32_t Cout, void* workspace, float* y, mivp_stream_t stream);
/* D [B*h*w*d][ldD] bf16, column tap*Cout + co: the adjoint of the gather applied to dy [B,2h,2w,2d,dy_stride] f32.
 * mivp_gemm_tn(D, x) then gives G[tap*Cout + co][c] = sum_u dy[u - tap][co] * upsample(x)[u][c] and the column
 * sums of D give S = sum_{u in bounds} dy[u - tap][co]: the (G, S) of mivp_conv3d_wgrad_rows. */
int mivp_uphead_adjoint(const float* dy, int32_t dy_stride, int32_t B, int32_t h, int32_t w, int32_t d, int32_t Cout,
                        void* D, int32_t ldD, mivp_stream_t stream);
/* BatchNorm affine folded into the head conv (swin_unetr.py:229-237): wf bf16 [2][16*ceil(27*Cout/16)][64], row tap*Cout + co
 * = ( conv_w[co][c][tap] * scale[c]  for c < Cin | sum_c conv_w[co][c][tap] * shift[c] | 0 ... );  Cin < 64.  Plane 0 holds
 * the bf16 rounding of each value, plane 1 the bf16 rounding of the remainder (hi + lo pair, both fed to the MFMA). */
int mivp_uphead_fold(const float* conv_w, const float* scale, const float* shift, int32_t Cout, int32_t Cin, void* wf,
                     mivp_stream_t stream);

/* the four parameter gradients of the (BatchNorm -> conv 3^3) head from the (G, S) sums of mivp_uphead_adjoint +
 * mivp_gemm_tn (or mivp_conv3d_wgrad_rows): element (co, tap, ci) of G at G[co*gs_co + tap*gs_tap + ci], (co, tap) of S
 * at S[co*ss_co + tap*ss_tap];  dW [Cout][Cin][27], db [Cout], dgamma / dbeta [Cin] (autograd of swin_unetr.py:229-237) */
int mivp_head_grads(const float* G, int64_t gs_co, int64_t gs_tap, const float* S, int64_t ss_co, int64_t ss_tap,
                    const float* conv_w, const float* scale, const float* shift, const float* mean_rstd, int32_t Cout,
                    int32_t Cin, float* dW, float* db, float* dgamma, float* dbeta, mivp_stream_t stream);

/* the two small operands of mivp_uphead_dx from the head's parameters and batch statistics (layouts: see mivp_uphead_dx);
 * n_hr = number of high-resolution voxels 8*B*h*w*d; dgamma / dbeta may be NULL when training == 0 */
int mivp_uphead_dx_prep(const float* conv_w, const float* scale, const float* mean_rstd, const float* dgamma,
                        const float* dbeta, double n_hr, int32_t training, int32_t Cout, int32_t Cin, void* wc, float* coef,
                        mivp_stream_t stream);

/* gradient w.r.t. x [B,h,w,d,C] (bf16) through upsample -> BatchNorm -> conv, all at low resolution:
 *   D with ldD == 64; wc bf16 [16*ceil(C/16)][64] = conv weight as [c][tap*Cout + co] (zero padded);
 *   coef f32 [4][C] = (BN scale | sum(dz)/N | rstd*sum(dz*xhat)/N | batch mean), rows 1-2 zero for eval-mode BN;
 *   gx: mivp_uphead_stats' optional output for the same x, or NULL (the kernel then gathers the stencil itself) */
int mivp_uphead_dx(const void* D, const void* wc, const void* x, const float* coef, const float* gx, int32_t B, int32_t h,
                   int32_t w, int32_t d, int32_t C, void* dx, mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* weight gradients: out[M][N] (+)= alpha * sum_t A[t][m] * B[t][n]          */
/* (the dW of every nn.Linear / nn.Conv3d on the path when autograd reaches */
/*  it: `loss.backward()` in segmentation.py:104, students_teacher.py:176)   */
/* ------------------------------------------------------------------------ */
typedef struct MivpOperandDesc {   /* where element (t, c) of a bf16 token-major operand lives */
    int32_t mode;       /* 0: t*ld + c                                                          */
                        /* 1: head-split [T/rows][C/hd][rows][hd] (attention q/k/v gradients)   */
                        /* 2: conv tap: column c = tap*cin + ci reads voxel t displaced by the  */
                        /*    tap (kh-1, kw-1, kd-1), zero outside the volume                   */
    int32_t ld;         /* mode 0/2: elements per row, multiple of 4                             */
    int32_t rows, hd;   /* mode 1 (hd multiple of 4)                                            */
    int32_t dims[3];    /* mode 2: H, W, D; t = ((b*H + h)*W + w)*D + d                          */
    int32_t cin;        /* mode 2: channels per tap, multiple of 4                               */
} MivpOperandDesc;

typedef struct MivpGemmTnDesc {
    int64_t T;          /* tokens / voxels summed over                                          */
    int32_t M, N;       /* out is fp32 [M][N] row-major                                          */
    MivpOperandDesc a;  /* supplies the M index (gradient w.r.t. the layer output)              */
    MivpOperandDesc b;  /* supplies the N index (the layer input)                               */
    float   alpha;
    int32_t accumulate; /* 1: out += result                                                      */
    int32_t perm_cin;   /* > 0 (== b.cin, mode 2): store column tap*cin + ci at ci*27 + tap,     */
                        /* i.e. out is nn.Conv3d.weight's [Cout][Cin][3][3][3]                   */
} MivpGemmTnDesc;

/* LayerNorm parameter gradients and the normalised rows the Linear weight gradients multiply with
 * (nn.LayerNorm of swin_block.py:117-118, down.py:16): rows x[t] are read directly (tok_src NULL) or
 * gathered through the block's tok_src table; dn = gradient w.r.t. the LayerNorm output.
 *   stats  [T][2] f32 scratch;  n_out [T][C] bf16 = LN(x) (zero rows on padding slots);
 *   part   [nblk][2C] f32: per-block (dbeta | dgamma) partials, reduce with mivp_reduce_rows;
 *   (nblk*256) % (C/8) == 0 */
int mivp_ln_wgrad(const void* x, const int32_t* tok_src, const void* dn, int64_t T, int32_t C, int32_t Nqp,
                  int32_t P, int64_t vol, float eps, const float* gamma, const float* beta, float* stats,
                  void* n_out, int32_t nblk, float* part, mivp_stream_t stream);

size_t mivp_gemm_tn_ws(const MivpGemmTnDesc* d);   /* fp32 split partials */
int mivp_gemm_tn(const MivpGemmTnDesc* d, const void* a, const void* b, void* workspace, size_t ws_bytes,
                 float* out, mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* 1x1x1 convolution with <= 4 output channels (last layer of the            */
/* reconstruction head, swin_unetr.py:204-209)                               */
/* ------------------------------------------------------------------------ */
/* y [n_vox][Cout] f32 = bias + x [n_vox][C] bf16 . w [Cout][C] f32 */
int mivp_pointwise_fwd(const void* x, const float* w, const float* bias, int64_t n_vox, int32_t C, int32_t Cout,
                       float* y, mivp_stream_t stream);
/* dx [n_vox][C] bf16 = dy [n_vox][Cout] f32 . w;  dyb (optional) [n_vox][4] bf16 = dy zero-padded: the A operand of
 * mivp_gemm_tn for the weight gradient */
int mivp_pointwise_bwd(const float* dy, const float* w, int64_t n_vox, int32_t C, int32_t Cout, void* dx, void* dyb,
                       mivp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* small utilities                                                          */
/* ------------------------------------------------------------------------ */
/* fp32 -> bf16 cast of n elements (weight staging) */
int mivp_cast_f32_bf16(const float* in, int64_t n, void* out, mivp_stream_t stream);
/* y = a + b (bf16, n elements) : gradient accumulation at skip joins */
int mivp_add_bf16(const void* a, const void* b, int64_t n, void* y, mivp_stream_t stream);
/* MFMA lane-map self test: c[16][16] f32 = a[16][32] * b[16][32]^T via one 16x16x32 MFMA */
int mivp_selftest_mfma(const void* a, const void* b, float* c, mivp_stream_t stream);

/* ---- students/teacher objective and optimizer side ---------------------------------------------------
 * Point sampling of clustered_prototype_loss.py:162-204 (identity affine_grid + bilinear grid_sample, align_corners = False,
 * on an optional jitter crop): out f32 [B][o0*o1*o2][C] from vol [B][H][W][D][C] (channels_last) or [B][C][H][W][D], bf16 or
 * f32.  lo / hi / w: per axis, per output index: absolute voxel coordinates of the two taps and the weight of the upper one
 * (arrays of three device pointers).  The backward is a gather over voxels: per axis and voxel coordinate up to two
 * (output index, weight) pairs (index -1 = none); gvol bf16 [B][H][W][D][C], C % 8 == 0. */
int mivp_sample_points_fwd(const void* vol, int32_t is_bf16, int32_t channels_last, int32_t B, int32_t H, int32_t W, int32_t D,
                           int32_t C, const int32_t* out_dims, const int32_t* const* lo, const int32_t* const* hi,
                           const float* const* w, float* out, mivp_stream_t stream);
int mivp_sample_points_bwd(const float* gout, int32_t B, int32_t H, int32_t W, int32_t D, int32_t C, const int32_t* out_dims,
                           const int32_t* const* i1, const int32_t* const* i2, const float* const* w1, const float* const* w2,
                           void* gvol, mivp_stream_t stream);
/* torch.optim.AdamW over many tensors in one launch (students_teacher.py:27-68, segmentation.py:25-39).
 * tensors: device array of {float* p, float* exp_avg, float* exp_avg_sq, int64 n, int32 group, int32 pad} (40 bytes), static
 * across steps; grads: HOST array of n_tensors device pointers (they change every backward and travel as kernel arguments,
 * 384 per launch); chunk_begin: HOST int32 [n_tensors + 1], first chunk of each tensor; groups: HOST [n_groups][8] f32 =
 * {lr, beta1, beta2, eps, weight_decay, 1 - beta1^step, sqrt(1 - beta2^step), 0}; chunks: device int32 [.][2] = (tensor
 * index, 1024-element chunk), tensor-major. */
int mivp_adamw_multi(const void* tensors, const void* const* grads, int32_t n_tensors, const int32_t* chunk_begin,
                     const float* groups, int32_t n_groups, const void* chunks, mivp_stream_t stream);
/* The same step with the per-group hyper-parameters read from DEVICE memory (groups_dev: [8][8] f32, rows as above): the
 * form a HIP graph records, so that every replay uses the learning rate (WarmupCosineSchedule, modules/utils.py:67-89) and
 * bias corrections of its own step.  Bit-equal to mivp_adamw_multi for equal values. */
int mivp_adamw_multi_dev(const void* tensors, const void* const* grads, int32_t n_tensors, const int32_t* chunk_begin,
                         const float* groups_dev, int32_t n_groups, const void* chunks, mivp_stream_t stream);
/* n <= 64 floats from a HOST array into device memory as the arguments of a one-wave kernel (stream-ordered; refreshes the
 * scalars a recorded graph reads: the optimizer's groups_dev) */
int mivp_store_floats(float* dst, const float* host_values, int32_t n, mivp_stream_t stream);
/* EMA teacher update (momentum_model.py:27-36): tensors = device array of {float* teacher, const float* student, int64 n} */
int mivp_ema_multi(const void* tensors, const void* chunks, int32_t n_chunks, float tau, mivp_stream_t stream);
int mivp_sizeof_opt(int which);   /* 0: AdamW tensor record, 1: group record, 2: EMA record */
/* sizeof of the descriptor structs as the LIBRARY was compiled (bindings compare them with their own layout):
 * 0 MivpSwinDesc, 1 MivpMergeDesc, 2 MivpConvDesc, 3 MivpEmbedDesc, 4 MivpUpcatDesc, 5 MivpOperandDesc, 6 MivpGemmTnDesc */
int mivp_sizeof_desc(int which);

/* Per-class counts of arg-max(logits) against a label volume (modules/utils.py:14-64 MeanIoU / DiceCoefficient without
 * host round trips): counts int64 [C][3] += (|pred = c and target = c|, |pred = c|, |target = c|); logits f32 [B][vol][C]
 * (channels_last) or [B][C][vol], target f32 [B][vol] class indices, nvox = B * vol, C <= 16. */
int mivp_seg_counts(const float* logits, const float* target, int64_t nvox, int32_t C, int32_t channels_last, int64_t vol,
                    void* counts, mivp_stream_t stream);

/* Transposed convolution with kernel == stride in {1, 2} per axis, no bias (the up-sampling step of MONAI's UnetrUpBlock,
 * swin_unetr.py:338-348,372-380): x bf16 [B][h][w][d][Cin] -> y bf16 [B][h*s0][w*s1][d*s2][Cout].
 * w1 bf16 [taps*Cout][Cin] (row (tap, co), tap = (a*s1 + b)*s2 + c of the nn.ConvTranspose3d weight [Cin][Cout][a][b][c]);
 * the data gradient takes w2 bf16 [Cin][taps*Cout].  Cin % 8 == 0, Cout % 8 == 0.  (Weight gradient: mivp_gemm_tn.) */
int mivp_convt_fwd(int32_t B, const int32_t* dims, const int32_t* stride, int32_t Cin, int32_t Cout, const void* x, const void* w1,
                   void* y, mivp_stream_t stream);
int mivp_convt_dgrad(int32_t B, const int32_t* dims, const int32_t* stride, int32_t Cin, int32_t Cout, const void* dy,
                     const void* w2, void* dx, mivp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MIVP_H */
