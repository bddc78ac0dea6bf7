// Dice + (sigmoid) focal loss of the downstream step, forward value and gradient in two streaming passes
// over the logits.  Reference: modules/segmentation.py:44-50 constructs MONAI
// DiceFocalLoss(include_background, to_onehot_y=True, softmax=True, gamma=4.0); the formula is restated
// from MONAI's documentation (MONAI is absent from the image: parity unpinned, see oracle/loss_ref.py).
//   dice_bc = 1 - (2 I_bc + e) / (P_bc + T_bc + e),  I = sum p*t, P = sum p, T = sum t,  p = softmax_c(z)
//   focal   = mean over (b, c, voxel) of  exp(gamma * logsig(-z*(2t-1))) * (z - z*t - logsig(z))
//   loss    = mean_bc dice_bc + focal          (class 0 dropped from both when include_background == 0)
#include "common.hpp"

namespace {
constexpr int LOSS_MAXC = 8;
// partial rows per sample: the statistics pass runs B * bpb workgroups (4 x 256 = four per CU at batch 4) and the
// single-workgroup finalize reads bpb partials per sum -- with 1024 it was two load round trips per sum and 15 us
constexpr int LOSS_MAX_BPB = 256;
// log(1 + e) with e in (0, 1]: the hardware log is accurate to ~1 ulp of the RESULT's magnitude near 1, i.e. an absolute
// error <= 6e-8 -- far below the 1e-4 gradient tolerance -- and ~20x cheaper than log1pf.
// log_sigmoid(+-z) and sigmoid(+-z) share e = exp(-|z|), log(1 + e) and 1 / (1 + e): one exp, one log, one rcp per logit.
struct Sig { float ls_pos, ls_neg, sg_pos, sg_neg; };          // log_sigmoid(z), log_sigmoid(-z), sigmoid(z), sigmoid(-z)
MIVP_DEV Sig sig_all(float z) {
    const float e = __expf(-fabsf(z)), l = __logf(1.f + e), r = __builtin_amdgcn_rcpf(1.f + e);   // v_rcp_f32: 1 ulp
    Sig g;
    g.ls_pos = fminf(z, 0.f) - l;
    g.ls_neg = fminf(-z, 0.f) - l;
    g.sg_pos = z >= 0.f ? r : e * r;
    g.sg_neg = z >= 0.f ? e * r : r;
    return g;
}

// FOCAL == false: the Dice term alone (MONAI DiceLoss, students_teacher.py:96-100): no sigmoid / log work, focal sum stays 0
template <int C, bool FOCAL>
MIVP_DEV void voxel_stats(const float* zz, int cls, int c0, float gamma, float* acc) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, zz[c]);
    float den = 0.f, e[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { e[c] = __expf(zz[c] - mx); den += e[c]; }
    const float inv_den = __builtin_amdgcn_rcpf(den);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (c >= c0) {
            const float p = e[c] * inv_den, t = (c == cls) ? 1.f : 0.f;
            acc[3 * c] += p * t;
            acc[3 * c + 1] += p;
            acc[3 * c + 2] += t;
            if (FOCAL) {
                const Sig g = sig_all(zz[c]);
                const float bce = zz[c] - zz[c] * t - g.ls_pos;
                acc[3 * LOSS_MAXC] += __expf(gamma * (c == cls ? g.ls_neg : g.ls_pos)) * bce;     // log_sigmoid(-z (2t - 1))
            }
        }
    }
}

// per-(sample, class) constants of the dice gradient: d(dice mean)/dp_c = qa[c] - (c == cls) * qb[c]
template <int C>
MIVP_DEV void dice_consts(const float* __restrict__ st, int c0, float wd, float* qa, float* qb) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        qa[c] = 0.f; qb[c] = 0.f;
        if (c >= c0) {
            const float I = st[3 * c], P = st[3 * c + 1], T = st[3 * c + 2];
            const float Dn = P + T + 1e-5f, Nn = 2.f * I + 1e-5f;
            qa[c] = wd * (Nn / (Dn * Dn));
            qb[c] = wd * (2.f / Dn);
        }
    }
}

template <int C, bool FOCAL>
MIVP_DEV void voxel_grad(const float* zz, int cls, int c0, float gamma, float wf, const float* qa, const float* qb, float* gz) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, zz[c]);
    float den = 0.f, p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = __expf(zz[c] - mx); den += p[c]; }
    const float inv_den = __builtin_amdgcn_rcpf(den);
    float q[C], pq = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        p[c] *= inv_den;
        q[c] = qa[c] - ((c == cls) ? qb[c] : 0.f);
        pq += p[c] * q[c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float g = p[c] * (q[c] - pq);                                        // through the softmax
        if (FOCAL && c >= c0) {
            const float t = (c == cls) ? 1.f : 0.f, sg = 2.f * t - 1.f;
            const Sig sv = sig_all(zz[c]);
            const float w = __expf(gamma * (c == cls ? sv.ls_neg : sv.ls_pos));               // log_sigmoid(-z sg)
            const float bce = zz[c] - zz[c] * t - sv.ls_pos;
            g += wf * (w * (sv.sg_pos - t) - bce * gamma * w * sg * (c == cls ? sv.sg_pos : sv.sg_neg));
        }
        gz[c] = g;
    }
}
}

// pass 1: per-block partial sums  [3*C (I, P, T per class) + 1 (focal sum)]  for ONE batch element per block row
// VEC: four consecutive voxels per thread and iteration through 16-byte loads (vol % 4 == 0): the loads of one iteration
// are independent, which is what hides the HBM latency of this short grid-stride walk
template <int C, bool VEC, bool FOCAL>
__global__ __launch_bounds__(256) void k_dice_focal_stats(const float* __restrict__ z, const float* __restrict__ y, long vol,
                                                          int c0, float gamma, int blocks_per_b,
                                                          float* __restrict__ part) {
    __shared__ float red[4][3 * LOSS_MAXC + 1];
    const int b = blockIdx.x / blocks_per_b, blk = blockIdx.x % blocks_per_b;
    float acc[3 * LOSS_MAXC + 1];
#pragma unroll
    for (int i = 0; i < 3 * LOSS_MAXC + 1; ++i) acc[i] = 0.f;
    if (VEC) {
        const long nq = vol >> 2;
        for (long q = (long)blk * 256 + threadIdx.x; q < nq; q += (long)blocks_per_b * 256) {
            const long v = (long)b * vol + 4 * q;
            float zb[4 * C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const float4 t4 = *reinterpret_cast<const float4*>(z + v * C + 4 * j);
                zb[4 * j] = t4.x; zb[4 * j + 1] = t4.y; zb[4 * j + 2] = t4.z; zb[4 * j + 3] = t4.w;
            }
            const float4 y4 = *reinterpret_cast<const float4*>(y + v);
            voxel_stats<C, FOCAL>(zb, (int)y4.x, c0, gamma, acc);
            voxel_stats<C, FOCAL>(zb + C, (int)y4.y, c0, gamma, acc);
            voxel_stats<C, FOCAL>(zb + 2 * C, (int)y4.z, c0, gamma, acc);
            voxel_stats<C, FOCAL>(zb + 3 * C, (int)y4.w, c0, gamma, acc);
        }
    } else {
        for (long v = (long)blk * 256 + threadIdx.x; v < vol; v += (long)blocks_per_b * 256) {
            const float* zv = z + ((long)b * vol + v) * C;
            float zz[C];
#pragma unroll
            for (int c = 0; c < C; ++c) zz[c] = zv[c];
            voxel_stats<C, FOCAL>(zz, (int)y[(long)b * vol + v], c0, gamma, acc);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 3 * LOSS_MAXC + 1; ++i) {
        float vsum = 0.f;
        if (i < 3 * C || i == 3 * LOSS_MAXC) {                    // the other columns are never touched: they stay zero
            vsum = acc[i];
            for (int o = 32; o > 0; o >>= 1) vsum += __shfl_xor(vsum, o);
        }
        if (lane == 0) red[wave][i] = vsum;
    }
    __syncthreads();
    if (threadIdx.x < 3 * LOSS_MAXC + 1) {
        const int i = threadIdx.x;
        part[(long)blockIdx.x * (3 * LOSS_MAXC + 1) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

// sums [B][blocks_per_b][25] -> stats [B][25] (fixed order), loss value
__global__ void k_dice_focal_finalize(const float* __restrict__ part, int B, int blocks_per_b, long vol, int C, int c0,
                                      float* __restrict__ stats, float* __restrict__ loss) {
    const int K = 3 * LOSS_MAXC + 1;
    // one wave per (b, k) sum: lanes stride over the per-block partials, then a butterfly (fixed order)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int used = 3 * C + 1;                                   // columns 0 .. 3C-1 and the focal sum; the rest stay zero
    for (int i = wv; i < B * K; i += nwv) stats[i] = 0.f;
    __syncthreads();
    for (int ii = wv; ii < B * used; ii += nwv) {
        const int b = ii / used, kk = ii - b * used;
        const int k = kk < 3 * C ? kk : 3 * LOSS_MAXC;
        const int i = b * K + k;
        double s = 0.0;
        int j = lane;
        for (; j + 64 * 7 < blocks_per_b; j += 64 * 8) {         // eight independent loads per round trip, same order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[((long)b * blocks_per_b + j + 64 * u) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; j < blocks_per_b; j += 64) s += (double)part[((long)b * blocks_per_b + j) * K + k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) stats[i] = (float)s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float d = 0.f, f = 0.f;
        for (int b = 0; b < B; ++b) {
            for (int c = c0; c < C; ++c) {
                const float I = stats[b * K + 3 * c], P = stats[b * K + 3 * c + 1], T = stats[b * K + 3 * c + 2];
                d += 1.f - (2.f * I + 1e-5f) / (P + T + 1e-5f);
            }
            f += stats[b * K + 3 * LOSS_MAXC];
        }
        const float ncls = (float)(C - c0);
        loss[0] = d / ((float)B * ncls) + f / ((float)B * ncls * (float)vol);
    }
}

// pass 2: dz = dL/dz  (f32, same layout as z); VEC as in pass 1 (a group of four voxels never straddles two samples)
template <int C, bool VEC, bool FOCAL>
__global__ __launch_bounds__(256) void k_dice_focal_grad(const float* __restrict__ z, const float* __restrict__ y, long vol,
                                                         int B, int c0, float gamma, const float* __restrict__ stats,
                                                         const float* __restrict__ gscale, float* __restrict__ dz) {
    const int K = 3 * LOSS_MAXC + 1;
    const long total = (long)B * vol;
    const float up = gscale ? gscale[0] : 1.f;                 // the incoming d(total)/d(loss): folded in here, not a second pass
    const float ncls = (float)(C - c0);
    const float wd = 1.f / ((float)B * ncls), wf = 1.f / ((float)B * ncls * (float)vol);
    if (VEC) {
        const long nq = total >> 2;
        for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
            const long i = 4 * q;
            const int b = (int)(i / vol);
            float zb[4 * C], gb[4 * C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const float4 t4 = *reinterpret_cast<const float4*>(z + i * C + 4 * j);
                zb[4 * j] = t4.x; zb[4 * j + 1] = t4.y; zb[4 * j + 2] = t4.z; zb[4 * j + 3] = t4.w;
            }
            const float4 y4 = *reinterpret_cast<const float4*>(y + i);
            float qa[C], qb[C];
            dice_consts<C>(stats + b * K, c0, wd, qa, qb);
            voxel_grad<C, FOCAL>(zb, (int)y4.x, c0, gamma, wf, qa, qb, gb);
            voxel_grad<C, FOCAL>(zb + C, (int)y4.y, c0, gamma, wf, qa, qb, gb + C);
            voxel_grad<C, FOCAL>(zb + 2 * C, (int)y4.z, c0, gamma, wf, qa, qb, gb + 2 * C);
            voxel_grad<C, FOCAL>(zb + 3 * C, (int)y4.w, c0, gamma, wf, qa, qb, gb + 3 * C);
#pragma unroll
            for (int j = 0; j < C; ++j)
                *reinterpret_cast<float4*>(dz + i * C + 4 * j) = make_float4(up * gb[4 * j], up * gb[4 * j + 1], up * gb[4 * j + 2], up * gb[4 * j + 3]);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int b = (int)(i / vol);
            float zz[C], gz[C], qa[C], qb[C];
#pragma unroll
            for (int c = 0; c < C; ++c) zz[c] = z[i * C + c];
            dice_consts<C>(stats + b * K, c0, wd, qa, qb);
            voxel_grad<C, FOCAL>(zz, (int)y[i], c0, gamma, wf, qa, qb, gz);
#pragma unroll
            for (int c = 0; c < C; ++c) dz[i * C + c] = up * gz[c];
        }
    }
}

extern "C" size_t mivp_dice_focal_ws(int32_t B, int64_t vol) {
    long bpb = (vol + 256 * 4 - 1) / (256 * 4);
    if (bpb > LOSS_MAX_BPB) bpb = LOSS_MAX_BPB;
    if (bpb < 1) bpb = 1;
    return (size_t)B * (bpb + 1) * (3 * LOSS_MAXC + 1);
}

static int dice_focal_run(const float* logits, const float* target, int32_t B, int64_t vol, int32_t C, int32_t include_background,
                          float gamma, float* workspace, float* loss, const float* gscale, float* dlogits, bool do_loss,
                          mivp_stream_t stream) {
    const int K = 3 * LOSS_MAXC + 1;
    long bpb = (vol + 256 * 4 - 1) / (256 * 4);
    if (bpb > LOSS_MAX_BPB) bpb = LOSS_MAX_BPB;
    if (bpb < 1) bpb = 1;
    const int c0 = include_background ? 0 : 1;
    float* part = workspace;
    float* stats = workspace + (long)B * bpb * K;
    hipStream_t st = (hipStream_t)stream;
#define LOSS_C_SWITCH(LAUNCH)                                                                        \
    switch (C) {                                                                                     \
        case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; case 4: LAUNCH(4); break; case 5: LAUNCH(5); break; \
        case 6: LAUNCH(6); break; case 7: LAUNCH(7); break; default: LAUNCH(8); break;              \
    }
    const bool vec = vol % 4 == 0;                             // 16-byte loads of four voxels
    const bool focal = gamma >= 0.f;                           // gamma < 0: the Dice term alone (DiceLoss)
    if (do_loss) {
#define L_STATS1(CC, VV, FF) hipLaunchKernelGGL((k_dice_focal_stats<CC, VV, FF>), dim3((unsigned)(B * bpb)), dim3(256), 0, st, \
                                                logits, target, (long)vol, c0, gamma, (int)bpb, part)
#define L_STATS(CC) do { if (focal) { if (vec) L_STATS1(CC, true, true); else L_STATS1(CC, false, true); }                    \
                         else { if (vec) L_STATS1(CC, true, false); else L_STATS1(CC, false, false); } } while (0)
        LOSS_C_SWITCH(L_STATS)
#undef L_STATS
#undef L_STATS1
        int rc = mivp_check_launch("dice_focal_stats");
        if (rc) return rc;
        hipLaunchKernelGGL(k_dice_focal_finalize, dim3(1), dim3(1024), 0, st, part, (int)B, (int)bpb, (long)vol, (int)C, c0, stats, loss);
        rc = mivp_check_launch("dice_focal_finalize");
        if (rc) return rc;
    }
    if (!dlogits) return MIVP_OK;
    const long total = vec ? (long)B * vol / 4 : (long)B * vol;
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
#define L_GRAD1(CC, VV, FF) hipLaunchKernelGGL((k_dice_focal_grad<CC, VV, FF>), dim3(grid), dim3(256), 0, st, logits, target, \
                                               (long)vol, (int)B, c0, gamma, stats, gscale, dlogits)
#define L_GRAD(CC) do { if (focal) { if (vec) L_GRAD1(CC, true, true); else L_GRAD1(CC, false, true); }                        \
                        else { if (vec) L_GRAD1(CC, true, false); else L_GRAD1(CC, false, false); } } while (0)
    LOSS_C_SWITCH(L_GRAD)
#undef L_GRAD
#undef L_GRAD1
#undef LOSS_C_SWITCH
    return mivp_check_launch("dice_focal_grad");
}

extern "C" int mivp_dice_focal(const float* logits, const float* target, int32_t B, int64_t vol, int32_t C,
                               int32_t include_background, float gamma, float* workspace, float* loss, float* dlogits,
                               mivp_stream_t stream) {
    MIVP_REQUIRE(logits && target && workspace && loss);
    MIVP_REQUIRE(B > 0 && vol > 0 && C >= 2 && C <= LOSS_MAXC);
    return dice_focal_run(logits, target, B, vol, C, include_background, gamma, workspace, loss, nullptr, dlogits, true, stream);
}

extern "C" int mivp_dice_focal_grad(const float* logits, const float* target, int32_t B, int64_t vol, int32_t C,
                                    int32_t include_background, float gamma, float* workspace, const float* gscale,
                                    float* dlogits, mivp_stream_t stream) {
    MIVP_REQUIRE(logits && target && workspace && dlogits);
    MIVP_REQUIRE(B > 0 && vol > 0 && C >= 2 && C <= LOSS_MAXC);
    return dice_focal_run(logits, target, B, vol, C, include_background, gamma, workspace, nullptr, gscale, dlogits, false, stream);
}
