// The downstream segmentation head WITHOUT its 48-channel full-resolution input.
//
// Reference: swin_unetr/swin_unetr.py:351-355 (output_layer = nn.Upsample(x2, trilinear, align_corners=False)) followed
// by :229-237 (extra_heads['downstream'] = BatchNorm3d -> Conv3d 3^3, Cout = output_channels_downstream).  At 96^3 the
// upsampled tensor is 340 MB per batch of 4 and every stage around it (upsample, BN statistics, head conv, head weight
// gradient) is bound by streaming it.  All of them are linear in the LOW-resolution tensor x (U = the upsample matrix):
//
//   forward   y[u][co] = b[co] + sum_tap [u+tap inside] * (U Y_tap)[u+tap][co],    Y_tap[p][co] = sum_c W'[co][tap][c] x[p][c]
//             (W' = conv weight with the BatchNorm affine folded in; a constant-one channel carries the shift, it
//              upsamples to one and is masked with the tap at the zero-padded border exactly like the data)
//   backward  G[co][tap][c] = sum_u dy[u - tap][co] * (U x)[u][c] = sum_p x[p][c] * D_tap[p][co],
//             D_tap[p][co] = sum_u U[u + tap... ] (the adjoint of the gather above applied to dy)
//
// so the work at full resolution touches only Cout (= 2) channels:
//   k_uphead_stats    BatchNorm statistics of U x from x (interpolates on the fly, no store)
//   k_uphead_taps     Y = x W'^T on MFMA at low resolution -> fp16 planes [27][T_lr][Cout]
//   k_uphead_gather   y from the 27 planes: one thread per low-res cell = 8 output voxels, static 2-point stencils
//   k_uphead_adjoint  D (bf16 [T_lr][27*Cout]) from dy, separable accumulation; then mivp_gemm_tn(D, x) gives G
// Trilinear weights: source index max(0, (o + 0.5)/2 - 0.5), neighbours clamped (ATen area_pixel_compute_source_index).
#include "common.hpp"

namespace {

struct Lerp2 { int i0, i1; float w0, w1; };
MIVP_DEV Lerp2 lerp2(int o, int n_in) {                       // scale 2
    Lerp2 l;
    float src = ((float)o + 0.5f) * 0.5f - 0.5f;
    if (src < 0.f) src = 0.f;
    l.i0 = (int)src;
    l.i1 = l.i0 + (l.i0 < n_in - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

// ---------------------------------------------------------------------------------------------
// per-channel sum / sum of squares of the x2-upsampled tensor U x, from x alone.  Per axis the columns of U sum to 2
// and its Gram matrix U^T U is tridiagonal: diagonal 1.25 (1.625 at the two clamped ends, 2 if the axis has one
// cell), off-diagonal 0.375.  Hence
//     sum_u (Ux)[u]    = 8 * sum_p x[p]
//     sum_u (Ux)[u]^2  = sum_p sum_{m in {-1,0,1}^3, p+m inside} g0(p0,m0) g1(p1,m1) g2(p2,m2) * x[p] * x[p+m]
// i.e. a 27-point product stencil at LOW resolution (3x fewer multiply-adds than interpolating every output
// voxel, and no full-resolution pass at all).  Work item = (cell, 8-channel group); a thread keeps its channel
// group (total threads is a multiple of C/8), blocks reduce in a fixed order: part[block][2C].
// ---------------------------------------------------------------------------------------------
MIVP_DEV float gram_w(int p, int m, int n) {                 // (U^T U)[p][p+m] along one axis, 0 when p+m is outside
    if (m == 0) return n == 1 ? 2.0f : ((p == 0 || p == n - 1) ? 1.625f : 1.25f);
    return ((unsigned)(p + m) < (unsigned)n) ? 0.375f : 0.f;
}

// The stencil is separable: A[d] = sum_{m0,m1} g0 g1 x[p0+m0, p1+m1, d] (nine loads), then
// q[d] = g2(-1) A[d-1] + g2(0) A[d] + g2(+1) A[d+1].  A thread owns a (b, p0, p1, 8-channel group) line and slides
// along a segment of d keeping the last three A's: 9 loads per cell instead of 27.  Lines are cut into segments of
// ST_SEG cells for parallelism (the two A's beyond a segment's ends are recomputed).
constexpr int ST_SEG = 16;
__global__ __launch_bounds__(256) void k_uphead_stats(const bf16_t* __restrict__ x, int B, int h, int w, int d, int C,
                                                      float* __restrict__ part, float* __restrict__ gx) {
    __shared__ float lds[256 * 16];
    const int G = C / 8;
    const int nseg = (d + ST_SEG - 1) / ST_SEG;
    const long items = (long)B * h * w * nseg * G;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const int cg = (int)(gtid % G);
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    for (long it = gtid; it < items; it += stride) {
        unsigned rest = (unsigned)it / (unsigned)G;               // host checks items < 2^31: 32-bit decode
        unsigned nx = rest / (unsigned)nseg;
        const int seg = (int)(rest - nx * (unsigned)nseg);
        rest = nx; nx = rest / (unsigned)w;
        const int p1 = (int)(rest - nx * (unsigned)w);
        rest = nx; nx = rest / (unsigned)h;
        const int p0 = (int)(rest - nx * (unsigned)h);
        const long b = nx;
        const int lo = seg * ST_SEG, hi = (lo + ST_SEG) < d ? (lo + ST_SEG) : d;
        float g01[3][3];
        long roff[3][3];
#pragma unroll
        for (int m0 = 0; m0 < 3; ++m0)
#pragma unroll
            for (int m1 = 0; m1 < 3; ++m1) {
                g01[m0][m1] = gram_w(p0, m0 - 1, h) * gram_w(p1, m1 - 1, w);
                const int q0 = g01[m0][m1] != 0.f ? p0 + m0 - 1 : p0, q1 = g01[m0][m1] != 0.f ? p1 + m1 - 1 : p1;
                roff[m0][m1] = (((b * h + q0) * w + q1) * (long)d) * C + cg * 8;
            }
        float Am[8], Ac[8], xc[8];                 // A[dd-1], A[dd] and the centre value x[dd] while A[dd+1] is formed
#pragma unroll
        for (int i = 0; i < 8; ++i) { Am[i] = 0.f; Ac[i] = 0.f; xc[i] = 0.f; }
        for (int dd = lo - 1; dd <= hi; ++dd) {    // A[dd] for dd = lo-1 .. hi; cell dd-1 is completed when A[dd] arrives
            float An[8], xn[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { An[i] = 0.f; xn[i] = 0.f; }
            if (dd >= 0 && dd < d) {
#pragma unroll
                for (int m0 = 0; m0 < 3; ++m0)
#pragma unroll
                    for (int m1 = 0; m1 < 3; ++m1) {
                        const bf16x8 v = ld8(x + roff[m0][m1] + (long)dd * C);
#pragma unroll
                        for (int i = 0; i < 8; ++i) An[i] += g01[m0][m1] * (float)v[i];
                        if (m0 == 1 && m1 == 1) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) xn[i] = (float)v[i];
                        }
                    }
            }
            const int cell = dd - 1;               // its three A's are Am (cell-1), Ac (cell), An (cell+1)
            if (cell >= lo && cell < hi) {
                const float gm = gram_w(cell, -1, d), g0 = gram_w(cell, 0, d), gp = gram_w(cell, 1, d);
                float q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    q[i] = gm * Am[i] + g0 * Ac[i] + gp * An[i];     // (U^T U x)[cell]: the Gram stencil applied to x
                    s1[i] += 8.f * xc[i];
                    s2[i] += xc[i] * q[i];
                }
                if (gx) {                                            // kept for k_uphead_dx (BatchNorm backward needs it per cell)
                    float* o = gx + ((((long)b * h + p0) * w + p1) * d + cell) * C + cg * 8;
                    *reinterpret_cast<float4*>(o) = make_float4(q[0], q[1], q[2], q[3]);
                    *reinterpret_cast<float4*>(o + 4) = make_float4(q[4], q[5], q[6], q[7]);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { Am[i] = Ac[i]; Ac[i] = An[i]; xc[i] = xn[i]; }
        }
    }
    block_reduce_groups(lds, s1, s2, G, C, (long)blockIdx.x * 256, part + (long)blockIdx.x * 2 * C);
}

// ---------------------------------------------------------------------------------------------
// Y[tap][t][co] (fp16) = sum_c Wf[tap*Cout + co][c] * x[t][c] + Wf[..][C] * 1     (K = C + 1 <= 64, M = 27*Cout <= 16*MT)
// ---------------------------------------------------------------------------------------------
constexpr int TAPS_TILES = 8;                                   // 16-token tiles per wave: the folded weights stay in registers

template <int MT, int COUT>
__global__ __launch_bounds__(256) void k_uphead_taps(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wf, long T, int C,
                                                     _Float16* __restrict__ Y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    // folded weights as a bf16 hi + lo pair: a single bf16 rounding of w * scale is a systematic 2^-9 error per weight
    // (1.7e-3 of the logits, same sign for every voxel: it biased the arg-max near class boundaries).  Loaded ONCE per
    // wave and reused for TAPS_TILES token tiles (per-tile reloads made the kernel weight-fetch-bound: 16 KB per 16 tokens).
    bf16x8 wh[MT][2], wl[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            wh[mt][s] = ld8(wf + (long)(16 * mt + r) * 64 + 32 * s + 8 * g);
            wl[mt][s] = ld8(wf + (long)(16 * (MT + mt) + r) * 64 + 32 * s + 8 * g);
        }
    const long tile0 = ((long)blockIdx.x * 4 + wave) * TAPS_TILES;
#pragma unroll 2
    for (int it = 0; it < TAPS_TILES; ++it) {
        const long t = (tile0 + it) * 16 + r;
        const bool live = t < T;
        bf16x8 xb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = 32 * s + 8 * g;
            bf16x8 v = zero8();
            if (live) {
                if (c + 8 <= C) v = ld8(x + t * C + c);
                else if (c == C) v[0] = (bf16_t)1.0f;          // the constant-one channel (C % 8 == 0)
            }
            xb[s] = v;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 acc = fzero4();
#pragma unroll
            for (int s = 0; s < 2; ++s) { acc = mfma16(wh[mt][s], xb[s], acc); acc = mfma16(wl[mt][s], xb[s], acc); }
            if (live) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = 16 * mt + 4 * g + j;
                    if (m < 27 * COUT) Y[((long)(m / COUT) * T + t) * COUT + (m % COUT)] = (_Float16)acc[j];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gather: one thread per low-res cell p = 8 output voxels 2p + a.  Per axis, output i = 2p + a + k (k = tap - 1) reads
//   (a,k) = (0,-1): .75 N[-1] + .25 N[0]   (0,0): .25 N[-1] + .75 N[0]   (0,+1): .75 N[0] + .25 N[+1]
//           (1,-1): .25 N[-1] + .75 N[0]   (1,0): .75 N[0] + .25 N[+1]   (1,+1): .25 N[0] + .75 N[+1]
// of the CLAMPED neighbours N[m] = Y[clamp(p + m)], which reproduces the upsample's edge clamping exactly; taps whose
// i falls outside [0, 2n) are the conv's zero padding and are masked.
// ---------------------------------------------------------------------------------------------
MIVP_DEV constexpr int up_lo(int a, int k) { return (a + k) <= 0 ? 0 : 1; }                     // index of the first neighbour (0: N[-1], 1: N[0])
MIVP_DEV constexpr float up_w0(int a, int k) { return ((a + k) == -1 || (a + k) == 1) ? 0.75f : 0.25f; }

template <int COUT>
__global__ __launch_bounds__(256) void k_uphead_gather(const _Float16* __restrict__ Y, const float* __restrict__ bias, int B,
                                                       int h, int w, int d, float* __restrict__ y) {
    const long T = (long)B * h * w * d;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= T) return;
    const int p2 = (int)(p % d);
    long rest = p / d;
    const int p1 = (int)(rest % w);
    rest /= w;
    const int p0 = (int)(rest % h);
    const long b = rest / h;
    int n0[3], n1[3], n2[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        n0[m] = min(max(p0 + m - 1, 0), h - 1);
        n1[m] = min(max(p1 + m - 1, 0), w - 1);
        n2[m] = min(max(p2 + m - 1, 0), d - 1);
    }
    float acc[8][COUT];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[c][co] = bias ? bias[co] : 0.f;

    // the tap loop is fully unrolled: every neighbour index and weight below is a compile-time constant
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int k0 = tap / 9 - 1, k1 = (tap / 3) % 3 - 1, k2 = tap % 3 - 1;
        const _Float16* Yt = Y + (long)tap * T * COUT;
        float N[3][3][3][COUT];
#pragma unroll
        for (int m0 = 0; m0 < 3; ++m0)
#pragma unroll
            for (int m1 = 0; m1 < 3; ++m1)
#pragma unroll
                for (int m2 = 0; m2 < 3; ++m2) {
                    // slices no parity of this tap reads are skipped (k = -1 never touches N[+1], k = +1 never N[-1])
                    if ((k0 == -1 && m0 == 2) || (k0 == 1 && m0 == 0) || (k1 == -1 && m1 == 2) || (k1 == 1 && m1 == 0) ||
                        (k2 == -1 && m2 == 2) || (k2 == 1 && m2 == 0)) continue;
                    const _Float16* src = Yt + ((((b * h + n0[m0]) * w + n1[m1]) * d) + n2[m2]) * COUT;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) N[m0][m1][m2][co] = (float)src[co];
                }
        const bool in0[2] = {(unsigned)(2 * p0 + k0) < (unsigned)(2 * h), (unsigned)(2 * p0 + 1 + k0) < (unsigned)(2 * h)};
        const bool in1[2] = {(unsigned)(2 * p1 + k1) < (unsigned)(2 * w), (unsigned)(2 * p1 + 1 + k1) < (unsigned)(2 * w)};
        const bool in2[2] = {(unsigned)(2 * p2 + k2) < (unsigned)(2 * d), (unsigned)(2 * p2 + 1 + k2) < (unsigned)(2 * d)};
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            // separable: interpolate along d, then w, then h
            float R[3][3][2];
#pragma unroll
            for (int m0 = 0; m0 < 3; ++m0)
#pragma unroll
                for (int m1 = 0; m1 < 3; ++m1)
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        if ((k0 == -1 && m0 == 2) || (k0 == 1 && m0 == 0) || (k1 == -1 && m1 == 2) || (k1 == 1 && m1 == 0)) continue;
                        const int lo = up_lo(a2, k2);
                        const float wa = up_w0(a2, k2);
                        R[m0][m1][a2] = wa * N[m0][m1][lo][co] + (1.f - wa) * N[m0][m1][lo + 1][co];
                    }
            float Q[3][2][2];
#pragma unroll
            for (int m0 = 0; m0 < 3; ++m0)
#pragma unroll
                for (int a1 = 0; a1 < 2; ++a1)
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        if ((k0 == -1 && m0 == 2) || (k0 == 1 && m0 == 0)) continue;
                        const int lo = up_lo(a1, k1);
                        const float wa = up_w0(a1, k1);
                        Q[m0][a1][a2] = wa * R[m0][lo][a2] + (1.f - wa) * R[m0][lo + 1][a2];
                    }
#pragma unroll
            for (int a0 = 0; a0 < 2; ++a0)
#pragma unroll
                for (int a1 = 0; a1 < 2; ++a1)
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        const int lo = up_lo(a0, k0);
                        const float wa = up_w0(a0, k0);
                        const float v = wa * Q[lo][a1][a2] + (1.f - wa) * Q[lo + 1][a1][a2];
                        acc[(a0 * 2 + a1) * 2 + a2][co] += (in0[a0] && in1[a1] && in2[a2]) ? v : 0.f;
                    }
        }
    }
    const int OH = 2 * h, OW = 2 * w, OD = 2 * d;
#pragma unroll
    for (int a0 = 0; a0 < 2; ++a0)
#pragma unroll
        for (int a1 = 0; a1 < 2; ++a1)
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
                float* dst = y + ((((b * OH + 2 * p0 + a0) * OW + 2 * p1 + a1) * OD) + 2 * p2 + a2) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) dst[co] = acc[(a0 * 2 + a1) * 2 + a2][co];
            }
}


// ---------------------------------------------------------------------------------------------
// adjoint of the gather: D[q][tap*COUT + co] = sum_u [u + tap inside] * U3[u + tap, q] * dy[u][co]   (bf16, row stride ldD)
// Per axis the hr positions whose interpolation touches low-res q are i_j = 2q - 1 + j, j = 0..3, with weight c[j]
// (lerp2 handles the clamped edges; i outside [0, 2n) is the conv's zero padding: weight 0), and u = i - k.
// One thread per low-res cell, separable accumulation: d first (E), then w (F), then h (D).
// ---------------------------------------------------------------------------------------------
MIVP_DEV void axis_coef(int q, int n, float (&c)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 2 * q - 1 + j;
        float v = 0.f;
        if (i >= 0 && i < 2 * n) {
            const Lerp2 l = lerp2(i, n);
            v = (l.i0 == q ? l.w0 : 0.f) + (l.i1 == q ? l.w1 : 0.f);
        }
        c[j] = v;
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void k_uphead_adjoint(const float* __restrict__ dy, int dy_stride, int B, int h, int w, int d,
                                                        bf16_t* __restrict__ D, int ldD) {
    const long T = (long)B * h * w * d;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= T) return;
    const int q2 = (int)(p % d);
    long rest = p / d;
    const int q1 = (int)(rest % w);
    rest /= w;
    const int q0 = (int)(rest % h);
    const long b = rest / h;
    float c0[4], c1[4], c2[4];
    axis_coef(q0, h, c0);
    axis_coef(q1, w, c1);
    axis_coef(q2, d, c2);
    const int OH = 2 * h, OW = 2 * w, OD = 2 * d;
    float acc[3][3][3][COUT];
#pragma unroll
    for (int k0 = 0; k0 < 3; ++k0)
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[k0][k1][k2][co] = 0.f;
    // the h loop stays a real loop (its three tap weights are recomputed per slice) so that only one 6x6 slice of dy
    // is in flight: fully unrolled, the 108 loads of a cell were hoisted together and spilled
#pragma unroll 1
    for (int v0 = 0; v0 < 6; ++v0) {
        const int u0 = 2 * q0 - 2 + v0;
        float F[3][3][COUT];
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                for (int co = 0; co < COUT; ++co) F[k1][k2][co] = 0.f;
        const bool ok0 = (unsigned)u0 < (unsigned)OH;
#pragma unroll
        for (int v1 = 0; v1 < 6; ++v1) {
            const int u1 = 2 * q1 - 2 + v1;
            const bool ok1 = ok0 && (unsigned)u1 < (unsigned)OW;
            const float* row = dy + (((b * OH + (ok0 ? u0 : 0)) * OW + (ok1 ? u1 : 0)) * (long)OD) * dy_stride;
            float rv[6][COUT];
            if (COUT == 2 && dy_stride == 2) {
                // u2 = 2q2-2 .. 2q2+3 is three aligned voxel pairs, each entirely inside or outside: three 16-byte loads
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
                    const int u2 = 2 * q2 - 2 + 2 * pr;
                    const bool ok = ok1 && (unsigned)u2 < (unsigned)OD;
                    const float4 v = ok ? *reinterpret_cast<const float4*>(row + (long)u2 * 2) : make_float4(0.f, 0.f, 0.f, 0.f);
                    rv[2 * pr][0] = v.x; rv[2 * pr][COUT - 1] = v.y; rv[2 * pr + 1][0] = v.z; rv[2 * pr + 1][COUT - 1] = v.w;
                }
            } else {
#pragma unroll
                for (int v2 = 0; v2 < 6; ++v2) {
                    const int u2 = 2 * q2 - 2 + v2;
                    const bool ok = ok1 && (unsigned)u2 < (unsigned)OD;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) rv[v2][co] = ok ? row[(long)u2 * dy_stride + co] : 0.f;
                }
            }
            // along d: tap index k2 (offset k = k2 - 1) reads u2 = i_j - k, i.e. local v2 = j + 1 - k = j + 2 - k2
            float E[3][COUT];
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    float e = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) e += c2[j] * rv[j + 2 - k2][co];
                    E[k2][co] = e;
                }
            // along w: this v1 serves tap k1 through j1 = v1 - 2 + k1
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1) {
                const int j1 = v1 - 2 + k1;
                if (j1 < 0 || j1 > 3) continue;
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) F[k1][k2][co] += c1[j1] * E[k2][co];
            }
        }
#pragma unroll
        for (int k0 = 0; k0 < 3; ++k0) {
            const int j0 = v0 - 2 + k0;                            // c0[j0], 0 outside 0..3 (selects: v0 is a run-time value)
            const float cw = j0 == 0 ? c0[0] : (j0 == 1 ? c0[1] : (j0 == 2 ? c0[2] : (j0 == 3 ? c0[3] : 0.f)));
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[k0][k1][k2][co] += cw * F[k1][k2][co];
        }
    }
    bf16_t* dst = D + p * ldD;
#pragma unroll
    for (int k0 = 0; k0 < 3; ++k0)
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                for (int co = 0; co < COUT; ++co) dst[((k0 * 3 + k1) * 3 + k2) * COUT + co] = (bf16_t)acc[k0][k1][k2][co];
    for (int c = 27 * COUT; c < ldD; ++c) dst[c] = (bf16_t)0.0f;
}

// Cout == 2, contiguous dy: the same adjoint with the dy neighbourhoods served from LDS.  A workgroup owns a brick of
// 4 x 4 x 16 low-res cells (one thread each) and stages the 12 x 12 x 36 high-res voxels they touch ONCE (10 16-byte
// loads per thread instead of 108 overlapping ones per cell: the plain kernel is bound by its load instruction count).
constexpr int AB_H = 4, AB_W = 4, AB_D = 16;
constexpr int AV_H = 2 * AB_H + 4, AV_W = 2 * AB_W + 4, AV_D = 2 * AB_D + 4;     // staged high-res voxels per axis
__global__ __launch_bounds__(256) void k_uphead_adjoint_brick(const float* __restrict__ dy, int B, int h, int w, int d,
                                                              bf16_t* __restrict__ D, int ldD) {
    constexpr int COUT = 2;
    __shared__ __attribute__((aligned(16))) float sdy[AV_H * AV_W * AV_D * 2];
    const int nbd = (d + AB_D - 1) / AB_D, nbw = (w + AB_W - 1) / AB_W, nbh = (h + AB_H - 1) / AB_H;
    int brick = blockIdx.x;
    const int bd = brick % nbd; brick /= nbd;
    const int bw = brick % nbw; brick /= nbw;
    const int bh = brick % nbh;
    const long b = brick / nbh;
    const int OH = 2 * h, OW = 2 * w, OD = 2 * d;
    const int o0 = 2 * bh * AB_H - 2, o1 = 2 * bw * AB_W - 2, o2 = 2 * bd * AB_D - 2;     // brick origin in high-res voxels (even)
    for (int e = threadIdx.x; e < AV_H * AV_W * (AV_D / 2); e += 256) {                   // voxel pairs along d
        const int pr = e % (AV_D / 2), v1 = (e / (AV_D / 2)) % AV_W, v0 = e / ((AV_D / 2) * AV_W);
        const int u0 = o0 + v0, u1 = o1 + v1, u2 = o2 + 2 * pr;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)u0 < (unsigned)OH && (unsigned)u1 < (unsigned)OW && (unsigned)u2 < (unsigned)OD)
            val = *reinterpret_cast<const float4*>(dy + ((((b * OH + u0) * OW + u1) * (long)OD) + u2) * 2);
        *reinterpret_cast<float4*>(sdy + ((v0 * AV_W + v1) * AV_D + 2 * pr) * 2) = val;
    }
    __syncthreads();
    const int t = threadIdx.x;
    const int l2 = t % AB_D, l1 = (t / AB_D) % AB_W, l0 = t / (AB_D * AB_W);
    const int q0 = bh * AB_H + l0, q1 = bw * AB_W + l1, q2 = bd * AB_D + l2;
    const bool inside = q0 < h && q1 < w && q2 < d;
    float c0[4], c1[4], c2[4];
    axis_coef(inside ? q0 : 0, h, c0);
    axis_coef(inside ? q1 : 0, w, c1);
    axis_coef(inside ? q2 : 0, d, c2);
    float acc[3][3][3][COUT];
#pragma unroll
    for (int k0 = 0; k0 < 3; ++k0)
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[k0][k1][k2][co] = 0.f;
    if (inside) {
#pragma unroll 1
    for (int v0 = 0; v0 < 6; ++v0) {
        float F[3][3][COUT];
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                for (int co = 0; co < COUT; ++co) F[k1][k2][co] = 0.f;
#pragma unroll
        for (int v1 = 0; v1 < 6; ++v1) {
            // cell-local voxel (v0, v1, .) = staged voxel (2 l0 + v0, 2 l1 + v1, 2 l2 + .); outside the volume: staged zeros
            const float* row = sdy + (((2 * l0 + v0) * AV_W + (2 * l1 + v1)) * AV_D + 2 * l2) * 2;
            float rv[6][COUT];
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {
                const float4 v = *reinterpret_cast<const float4*>(row + 4 * pr);
                rv[2 * pr][0] = v.x; rv[2 * pr][1] = v.y; rv[2 * pr + 1][0] = v.z; rv[2 * pr + 1][1] = v.w;
            }
            float E[3][COUT];
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    float e = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) e += c2[j] * rv[j + 2 - k2][co];
                    E[k2][co] = e;
                }
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1) {
                const int j1 = v1 - 2 + k1;
                if (j1 < 0 || j1 > 3) continue;
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) F[k1][k2][co] += c1[j1] * E[k2][co];
            }
        }
#pragma unroll
        for (int k0 = 0; k0 < 3; ++k0) {
            const int j0 = v0 - 2 + k0;
            const float cw = j0 == 0 ? c0[0] : (j0 == 1 ? c0[1] : (j0 == 2 ? c0[2] : (j0 == 3 ? c0[3] : 0.f)));
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[k0][k1][k2][co] += cw * F[k1][k2][co];
        }
    }
    }
    // rows leave through LDS: a thread's own 128-byte row would be written as eight 16-byte stores at a 128-byte lane
    // stride (64 partial lines per instruction); staged (row stride 144 B: no bank conflicts), a wave writes whole rows
    __syncthreads();                                          // every wave is done reading the dy brick
    constexpr int RSTRIDE = 144;
    char* srow = reinterpret_cast<char*>(sdy);
    if (inside) {
        bf16_t vals[64];
#pragma unroll
        for (int k0 = 0; k0 < 3; ++k0)
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) vals[((k0 * 3 + k1) * 3 + k2) * COUT + co] = (bf16_t)acc[k0][k1][k2][co];
#pragma unroll
        for (int c = 27 * COUT; c < 64; ++c) vals[c] = (bf16_t)0.0f;
#pragma unroll
        for (int sgm = 0; sgm < 8; ++sgm) {
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = vals[8 * sgm + i];
            *reinterpret_cast<bf16x8*>(srow + t * RSTRIDE + 16 * sgm) = v;
        }
    }
    __syncthreads();
    const int segs = ldD / 8;                                 // 16-byte pieces per D row (ldD <= 64)
    for (int e = t; e < 256 * segs; e += 256) {
        const int row = e / segs, sgm = e - row * segs;
        const int r2 = row % AB_D, r1 = (row / AB_D) % AB_W, r0 = row / (AB_D * AB_W);
        const int g0 = bh * AB_H + r0, g1 = bw * AB_W + r1, g2 = bd * AB_D + r2;
        if (g0 < h && g1 < w && g2 < d) {
            const long pcell = ((b * h + g0) * w + g1) * (long)d + g2;
            st8(D + pcell * ldD + 8 * sgm, *reinterpret_cast<const bf16x8*>(srow + row * RSTRIDE + 16 * sgm));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gradient w.r.t. the low-resolution input.  With z = scale*Ux + shift, dz = conv^T dy, the training-mode BatchNorm
// backward is dUx = scale * (dz - S1/N - xhat * S2/N) (S1 = sum dz = dbeta, S2 = sum dz*xhat = dgamma, xhat = (Ux - mu) rstd)
// and dx = U^T dUx.  Every piece lives at low resolution:
//     U^T dz      = D Wc^T          (D = the adjoint tensor of k_uphead_adjoint, Wc[c][tap*Cout+co] = conv weight)
//     U^T 1       = 8
//     U^T (U x)   = the 27-point Gram stencil of k_uphead_stats applied to x
//   dx[p][c] = scale_c * ( (D Wc^T)[p][c] - 8*k1_c - k2_c * (gram(x)[p][c] - 8*mu_c) ),  k1 = S1/N, k2 = rstd*S2/N
// (eval-mode BatchNorm: k1 = k2 = 0).  One wave = 16 cells; MFMA for the D Wc^T part, lane = (cell r, 4 channels).
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256) void k_uphead_dx(const bf16_t* __restrict__ D, const bf16_t* __restrict__ wc,
                                                   const bf16_t* __restrict__ x, const float* __restrict__ coef,
                                                   const float* __restrict__ gx, int B, int h, int w, int d, int C,
                                                   bf16_t* __restrict__ dx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const long T = (long)B * h * w * d;
    const long p = ((long)blockIdx.x * 4 + wave) * 16 + r;
    const bool live = p < T;
    const long pp = live ? p : 0;
    bf16x8 db[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) db[s] = live ? ld8(D + pp * 64 + 32 * s + 8 * g) : zero8();
    const int p2 = (int)(pp % d);
    long rest = pp / d;
    const int p1 = (int)(rest % w);
    rest /= w;
    const int p0 = (int)(rest % h);
    const float* sc = coef, *k1 = coef + C, *k2 = coef + 2 * C, *mu = coef + 3 * C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc = fzero4();
#pragma unroll
        for (int s = 0; s < 2; ++s) acc = mfma16(ld8(wc + (long)(16 * ct + r) * 64 + 32 * s + 8 * g), db[s], acc);
        const int c0 = 16 * ct + 4 * g;
        if (!live || c0 >= C) continue;
        f32x4 gram = fzero4();
        const bf16_t* xp = x + pp * C + c0;
        if (gx) {                                                 // the forward's statistics pass kept U^T U x
            const float4 q = *reinterpret_cast<const float4*>(gx + pp * C + c0);
            gram[0] = q.x; gram[1] = q.y; gram[2] = q.z; gram[3] = q.w;
        } else
#pragma unroll
        for (int m0 = -1; m0 <= 1; ++m0) {
            const float g0 = gram_w(p0, m0, h);
#pragma unroll
            for (int m1 = -1; m1 <= 1; ++m1) {
                const float g01 = g0 * gram_w(p1, m1, w);
#pragma unroll
                for (int m2 = -1; m2 <= 1; ++m2) {
                    const float gg = g01 * gram_w(p2, m2, d);
                    if (gg != 0.f) {
                        const bf16x4 nv = ld4(xp + (((long)m0 * w + m1) * d + m2) * C);
#pragma unroll
                        for (int j = 0; j < 4; ++j) gram[j] += gg * (float)nv[j];
                    }
                }
            }
        }
        f32x4 out;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            out[j] = sc[c0 + j] * (acc[j] - 8.f * k1[c0 + j] - k2[c0 + j] * (gram[j] - 8.f * mu[c0 + j]));
        st4(dx + pp * C + c0, pack4(out));
    }
}

}  // namespace

static int uphead_checks(int B, int h, int w, int d, int C, int Cout) {
    MIVP_REQUIRE(B > 0 && h > 0 && w > 0 && d > 0);
    MIVP_REQUIRE(C % 8 == 0 && C + 1 <= 64);
    MIVP_REQUIRE(Cout >= 1 && Cout <= 4);
    return MIVP_OK;
}

// ---------------------------------------------------------------------------------------------
// The head's small tensor algebra as two single-workgroup kernels (it used to be ~20 tiny torch launches per step,
// ~5 us each: 3 % of the downstream training step).
// ---------------------------------------------------------------------------------------------
namespace {
// wf bf16 [2][mp][64]: row tap*Cout + co = ( w[co][c][tap] * scale[c] | sum_c w[co][c][tap] * shift[c] | 0 ... ) as a
// hi + lo pair (hi = bf16(v), lo = bf16(v - hi)): plane 0 hi, plane 1 lo
__global__ __launch_bounds__(256) void k_uphead_fold(const float* __restrict__ w, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int cout, int cin, int mp,
                                                     bf16_t* __restrict__ wf) {
    // one wave per row, lane = column: every load of a row is independent, the shift column is a wave reduction
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < mp; row += 4 * gridDim.x) {
        float v = 0.f, part = 0.f;
        if (row < 27 * cout && lane < cin) {
            const int tap = row / cout, co = row - tap * cout;
            const float wv = w[((long)co * cin + lane) * 27 + tap];
            v = wv * scale[lane];
            part = wv * shift[lane];
        }
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == cin) v = part;
        const bf16_t hi = (bf16_t)v;
        wf[row * 64 + lane] = hi;
        wf[(mp + row) * 64 + lane] = (bf16_t)(v - (float)hi);
    }
}

// (G, S) -> conv dW [Cout][Cin][27], conv db [Cout], BatchNorm dgamma / dbeta [Cin]   (fixed summation order)
__global__ __launch_bounds__(256) void k_head_grads(const float* __restrict__ G, long gs_co, long gs_tap,
                                                    const float* __restrict__ S, long ss_co, long ss_tap,
                                                    const float* __restrict__ w, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float* __restrict__ mean_rstd,
                                                    int cout, int cin, float* __restrict__ dW, float* __restrict__ db,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    for (int e = threadIdx.x; e < cout * cin * 27; e += 256) {
        const int co = e / (cin * 27), ci = (e / 27) % cin, tap = e % 27;
        dW[e] = G[co * gs_co + tap * gs_tap + ci] * scale[ci] + S[co * ss_co + tap * ss_tap] * shift[ci];
    }
    for (int co = threadIdx.x; co < cout; co += 256) db[co] = S[co * ss_co + 13 * ss_tap];
    for (int ci = threadIdx.x; ci < cin; ci += 256) {
        const float mean = mean_rstd[ci], rstd = mean_rstd[cin + ci];
        float sb = 0.f, sg = 0.f;
        for (int co = 0; co < cout; ++co)
            for (int tap = 0; tap < 27; ++tap) {
                const float wv = w[((long)co * cin + ci) * 27 + tap];
                const float sv = S[co * ss_co + tap * ss_tap];
                sb += wv * sv;
                sg += wv * (G[co * gs_co + tap * gs_tap + ci] - sv * mean);
            }
        dbeta[ci] = sb;
        dgamma[ci] = rstd * sg;
    }
}
}  // namespace

// operands of k_uphead_dx from the head's parameters and statistics (one workgroup; was ~10 tiny torch launches):
//   wc bf16 [16*ceil(C/16)][64] = conv weight as [c][tap*Cout + co] (zero padded)
//   coef f32 [4][C] = (BN scale | dbeta / N | rstd * dgamma / N | batch mean); rows 1-2 zero for eval-mode BatchNorm
namespace {
__global__ __launch_bounds__(256) void k_uphead_dx_prep(const float* __restrict__ w, const float* __restrict__ scale,
                                                        const float* __restrict__ mean_rstd, const float* __restrict__ dgamma,
                                                        const float* __restrict__ dbeta, float inv_n, int training, int cout,
                                                        int cin, int cp, bf16_t* __restrict__ wc, float* __restrict__ coef) {
    for (int e = threadIdx.x; e < cp * 64; e += 256) {
        const int c = e >> 6, col = e & 63;
        float v = 0.f;
        if (c < cin && col < 27 * cout) { const int tap = col / cout, co = col - tap * cout; v = w[((long)co * cin + c) * 27 + tap]; }
        wc[e] = (bf16_t)v;
    }
    for (int c = threadIdx.x; c < cin; c += 256) {
        coef[c] = scale[c];
        coef[cin + c] = training ? dbeta[c] * inv_n : 0.f;
        coef[2 * cin + c] = training ? mean_rstd[cin + c] * dgamma[c] * inv_n : 0.f;
        coef[3 * cin + c] = mean_rstd[c];
    }
}
}  // namespace

extern "C" int mivp_uphead_dx_prep(const float* conv_w, const float* scale, const float* mean_rstd, const float* dgamma,
                                   const float* dbeta, double n_hr, int32_t training, int32_t Cout, int32_t Cin, void* wc,
                                   float* coef, mivp_stream_t stream) {
    MIVP_REQUIRE(conv_w && scale && mean_rstd && wc && coef && Cout >= 1 && Cin >= 1 && 27 * Cout <= 64 && n_hr > 0);
    MIVP_REQUIRE(!training || (dgamma && dbeta));
    const int cp = (Cin + 15) / 16 * 16;
    hipLaunchKernelGGL(k_uphead_dx_prep, dim3(1), dim3(256), 0, (hipStream_t)stream, conv_w, scale, mean_rstd, dgamma, dbeta,
                       (float)(1.0 / n_hr), (int)training, (int)Cout, (int)Cin, cp, (bf16_t*)wc, coef);
    return mivp_check_launch("uphead_dx_prep");
}

extern "C" int mivp_uphead_fold(const float* conv_w, const float* scale, const float* shift, int32_t Cout, int32_t Cin,
                                void* wf, mivp_stream_t stream) {
    MIVP_REQUIRE(conv_w && scale && shift && wf);
    MIVP_REQUIRE(Cout >= 1 && Cin >= 1 && Cin < 64);
    const int mp = (27 * Cout + 15) / 16 * 16;
    hipLaunchKernelGGL(k_uphead_fold, dim3((mp + 3) / 4), dim3(256), 0, (hipStream_t)stream, conv_w, scale, shift, Cout, Cin, mp, (bf16_t*)wf);
    return mivp_check_launch("uphead_fold");
}

extern "C" int mivp_head_grads(const float* G, int64_t gs_co, int64_t gs_tap, const float* S, int64_t ss_co, int64_t ss_tap,
                               const float* conv_w, const float* scale, const float* shift, const float* mean_rstd,
                               int32_t Cout, int32_t Cin, float* dW, float* db, float* dgamma, float* dbeta,
                               mivp_stream_t stream) {
    MIVP_REQUIRE(G && S && conv_w && scale && shift && mean_rstd && dW && db && dgamma && dbeta);
    MIVP_REQUIRE(Cout >= 1 && Cin >= 1);
    hipLaunchKernelGGL(k_head_grads, dim3(1), dim3(256), 0, (hipStream_t)stream, G, (long)gs_co, (long)gs_tap, S, (long)ss_co,
                       (long)ss_tap, conv_w, scale, shift, mean_rstd, Cout, Cin, dW, db, dgamma, dbeta);
    return mivp_check_launch("head_grads");
}

extern "C" int mivp_uphead_nblk(int32_t B, int32_t h, int32_t w, int32_t d, int32_t C) {
    const int nseg = (d + ST_SEG - 1) / ST_SEG;
    return (int)fixed_group_grid((long)B * h * w * nseg * (C / 8), C / 8, 2048);
}

extern "C" int mivp_uphead_stats(const void* x, int32_t B, int32_t h, int32_t w, int32_t d, int32_t C, float* part,
                                 float* gx, mivp_stream_t stream) {
    int rc = uphead_checks(B, h, w, d, C, 1);
    if (rc) return rc;
    MIVP_REQUIRE(x && part);
    MIVP_REQUIRE((long)B * h * w * ((d + ST_SEG - 1) / ST_SEG) * (C / 8) < (1L << 31));     // 32-bit decode in the kernel
    hipLaunchKernelGGL(k_uphead_stats, dim3(mivp_uphead_nblk(B, h, w, d, C)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (int)B, (int)h, (int)w, (int)d, (int)C, part, gx);
    return mivp_check_launch("uphead_stats");
}

extern "C" size_t mivp_uphead_fwd_ws(int32_t B, int32_t h, int32_t w, int32_t d, int32_t Cout) {
    return (size_t)27 * B * h * w * d * Cout * sizeof(_Float16);
}

extern "C" int mivp_uphead_fwd(const void* x, const void* wf, const float* bias, int32_t B, int32_t h, int32_t w, int32_t d,
                               int32_t C, int32_t Cout, void* workspace, float* y, mivp_stream_t stream) {
    int rc = uphead_checks(B, h, w, d, C, Cout);
    if (rc) return rc;
    MIVP_REQUIRE(x && wf && workspace && y);
    const long T = (long)B * h * w * d;
    hipStream_t st = (hipStream_t)stream;
    _Float16* Y = (_Float16*)workspace;
    const unsigned g1 = (unsigned)((T + 64 * TAPS_TILES - 1) / (64 * TAPS_TILES)), g2 = (unsigned)((T + 255) / 256);
#define UP_FWD(MT, CO)                                                                                             \
    do {                                                                                                           \
        hipLaunchKernelGGL((k_uphead_taps<MT, CO>), dim3(g1), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wf, T, (int)C, Y); \
        hipLaunchKernelGGL((k_uphead_gather<CO>), dim3(g2), dim3(256), 0, st, (const _Float16*)Y, bias, (int)B, (int)h, (int)w,     \
                           (int)d, y);                                                                             \
    } while (0)
    switch (Cout) {
        case 1: UP_FWD(2, 1); break;
        case 2: UP_FWD(4, 2); break;
        case 3: UP_FWD(6, 3); break;
        default: UP_FWD(7, 4); break;
    }
#undef UP_FWD
    return mivp_check_launch("uphead_fwd");
}

extern "C" int mivp_uphead_adjoint(const float* dy, int32_t dy_stride, int32_t B, int32_t h, int32_t w, int32_t d, int32_t Cout,
                                   void* D, int32_t ldD, mivp_stream_t stream) {
    int rc = uphead_checks(B, h, w, d, 8, Cout);
    if (rc) return rc;
    MIVP_REQUIRE(dy && D && dy_stride >= Cout && ldD >= 27 * Cout && ldD % 8 == 0);
    const long T = (long)B * h * w * d;
    const unsigned grid = (unsigned)((T + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (Cout == 2 && dy_stride == 2 && ldD <= 64) {          // LDS-staged brick form (the downstream head's case)
        const unsigned bricks = (unsigned)((long)B * ((h + AB_H - 1) / AB_H) * ((w + AB_W - 1) / AB_W) * ((d + AB_D - 1) / AB_D));
        hipLaunchKernelGGL(k_uphead_adjoint_brick, dim3(bricks), dim3(256), 0, st, dy, (int)B, (int)h, (int)w, (int)d, (bf16_t*)D, (int)ldD);
        return mivp_check_launch("uphead_adjoint");
    }
#define UP_ADJ(CO) hipLaunchKernelGGL((k_uphead_adjoint<CO>), dim3(grid), dim3(256), 0, st, dy, (int)dy_stride, (int)B, (int)h, (int)w, \
                                      (int)d, (bf16_t*)D, (int)ldD)
    switch (Cout) {
        case 1: UP_ADJ(1); break;
        case 2: UP_ADJ(2); break;
        case 3: UP_ADJ(3); break;
        default: UP_ADJ(4); break;
    }
#undef UP_ADJ
    return mivp_check_launch("uphead_adjoint");
}

/* dx [B,h,w,d,C] bf16 from D (row stride 64, columns >= 27*Cout zero), wc bf16 [16*ceil(C/16)][64] = conv weight as
 * [c][tap*Cout + co], coef f32 [4][C] = (scale | S1/N | rstd*S2/N | mean); see k_uphead_dx. */
extern "C" int mivp_uphead_dx(const void* D, const void* wc, const void* x, const float* coef, const float* gx, int32_t B,
                              int32_t h, int32_t w, int32_t d, int32_t C, void* dx, mivp_stream_t stream) {
    int rc = uphead_checks(B, h, w, d, C, 1);
    if (rc) return rc;
    MIVP_REQUIRE(D && wc && x && coef && dx);
    const long T = (long)B * h * w * d;
    const unsigned grid = (unsigned)((T + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
#define UP_DX(K) hipLaunchKernelGGL((k_uphead_dx<K>), dim3(grid), dim3(256), 0, st, (const bf16_t*)D, (const bf16_t*)wc,      \
                                    (const bf16_t*)x, coef, gx, (int)B, (int)h, (int)w, (int)d, (int)C, (bf16_t*)dx)
    switch ((C + 15) / 16) {
        case 1: UP_DX(1); break;
        case 2: UP_DX(2); break;
        case 3: UP_DX(3); break;
        default: UP_DX(4); break;
    }
#undef UP_DX
    return mivp_check_launch("uphead_dx");
}
