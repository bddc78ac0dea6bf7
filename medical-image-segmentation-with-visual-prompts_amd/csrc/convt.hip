// Transposed convolution with kernel == stride (2x2x2 or 2x2x1), no bias: the up-sampling step of MONAI's UnetrUpBlock
// (swin_unetr/swin_unetr.py:338-348,372-380 with ``unetr_up_block != 'swin'``).  Non-overlapping taps make it
// a per-token GEMM followed by a depth-to-space shuffle:
//     y[b, 2h+a, 2w+b', s d + c, co] = sum_ci x[b, h, w, d, ci] * W[ci, co, a, b', c]
// forward : Y[token][(tap, co)] = X[token][ci] W1[(tap, co)][ci]   and every 4-channel group is stored at its output voxel
// dgrad   : dX[token][ci] = dYg[token][(tap, co)] W2[ci][(tap, co)] with the B operand GATHERED from the tap's output voxel
// (the weight gradient is the TN GEMM of wgrad.hip on x and the space-to-depth view of dy).
// MFMA convention of common.hpp: weight tile on A, token tile on B -- a lane owns one token and four consecutive columns.
#include "common.hpp"

namespace {

struct ConvtGeom {
    int B, h, w, d, s0, s1, s2, Cin, Cout;
    MIVP_DEV int taps() const { return s0 * s1 * s2; }
    // output voxel (linear in the up-sampled volume of one batch element) of low-res token (hh, ww, dd) and tap
    MIVP_DEV long out_voxel(int hh, int ww, int dd, int tap) const {
        const int c = tap % s2, b = (tap / s2) % s1, a = tap / (s2 * s1);
        return ((long)(hh * s0 + a) * (w * s1) + (ww * s1 + b)) * (d * s2) + (dd * s2 + c);
    }
};

// MODE 0: forward (x low-res [T][Cin] -> y high-res, K = Cin, N = taps * Cout)
// MODE 1: dgrad   (dy high-res -> dx low-res [T][Cin], K = taps * Cout, N = Cin)
template <int MODE>
__global__ __launch_bounds__(256) void k_convt(ConvtGeom gm, const bf16_t* __restrict__ in, const bf16_t* __restrict__ wgt,
                                               bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const long T = (long)gm.B * gm.h * gm.w * gm.d;
    const long vol_lo = (long)gm.h * gm.w * gm.d, vol_hi = vol_lo * gm.taps();
    const int K = MODE == 0 ? gm.Cin : gm.taps() * gm.Cout;
    const int N = MODE == 0 ? gm.taps() * gm.Cout : gm.Cin;
    const long t = ((long)blockIdx.x * 4 + wave) * 16 + r;
    const bool live = t < T;
    const long tt = live ? t : 0;
    const long b = tt / vol_lo;
    const int rem = (int)(tt - b * vol_lo);
    const int dd = rem % gm.d, ww = (rem / gm.d) % gm.w, hh = rem / (gm.d * gm.w);
    const int nks = (K + 31) / 32;
    const int n_tiles = (N + 15) / 16;
    const int per_split = (n_tiles + gridDim.y - 1) / gridDim.y;
    const int nt0 = blockIdx.y * per_split;
    const int nt1 = nt0 + per_split < n_tiles ? nt0 + per_split : n_tiles;
    for (int nt = nt0; nt < nt1; ++nt) {
        f32x4 acc = fzero4();
        const int nrow = 16 * nt + r;
        for (int ks = 0; ks < nks; ++ks) {
            const int k0 = 32 * ks + 8 * g;
            bf16x8 a = zero8(), bv = zero8();
            if (nrow < N && k0 < K) a = ld8(wgt + (long)nrow * K + k0);
            if (live && k0 < K) {
                if (MODE == 0) {
                    bv = ld8(in + tt * gm.Cin + k0);
                } else {                                      // 8 consecutive channels of one tap's output voxel (Cout % 8 == 0)
                    const int tap = k0 / gm.Cout, co = k0 - tap * gm.Cout;
                    bv = ld8(in + (b * vol_hi + gm.out_voxel(hh, ww, dd, tap)) * gm.Cout + co);
                }
            }
            acc = mfma16(a, bv, acc);
        }
        const int n0 = 16 * nt + 4 * g;
        if (live && n0 < N) {
            if (MODE == 0) {
                const int tap = n0 / gm.Cout, co = n0 - tap * gm.Cout;
                st4(out + (b * vol_hi + gm.out_voxel(hh, ww, dd, tap)) * gm.Cout + co, pack4(acc));
            } else {
                st4(out + tt * gm.Cin + n0, pack4(acc));
            }
        }
    }
}

}  // namespace

/* x bf16 [B][h][w][d][Cin] -> y bf16 [B][h*s0][w*s1][d*s2][Cout];  w1 bf16 [taps*Cout][Cin], row (tap, co), tap = (a*s1 + b)*s2 + c */
extern "C" int mivp_convt_fwd(int32_t B, const int32_t* dims, const int32_t* stride, int32_t Cin, int32_t Cout, const void* x,
                              const void* w1, void* y, mivp_stream_t stream) {
    MIVP_REQUIRE(dims && stride && x && w1 && y && B > 0);
    MIVP_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0);
    for (int a = 0; a < 3; ++a) MIVP_REQUIRE(dims[a] > 0 && (stride[a] == 1 || stride[a] == 2));
    ConvtGeom gm{B, dims[0], dims[1], dims[2], stride[0], stride[1], stride[2], Cin, Cout};
    const long T = (long)B * dims[0] * dims[1] * dims[2];
    const unsigned gx = (unsigned)((T + 63) / 64);
    const int n_tiles = (stride[0] * stride[1] * stride[2] * Cout + 15) / 16;
    int split = (int)((2048 + gx - 1) / gx);
    if (split > n_tiles) split = n_tiles;
    if (split < 1) split = 1;
    hipLaunchKernelGGL(k_convt<0>, dim3(gx, (unsigned)split), dim3(256), 0, (hipStream_t)stream, gm, (const bf16_t*)x,
                       (const bf16_t*)w1, (bf16_t*)y);
    return mivp_check_launch("convt_fwd");
}

/* dy bf16 [B][h*s0][w*s1][d*s2][Cout] -> dx bf16 [B][h][w][d][Cin];  w2 bf16 [Cin][taps*Cout] */
extern "C" int mivp_convt_dgrad(int32_t B, const int32_t* dims, const int32_t* stride, int32_t Cin, int32_t Cout, const void* dy,
                                const void* w2, void* dx, mivp_stream_t stream) {
    MIVP_REQUIRE(dims && stride && dy && w2 && dx && B > 0);
    MIVP_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0);
    for (int a = 0; a < 3; ++a) MIVP_REQUIRE(dims[a] > 0 && (stride[a] == 1 || stride[a] == 2));
    ConvtGeom gm{B, dims[0], dims[1], dims[2], stride[0], stride[1], stride[2], Cin, Cout};
    const long T = (long)B * dims[0] * dims[1] * dims[2];
    const unsigned gx = (unsigned)((T + 63) / 64);
    const int n_tiles = (Cin + 15) / 16;
    int split = (int)((2048 + gx - 1) / gx);
    if (split > n_tiles) split = n_tiles;
    if (split < 1) split = 1;
    hipLaunchKernelGGL(k_convt<1>, dim3(gx, (unsigned)split), dim3(256), 0, (hipStream_t)stream, gm, (const bf16_t*)dy,
                       (const bf16_t*)w2, (bf16_t*)dx);
    return mivp_check_launch("convt_dgrad");
}
