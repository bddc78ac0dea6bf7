// Segmentation head forward: BatchNorm (as a per-channel affine) + Conv3d 3x3x3 to very few channels
// (swin_unetr/swin_unetr.py:229-237: BatchNorm3d(48) -> Conv3d(48 -> out_ch = 2)).
//
// With Cout = 2 an implicit GEMM wastes 14/16 of every MFMA tile and re-reads each voxel 27 times.  This
// kernel turns the problem around:
//     Y[u][(tap, co)] = sum_c x'[u][c] * w[co][c][tap]        one small GEMM per voxel u:  K = Cin, N = 27*Cout
//     out[v][co]      = bias[co] + sum_tap Y[v + tap][(tap, co)]     27-point gather of Y
// A workgroup owns a 4x4x16 brick of outputs: it computes Y for the brick plus a one-voxel halo on MFMA
// (BatchNorm folded into the weights: x' = scale*x + shift  =>  W' = scale*w and a constant column fed by a
// "ones" input channel that is 1 only for in-bounds voxels, which reproduces the conv's zero padding of the
// NORMALISED tensor), parks Y as fp16 in LDS, then every thread sums the 27 taps of one output voxel.
#include "common.hpp"

namespace {
constexpr int HB_H = 4, HB_W = 4, HB_D = 16;                       // output brick
constexpr int HH = HB_H + 2, HW = HB_W + 2, HD = HB_D + 2;         // with halo
constexpr int HALO = HH * HW * HD;                                 // 648 voxels
constexpr int HTILES = (HALO + 15) / 16;                           // 41 MFMA voxel tiles
constexpr int YROW = 120;                                          // bytes per voxel row of Y in LDS (>= 2*54, multiple of 8)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
}

// A operand, built once per call: row (tap, co) = tap*Cout + co ; column k < Cin: scale[k] * w[co][k][tap] ;
// column k == Cin: sum_c shift[c] * w[co][c][tap] (multiplies the "ones" channel) ; bf16 [2][64][64] (hi | lo parts)
__global__ __launch_bounds__(64) void k_head_pack(MivpConvDesc d, const float* __restrict__ w, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, bf16_t* __restrict__ apack) {
    // one 64-lane block per row (tap, co); lane = column k
    const int Cin = d.Cin, Cout = d.Cout, rows = 27 * Cout;
    const int row = blockIdx.x, k = threadIdx.x;
    float val = 0.f;
    if (row < rows) {
        const int tap = row / Cout, co = row - tap * Cout;
        float part = 0.f;
        for (int c = k; c < Cin; c += 64) part += shift[c] * w[((long)co * Cin + c) * 27 + tap];
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (k < Cin) val = scale[k] * w[((long)co * Cin + k) * 27 + tap];
        else if (k == Cin) val = part;
    }
    // hi + lo pair: a single bf16 rounding of scale * w is a systematic 2^-9 error per weight (see csrc/uphead.hip)
    const bf16_t hi = (bf16_t)val;
    apack[row * 64 + k] = hi;
    apack[(64 + row) * 64 + k] = (bf16_t)(val - (float)hi);
}

template <int KS>     // K steps of 32 covering Cin + 1 (the ones channel)
__global__ __launch_bounds__(256) void k_head_conv_fwd(MivpConvDesc d, const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ apack, const float* __restrict__ bias,
                                                       float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TPW = (HTILES + 3) / 4;                          // voxel tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int H = d.dims[0], W = d.dims[1], D = d.dims[2], Cin = d.Cin, Cout = d.Cout;
    const int rows = 27 * Cout;                                    // <= 64
    const int nbd = (D + HB_D - 1) / HB_D, nbw = (W + HB_W - 1) / HB_W, nbh = (H + HB_H - 1) / HB_H;
    // XCD-aware brick order (blocks b, b+8, ... share an XCD): neighbouring bricks re-read each other's halo,
    // so each XCD walks one contiguous range of bricks and serves those re-reads from its own L2
    const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned qb = nblk >> 3, rb = nblk & 7;
    long bid = (xcd < rb ? xcd * (qb + 1) : rb * (qb + 1) + (xcd - rb) * qb) + idx;
    const int bd = (int)(bid % nbd); bid /= nbd;
    const int bw = (int)(bid % nbw); bid /= nbw;
    const int bh = (int)(bid % nbh); bid /= nbh;
    const long b = bid;
    const int h0 = bh * HB_H, w0 = bw * HB_W, d0 = bd * HB_D;

    // phase 1: Y for the halo'd brick.  All of this wave's voxel fragments are requested first (one batch of
    // independent 16-byte loads), then the MFMAs run: with only ~8 waves per CU the load latency must overlap.
    bf16x8 xb[TPW][KS];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int i = 16 * (wave + 4 * tt) + r;                    // halo-linear voxel of this lane
        const int hh = i / (HW * HD), hw = (i / HD) % HW, hdd = i % HD;
        const int gh = h0 - 1 + hh, gw = w0 - 1 + hw, gd = d0 - 1 + hdd;
        const bool inb = i < HALO && gh >= 0 && gh < H && gw >= 0 && gw < W && gd >= 0 && gd < D;
        const bf16_t* xv = x + (((b * H + (inb ? gh : 0)) * (long)W + (inb ? gw : 0)) * D + (inb ? gd : 0)) * Cin;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 32 * s + 8 * g;
            // Cin % 8 == 0: a 16-byte chunk is either all data, or starts exactly at the "ones" channel, or is padding
            bf16x8 v = zero8();
            if (inb && k0 < Cin) v = ld8(xv + k0);
            if (inb && k0 == Cin) v[0] = (bf16_t)1.0f;            // the "ones" channel: only in-bounds voxels
            xb[tt][s] = v;
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        bf16x8 af[KS], al[KS];                                     // weight rows of this tile: hi and lo parts
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            af[s] = ld8(apack + (16 * mt + r) * 64 + 32 * s + 8 * g);
            al[s] = ld8(apack + (64 + 16 * mt + r) * 64 + 32 * s + 8 * g);
        }
        const int row0 = 16 * mt + 4 * g;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int i = 16 * (wave + 4 * tt) + r;
            f32x4 acc = fzero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) { acc = mfma16(af[s], xb[tt][s], acc); acc = mfma16(al[s], xb[tt][s], acc); }
            if (i < HALO && row0 < rows) {
                f16x4 hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = (_Float16)acc[j];
                *reinterpret_cast<f16x4*>(smem + (size_t)i * YROW + 2 * row0) = hv;
            }
        }
    }
    __syncthreads();

    // phase 2: one output voxel per thread: 27 taps x Cout
    const int oh = tid / (HB_W * HB_D), ow = (tid / HB_D) % HB_W, od = tid % HB_D;
    const int gh = h0 + oh, gw = w0 + ow, gd = d0 + od;
    if (gh < H && gw < W && gd < D) {
        float out[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) out[c] = c < Cout ? bias[c] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dh = tap / 9, dw = (tap / 3) % 3, dd = tap % 3;          // halo index = brick index + tap offset
            const int i = ((oh + dh) * HW + (ow + dw)) * HD + (od + dd);
            const _Float16* yr = reinterpret_cast<const _Float16*>(smem + (size_t)i * YROW) + tap * Cout;
#pragma unroll
            for (int c = 0; c < 8; ++c) if (c < Cout) out[c] += (float)yr[c];
        }
        float* yo = y + (((b * H + gh) * (long)W + gw) * D + gd) * Cout;
        for (int c = 0; c < Cout; ++c) yo[c] = out[c];
    }
}

extern "C" size_t mivp_head_conv_ws(void) { return 2 * 64 * 64 * sizeof(bf16_t); }

extern "C" int mivp_head_conv_fwd(const MivpConvDesc* d, const void* x, const float* w, const float* bias,
                                  const float* scale, const float* shift, void* workspace, float* y,
                                  mivp_stream_t stream) {
    MIVP_REQUIRE(d && x && w && bias && scale && shift && workspace && y);
    MIVP_REQUIRE(d->Cin % 8 == 0 && d->Cin + 1 <= 64 && d->Cout >= 1 && 27 * d->Cout <= 64 && 2 * 27 * d->Cout <= YROW);
    const int ks = (d->Cin + 1 + 31) / 32;
    const long nb = (long)d->B * ((d->dims[0] + HB_H - 1) / HB_H) * ((d->dims[1] + HB_W - 1) / HB_W) * ((d->dims[2] + HB_D - 1) / HB_D);
    const size_t lds = (size_t)HTILES * 16 * YROW;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_head_pack, dim3(64), dim3(64), 0, st, *d, w, scale, shift, (bf16_t*)workspace);   // [2][64][64]
    int rc = mivp_check_launch("head_pack");
    if (rc) return rc;
    if (ks == 1) {
        auto kern = k_head_conv_fwd<1>;
        MIVP_LDS_OPT_IN(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, *d, (const bf16_t*)x, (const bf16_t*)workspace, bias, y);
    } else {
        auto kern = k_head_conv_fwd<2>;
        MIVP_LDS_OPT_IN(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, *d, (const bf16_t*)x, (const bf16_t*)workspace, bias, y);
    }
    return mivp_check_launch("head_conv_fwd");
}
