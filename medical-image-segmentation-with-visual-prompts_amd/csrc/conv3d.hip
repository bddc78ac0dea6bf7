// 3x3x3 stride-1 pad-1 convolution on channels-last bf16 volumes as an implicit GEMM on MFMA.
// Reference call sites: swin_unetr/swin_unetr.py:87,260-266 (bottleneck), :229-237/:385-394 (heads),
// :292-308 (simple residual convs); swin_unetr/unet_blocks.py:46-56,74 (conv_concat, with the
// BatchNorm + LeakyReLU that precede it fused into the operand load as scale/shift + activation).
//
//   out[v][co] = bias[co] + sum_{tap, ci} act(x[v + tap][ci]) * w[co][tap*Cin + ci]
//
// GEMM view: M = B*H*W*D voxels, N = Cout, K = 27*Cin.  Workgroup tile 128 voxels x (16*NTN) channels,
// K step 32, double-buffered LDS, one barrier per step.  Weight tile on MFMA operand A, voxel tile on
// B (see common.hpp): each lane ends up with one voxel and 4 consecutive output channels.
#include "common.hpp"

namespace {
constexpr int BM = 128;      // voxels per workgroup
constexpr int BK = 32;       // k-step
constexpr int ROWB = (BK + 8) * 2;   // padded LDS row in bytes (80: conflict-free for the 16-row b128 reads)

struct VoxCoord { int b, h, w, d; bool ok; };
}

template <int NTN>
__global__ __launch_bounds__(256) void k_conv3d_fwd(MivpConvDesc d, const bf16_t* __restrict__ x,
                                                    const bf16_t* __restrict__ wgt, const float* __restrict__ bias,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const bf16_t* __restrict__ residual, void* __restrict__ yout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = 16 * NTN;
    constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB;
    auto Xs = [&](int buf) -> char* { return smem + buf * (XBYTES + WBYTES); };
    auto Ws = [&](int buf) -> char* { return smem + buf * (XBYTES + WBYTES) + XBYTES; };
    float* aff = reinterpret_cast<float*>(smem + 2 * (XBYTES + WBYTES));   // [2][Cin] when pro_affine

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int H = d.dims[0], W = d.dims[1], D = d.dims[2], Cin = d.Cin;
    const long vol = (long)H * W * D;
    const long M = (long)d.B * vol;
    const long m0 = (long)blockIdx.x * BM;
    const int n_blk0 = blockIdx.y * BN;
    const int Cout_p = (d.Cout + 15) / 16 * 16;
    const int K = 27 * Cin;
    const int nk = d.Kp / BK;

    if (d.pro_affine) {
        for (int c = tid; c < Cin; c += 256) { aff[c] = scale[c]; aff[Cin + c] = shift[c]; }
    }

    // this thread stages X chunks (row, kc) for e = tid and tid + 256 : row = e >> 2, kc = e & 3
    VoxCoord vc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = (tid + 256 * u) >> 2;
        const long m = m0 + row;
        vc[u].ok = m < M;
        const long mm = vc[u].ok ? m : 0;
        const long b = mm / vol;
        long rem = mm - b * vol;
        vc[u].b = (int)b;
        vc[u].h = (int)(rem / ((long)W * D));
        rem -= (long)vc[u].h * W * D;
        vc[u].w = (int)(rem / D);
        vc[u].d = (int)(rem - (long)vc[u].w * D);
    }
    const int kc = tid & 3;
    // W chunks: e = tid + 256*u < BN*4 : row = e >> 2
    constexpr int WCH = (BN * 4 + 255) / 256;

    bf16x8 xreg[2], wreg[WCH];

    auto load_tiles = [&](int ks) {
        const int k = ks * BK + 8 * kc;
        int tap = k / Cin;
        const int ci = k - tap * Cin;
        const bool kval = k < K;
        if (!kval) tap = 0;
        const int dh = tap / 9 - 1, dw = (tap / 3) % 3 - 1, dd = tap % 3 - 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 val = zero8();
            const int hh = vc[u].h + dh, ww = vc[u].w + dw, zz = vc[u].d + dd;
            const bool inb = kval && vc[u].ok && hh >= 0 && hh < H && ww >= 0 && ww < W && zz >= 0 && zz < D;
            if (inb) {
                val = ld8(x + ((((long)vc[u].b * H + hh) * W + ww) * D + zz) * (long)Cin + ci);
                if (d.pro_affine) {            // zero padding is applied AFTER norm + activation: only in-bounds voxels
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float f = (float)val[i] * aff[ci + i] + aff[Cin + ci + i];
                        if (d.pro_lrelu) f = f > 0.f ? f : 0.01f * f;
                        val[i] = (bf16_t)f;
                    }
                }
            }
            xreg[u] = val;
        }
#pragma unroll
        for (int u = 0; u < WCH; ++u) {
            const int e = tid + 256 * u;
            const int row = e >> 2;
            bf16x8 val = zero8();
            if (e < BN * 4 && n_blk0 + row < Cout_p) val = ld8(wgt + (long)(n_blk0 + row) * d.Kp + ks * BK + 8 * (e & 3));
            wreg[u] = val;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = (tid + 256 * u) >> 2;
            *reinterpret_cast<bf16x8*>(Xs(buf) + row * ROWB + 16 * kc) = xreg[u];
        }
#pragma unroll
        for (int u = 0; u < WCH; ++u) {
            const int e = tid + 256 * u;
            if (e < BN * 4) *reinterpret_cast<bf16x8*>(Ws(buf) + (e >> 2) * ROWB + 16 * (e & 3)) = wreg[u];
        }
    };

    f32x4 acc[NTN][2];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) { acc[nt][0] = fzero4(); acc[nt][1] = fzero4(); }

    __syncthreads();                 // aff[] visible before the first prologue
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) load_tiles(ks + 1);
        const bf16x8 xb0 = *reinterpret_cast<const bf16x8*>(Xs(cur) + (32 * wave + r) * ROWB + 16 * g);
        const bf16x8 xb1 = *reinterpret_cast<const bf16x8*>(Xs(cur) + (32 * wave + 16 + r) * ROWB + 16 * g);
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ws(cur) + (16 * nt + r) * ROWB + 16 * g);
            acc[nt][0] = mfma16(a, xb0, acc[nt][0]);
            acc[nt][1] = mfma16(a, xb1, acc[nt][1]);
        }
        if (ks + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // epilogue: lane (r, g) owns voxel m0 + 32*wave + 16*u + r, channels n_blk0 + 16*nt + 4g .. +3
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long m = m0 + 32 * wave + 16 * u + r;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const int co = n_blk0 + 16 * nt + 4 * g;
            if (co >= d.Cout) continue;
            f32x4 val = acc[nt][u];
            const int nval = d.Cout - co < 4 ? d.Cout - co : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < nval && bias) val[j] += bias[co + j];
            if (d.add_residual) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < nval) val[j] += (float)residual[m * d.Cout + co + j];
            }
            if (d.out_f32) {
                float* yo = reinterpret_cast<float*>(yout) + m * d.Cout + co;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < nval) yo[j] = val[j];
            } else {
                bf16_t* yo = reinterpret_cast<bf16_t*>(yout) + m * d.Cout + co;
                if (nval == 4 && (d.Cout & 3) == 0) st4(yo, pack4(val));
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (j < nval) yo[j] = (bf16_t)val[j];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient for a conv with few output channels (the segmentation heads: Cout = 2..8).
//   work item = (tap, 8-channel group of Cin) [+ one item for the bias]; each workgroup walks a
//   contiguous voxel range and keeps Cout x 8 partial sums per item in registers.
// ---------------------------------------------------------------------------------------------
constexpr int WG_MAXCO = 8;

__global__ __launch_bounds__(256) void k_conv3d_wgrad_small(MivpConvDesc d, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const bf16_t* __restrict__ dy, int dy_stride,
                                                            float* __restrict__ part, long vox_per_blk) {
    const int tid = threadIdx.x;
    const int H = d.dims[0], W = d.dims[1], D = d.dims[2], Cin = d.Cin, Cout = d.Cout;
    const int groups = Cin / 8;
    const int items = 27 * groups;
    const long vol = (long)H * W * D, M = (long)d.B * vol;
    const long v0 = (long)blockIdx.x * vox_per_blk;
    const long v1 = v0 + vox_per_blk < M ? v0 + vox_per_blk : M;
    const long rows = (long)Cout * 27 * Cin + Cout;
    float* my = part + (long)blockIdx.x * rows;

    if (tid < items) {
        const int tap = tid / groups, cg = tid - tap * groups;
        const int dh = tap / 9 - 1, dw = (tap / 3) % 3 - 1, dd = tap % 3 - 1;
        float sc[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = d.pro_affine ? scale[cg * 8 + i] : 1.f; sh[i] = d.pro_affine ? shift[cg * 8 + i] : 0.f; }
        float acc[WG_MAXCO][8];
#pragma unroll
        for (int c = 0; c < WG_MAXCO; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[c][i] = 0.f;
        for (long v = v0; v < v1; ++v) {
            const long b = v / vol;
            long rem = v - b * vol;
            const int h = (int)(rem / ((long)W * D));
            rem -= (long)h * W * D;
            const int w = (int)(rem / D);
            const int z = (int)(rem - (long)w * D);
            const int hh = h + dh, ww = w + dw, zz = z + dd;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W || zz < 0 || zz >= D) continue;
            const bf16x8 raw = ld8(x + (((b * H + hh) * W + ww) * (long)D + zz) * Cin + cg * 8);
            float xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float f = (float)raw[i] * sc[i] + sh[i];
                if (d.pro_lrelu) f = f > 0.f ? f : 0.01f * f;
                xv[i] = (float)(bf16_t)f;              // the forward conv saw the bf16-rounded operand
            }
#pragma unroll
            for (int c = 0; c < WG_MAXCO; ++c) {
                if (c < Cout) {
                    const float g = (float)dy[v * dy_stride + c];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[c][i] += g * xv[i];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < WG_MAXCO; ++c) {
            if (c < Cout) {
#pragma unroll
                for (int i = 0; i < 8; ++i) my[((long)c * 27 + tap) * Cin + cg * 8 + i] = acc[c][i];
            }
        }
    } else if (tid == items) {
        float acc[WG_MAXCO];
#pragma unroll
        for (int c = 0; c < WG_MAXCO; ++c) acc[c] = 0.f;
        for (long v = v0; v < v1; ++v) {
#pragma unroll
            for (int c = 0; c < WG_MAXCO; ++c) if (c < Cout) acc[c] += (float)dy[v * dy_stride + c];
        }
#pragma unroll
        for (int c = 0; c < WG_MAXCO; ++c) if (c < Cout) my[(long)Cout * 27 * Cin + c] = acc[c];
    }
}

static int conv_checks(const MivpConvDesc* d) {
    MIVP_REQUIRE(d != nullptr);
    MIVP_REQUIRE(d->B > 0 && d->dims[0] > 0 && d->dims[1] > 0 && d->dims[2] > 0);
    MIVP_REQUIRE(d->Cin > 0 && d->Cin % 8 == 0 && d->Cout > 0);
    MIVP_REQUIRE(d->Kp % 32 == 0 && d->Kp >= 27 * d->Cin && d->Kp < 27 * d->Cin + 32);
    return MIVP_OK;
}

template <int NTN>
static int launch_conv(const MivpConvDesc* d, const void* x, const void* w, const float* bias, const float* scale,
                       const float* shift, const void* residual, void* y, hipStream_t st) {
    const long M = (long)d->B * d->dims[0] * d->dims[1] * d->dims[2];
    const int cout_p = (d->Cout + 15) / 16 * 16;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((cout_p + 16 * NTN - 1) / (16 * NTN)));
    const size_t lds = 2 * (size_t)(BM + 16 * NTN) * ROWB + (d->pro_affine ? 2 * (size_t)d->Cin * 4 : 0);
    hipLaunchKernelGGL((k_conv3d_fwd<NTN>), grid, dim3(256), lds, st, *d, (const bf16_t*)x, (const bf16_t*)w, bias, scale,
                       shift, (const bf16_t*)residual, y);
    return mivp_check_launch("conv3d_fwd");
}

extern "C" int mivp_conv3d_fwd(const MivpConvDesc* d, const void* x, const void* w, const float* bias,
                               const float* scale, const float* shift, const void* residual, void* y,
                               mivp_stream_t stream) {
    int rc = conv_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(x && w && y);
    MIVP_REQUIRE(!d->pro_affine || (scale && shift));
    MIVP_REQUIRE(!d->add_residual || residual);
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (d->Cout + 15) / 16;
    // widest channel tile that divides the work without a mostly-empty last block
    if (tiles % 6 == 0) return launch_conv<6>(d, x, w, bias, scale, shift, residual, y, st);
    if (tiles % 4 == 0) return launch_conv<4>(d, x, w, bias, scale, shift, residual, y, st);
    if (tiles % 3 == 0) return launch_conv<3>(d, x, w, bias, scale, shift, residual, y, st);
    if (tiles % 2 == 0) return launch_conv<2>(d, x, w, bias, scale, shift, residual, y, st);
    return launch_conv<1>(d, x, w, bias, scale, shift, residual, y, st);
}

static int wgrad_blocks(const MivpConvDesc* d) {
    const long M = (long)d->B * d->dims[0] * d->dims[1] * d->dims[2];
    long nb = (M + 2047) / 2048;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int)nb;
}

extern "C" size_t mivp_conv3d_wgrad_small_ws(const MivpConvDesc* d) {
    if (!d) return 0;
    return (size_t)wgrad_blocks(d) * ((size_t)d->Cout * 27 * d->Cin + d->Cout);
}

extern "C" int mivp_reduce_rows(const float* in, int64_t n, int64_t rows, float* out, mivp_stream_t stream);

extern "C" int mivp_conv3d_wgrad_small(const MivpConvDesc* d, const void* x, const float* scale, const float* shift,
                                       const void* dy, int32_t dy_stride, float* part, float* dwdb,
                                       mivp_stream_t stream) {
    int rc = conv_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(x && dy && part && dwdb);
    MIVP_REQUIRE(d->Cout <= WG_MAXCO && dy_stride >= d->Cout);
    MIVP_REQUIRE(27 * (d->Cin / 8) + 1 <= 256);
    MIVP_REQUIRE(!d->pro_affine || (scale && shift));
    const long M = (long)d->B * d->dims[0] * d->dims[1] * d->dims[2];
    const int nb = wgrad_blocks(d);
    const long per = (M + nb - 1) / nb;
    hipLaunchKernelGGL(k_conv3d_wgrad_small, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d, (const bf16_t*)x, scale,
                       shift, (const bf16_t*)dy, (int)dy_stride, part, per);
    rc = mivp_check_launch("conv3d_wgrad_small");
    if (rc) return rc;
    return mivp_reduce_rows(part, nb, (long)d->Cout * 27 * d->Cin + d->Cout, dwdb, stream);
}
