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
constexpr int BM = 256;            // voxels per workgroup (NW waves x VT 16-voxel MFMA tiles each: NW * VT == 16)
constexpr int BK = 32;             // k-step
constexpr int ROWB = BK * 2;       // LDS row: 64 bytes = four 16-byte chunks, XOR-swizzled (no padding)

// chunk swizzle that makes the 16-row x 16-byte MFMA operand reads (ds_read_b128, lane = (row r, chunk g))
// conflict-free: ds_read_b128 is serviced in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
// (MI355X_MICROARCH.md, LDS); with f = {0,3,2,1}[(row >> 2) & 3] every group covers all sixteen 16-byte slots
// of the 256-byte bank row exactly once (SQ_LDS_BANK_CONFLICT = 0 measured).
MIVP_DEV int swz(int row, int chunk) { return chunk ^ ((0 - (row >> 2)) & 3); }
}

// Wave tile (16*VT) voxels x (16*NTN) channels: per k-step VT + NTN operand reads feed VT*NTN MFMAs.  Global
// loads run TWO k-steps ahead of their use through two named register sets, LDS is double buffered, one
// barrier per step.  Two shapes are instantiated: 8 waves x 32 voxels (more waves to hide latency) and
// 4 waves x 64 voxels (fewer LDS reads per MFMA; wins when NTN is 1).  blockIdx.z selects a K slice
// (split-K for convolutions with few voxels and very long K); slices write f32 partials.
template <int NTN, int NW, int VT>
__global__ __launch_bounds__(64 * NW) void k_conv3d_fwd(MivpConvDesc d, const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ wgt, const float* __restrict__ bias,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const bf16_t* __restrict__ residual, void* __restrict__ yout,
                                                       float* __restrict__ partial, int ksteps_per_slice) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(NW * VT == 16, "workgroup tile is 256 voxels");
    constexpr int CONV_T = 64 * NW;
    constexpr int XCH = BM * 4 / CONV_T;
    constexpr int BN = 16 * NTN;
    constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB;
    auto Xs = [&](int buf) -> char* { return smem + buf * (XBYTES + WBYTES); };
    auto Ws = [&](int buf) -> char* { return smem + buf * (XBYTES + WBYTES) + XBYTES; };
    int* tapoff = reinterpret_cast<int*>(smem + 2 * (XBYTES + WBYTES));    // [28] element offset of each tap (27 = none)
    float* aff = reinterpret_cast<float*>(tapoff + 32);                   // [2][Cin] when pro_affine

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int H = d.dims[0], W = d.dims[1], D = d.dims[2], Cin = d.Cin;
    const long vol = (long)H * W * D;
    const long M = (long)d.B * vol;
    // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch), so give each XCD one
    // CONTIGUOUS range of voxel tiles: the 27 taps re-read the neighbours of every voxel through that XCD's L2.
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned qb = nb >> 3, rb = nb & 7;
    const unsigned tile = (xcd < rb ? xcd * (qb + 1) : rb * (qb + 1) + (xcd - rb) * qb) + idx;
    const long m0 = (long)tile * BM;
    const int n_blk0 = blockIdx.y * BN;
    const int Cout_p = (d.Cout + 15) / 16 * 16;
    const int nk_total = d.Kp / BK;
    const int ks_begin = blockIdx.z * ksteps_per_slice;
    const int ks_end = ks_begin + ksteps_per_slice < nk_total ? ks_begin + ksteps_per_slice : nk_total;
    const int nk = ks_end - ks_begin;          // k-steps of this slice (>= 1 by construction)

    if (tid < 28) {
        const int dh = tid / 9 - 1, dw = (tid / 3) % 3 - 1, dd = tid % 3 - 1;
        tapoff[tid] = tid < 27 ? ((dh * W + dw) * D + dd) * Cin : 0;
    }
    if (d.pro_affine) {
        for (int c = tid; c < Cin; c += CONV_T) { aff[c] = scale[c]; aff[Cin + c] = shift[c]; }
    }

    // this thread stages X chunks (row, kc) for rows (tid >> 2) + 64*u, u < XCH, kc = tid & 3.
    // Per row: element offset of the voxel and a 27-bit mask of the taps that stay inside the volume.
    const int kc = tid & 3;
    long xoff[XCH];
    unsigned okmask[XCH];
#pragma unroll
    for (int u = 0; u < XCH; ++u) {
        const int row = (tid >> 2) + (CONV_T / 4) * u;
        const long m = m0 + row;
        const bool ok = m < M;
        const long mm = ok ? m : 0;
        const long b = mm / vol;
        long rem = mm - b * vol;
        const int h = (int)(rem / ((long)W * D));
        rem -= (long)h * W * D;
        const int w = (int)(rem / D);
        const int z = (int)(rem - (long)w * D);
        xoff[u] = mm * Cin;
        unsigned mask = 0;
        if (ok) {
            for (int t = 0; t < 27; ++t) {
                const int hh = h + t / 9 - 1, ww = w + (t / 3) % 3 - 1, zz = z + t % 3 - 1;
                if (hh >= 0 && hh < H && ww >= 0 && ww < W && zz >= 0 && zz < D) mask |= 1u << t;
            }
        }
        okmask[u] = mask;
    }
    // running (tap, ci) of this thread's k position  k = ks*32 + 8*kc  (no divisions in the loop)
    int tap_run = (ks_begin * BK + 8 * kc) / Cin, ci_run = (ks_begin * BK + 8 * kc) % Cin;

    constexpr int WCH = (BN * 4 + CONV_T - 1) / CONV_T;
    struct Stage { bf16x8 x[XCH]; bf16x8 w[WCH]; };

    auto load_tiles = [&](int ks, Stage& st) {
        const int tap = tap_run < 27 ? tap_run : 27;
        const int toff = tapoff[tap] + ci_run;
#pragma unroll
        for (int u = 0; u < XCH; ++u) {
            bf16x8 val = zero8();
            if (tap < 27 && ((okmask[u] >> tap) & 1u)) {
                val = ld8(x + (xoff[u] + toff));
                if (d.pro_affine) {            // zero padding is applied AFTER norm + activation: only in-bounds voxels
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float f = (float)val[i] * aff[ci_run + i] + aff[Cin + ci_run + i];
                        if (d.pro_lrelu) f = f > 0.f ? f : 0.01f * f;
                        val[i] = (bf16_t)f;
                    }
                }
            }
            st.x[u] = val;
        }
        ci_run += BK;
        while (ci_run >= Cin) { ci_run -= Cin; ++tap_run; }
#pragma unroll
        for (int u = 0; u < WCH; ++u) {
            const int e = tid + CONV_T * u;
            const int row = e >> 2;
            bf16x8 val = zero8();
            if (e < BN * 4 && n_blk0 + row < Cout_p) val = ld8(wgt + (long)(n_blk0 + row) * d.Kp + (ks_begin + ks) * BK + 8 * (e & 3));
            st.w[u] = val;
        }
    };
    auto store_tiles = [&](int buf, const Stage& st) {
#pragma unroll
        for (int u = 0; u < XCH; ++u) {
            const int row = (tid >> 2) + (CONV_T / 4) * u;
            *reinterpret_cast<bf16x8*>(Xs(buf) + row * ROWB + 16 * swz(row, kc)) = st.x[u];
        }
#pragma unroll
        for (int u = 0; u < WCH; ++u) {
            const int e = tid + CONV_T * u;
            if (e < BN * 4) *reinterpret_cast<bf16x8*>(Ws(buf) + (e >> 2) * ROWB + 16 * swz(e >> 2, e & 3)) = st.w[u];
        }
    };

    f32x4 acc[NTN][VT];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
        for (int u = 0; u < VT; ++u) acc[nt][u] = fzero4();

    const int sw = 16 * swz(r, g);   // same swizzle for every 16-row tile (tile bases are multiples of 16)
    auto compute = [&](int buf) {
        bf16x8 xb[VT];
#pragma unroll
        for (int u = 0; u < VT; ++u) xb[u] = *reinterpret_cast<const bf16x8*>(Xs(buf) + (16 * VT * wave + 16 * u + r) * ROWB + sw);
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ws(buf) + (16 * nt + r) * ROWB + sw);
#pragma unroll
            for (int u = 0; u < VT; ++u) acc[nt][u] = mfma16(a, xb[u], acc[nt][u]);
        }
    };

    Stage sa, sb;
    __syncthreads();                 // tapoff[] / aff[] visible before the first load
    load_tiles(0, sa);
    if (nk > 1) load_tiles(1, sb);
    store_tiles(0, sa);
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
        // even step: buffer 0 holds step ks, `sb` holds step ks+1 (in flight), `sa` is free
        if (ks + 2 < nk) load_tiles(ks + 2, sa);
        compute(0);
        if (ks + 1 < nk) store_tiles(1, sb);
        __syncthreads();
        if (ks + 1 >= nk) break;
        // odd step: buffer 1 holds step ks+1, `sa` holds step ks+2 (in flight), `sb` is free
        if (ks + 3 < nk) load_tiles(ks + 3, sb);
        compute(1);
        if (ks + 2 < nk) store_tiles(0, sa);
        __syncthreads();
    }

    // epilogue: lane (r, g) owns voxel m0 + 16*VT*wave + 16*u + r, channels n_blk0 + 16*nt + 4g .. +3
#pragma unroll
    for (int u = 0; u < VT; ++u) {
        const long m = m0 + 16 * VT * wave + 16 * u + r;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const int co = n_blk0 + 16 * nt + 4 * g;
            if (co >= d.Cout) continue;
            f32x4 val = acc[nt][u];
            const int nval = d.Cout - co < 4 ? d.Cout - co : 4;
            if (partial) {                       // split-K: raw f32 partial sums, epilogue kernel finishes
                float* po = partial + ((long)blockIdx.z * M + m) * d.Cout + co;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < nval) po[j] = val[j];
                continue;
            }
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

// split-K epilogue: y = sum_s partial[s] + bias + residual  (fixed summation order)
__global__ __launch_bounds__(256) void k_conv3d_splitk_epilogue(MivpConvDesc d, const float* __restrict__ partial, int slices,
                                                                const float* __restrict__ bias,
                                                                const bf16_t* __restrict__ residual, void* __restrict__ yout) {
    const long M = (long)d.B * d.dims[0] * d.dims[1] * d.dims[2];
    const long total = M * d.Cout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float acc = 0.f;
        for (int s = 0; s < slices; ++s) acc += partial[(long)s * total + i];
        const int co = (int)(i % d.Cout);
        if (bias) acc += bias[co];
        if (d.add_residual) acc += (float)residual[i];
        if (d.out_f32) reinterpret_cast<float*>(yout)[i] = acc;
        else reinterpret_cast<bf16_t*>(yout)[i] = (bf16_t)acc;
    }
}

static int conv_checks(const MivpConvDesc* d) {
    MIVP_REQUIRE(d != nullptr);
    MIVP_REQUIRE(d->B > 0 && d->dims[0] > 0 && d->dims[1] > 0 && d->dims[2] > 0);
    MIVP_REQUIRE(d->Cin > 0 && d->Cin % 8 == 0 && d->Cout > 0);
    MIVP_REQUIRE(d->Kp % 32 == 0 && d->Kp >= 27 * d->Cin && d->Kp < 27 * d->Cin + 32);
    return MIVP_OK;
}

static int conv_ntn(const MivpConvDesc* d) {
    const int tiles = (d->Cout + 15) / 16;
    // widest channel tile that divides the work without a mostly-empty last block
    if (tiles % 6 == 0) return 6;
    if (tiles % 4 == 0) return 4;
    if (tiles % 3 == 0) return 3;
    if (tiles % 2 == 0) return 2;
    return 1;
}

// number of K slices: fill the 256 CUs about twice when the voxel count alone cannot
static int conv_slices(const MivpConvDesc* d, int* ksteps_per_slice) {
    const long M = (long)d->B * d->dims[0] * d->dims[1] * d->dims[2];
    const int ntn = conv_ntn(d);
    const long wgs = ((M + BM - 1) / BM) * (((d->Cout + 15) / 16 + ntn - 1) / ntn);
    const int nk = d->Kp / BK;
    int slices = 1;
    if (wgs < 256) {
        slices = (int)((512 + wgs - 1) / wgs);
        if (slices > nk / 8) slices = nk / 8;
        if (slices < 1) slices = 1;
    }
    int per = (nk + slices - 1) / slices;
    slices = (nk + per - 1) / per;
    *ksteps_per_slice = per;
    return slices;
}

extern "C" size_t mivp_conv3d_fwd_ws(const MivpConvDesc* d) {
    if (!d) return 0;
    int per;
    const int slices = conv_slices(d, &per);
    if (slices <= 1) return 0;
    return (size_t)slices * d->B * d->dims[0] * d->dims[1] * d->dims[2] * d->Cout * sizeof(float);
}

template <int NTN, int NW, int VT>
static int launch_conv(const MivpConvDesc* d, const void* x, const void* w, const float* bias, const float* scale,
                       const float* shift, const void* residual, void* y, float* partial, int slices, int per,
                       hipStream_t st) {
    const long M = (long)d->B * d->dims[0] * d->dims[1] * d->dims[2];
    const int cout_p = (d->Cout + 15) / 16 * 16;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((cout_p + 16 * NTN - 1) / (16 * NTN)), (unsigned)slices);
    const size_t lds = 2 * (size_t)(BM + 16 * NTN) * ROWB + 32 * 4 + (d->pro_affine ? 2 * (size_t)d->Cin * 4 : 0);
    hipLaunchKernelGGL((k_conv3d_fwd<NTN, NW, VT>), grid, dim3(64 * NW), lds, st, *d, (const bf16_t*)x, (const bf16_t*)w, bias,
                       scale, shift, (const bf16_t*)residual, y, slices > 1 ? partial : nullptr, per);
    int rc = mivp_check_launch("conv3d_fwd");
    if (rc || slices <= 1) return rc;
    const long total = M * d->Cout;
    const unsigned g2 = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(k_conv3d_splitk_epilogue, dim3(g2), dim3(256), 0, st, *d, partial, slices, bias,
                       (const bf16_t*)residual, y);
    return mivp_check_launch("conv3d_splitk_epilogue");
}

extern "C" int mivp_conv3d_fwd(const MivpConvDesc* d, const void* x, const void* w, const float* bias,
                               const float* scale, const float* shift, const void* residual, void* y,
                               void* workspace, size_t ws_bytes, mivp_stream_t stream) {
    int rc = conv_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(x && w && y);
    MIVP_REQUIRE(!d->pro_affine || (scale && shift));
    MIVP_REQUIRE(!d->add_residual || residual);
    int per;
    int slices = conv_slices(d, &per);
    if (slices > 1 && (workspace == nullptr || ws_bytes < mivp_conv3d_fwd_ws(d))) { slices = 1; per = d->Kp / BK; }   // no room: run unsplit
    hipStream_t st = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(workspace);
    switch (conv_ntn(d)) {
        case 6: return launch_conv<6, 8, 2>(d, x, w, bias, scale, shift, residual, y, part, slices, per, st);
        case 4: return launch_conv<4, 8, 2>(d, x, w, bias, scale, shift, residual, y, part, slices, per, st);
        case 3: return launch_conv<3, 8, 2>(d, x, w, bias, scale, shift, residual, y, part, slices, per, st);
        case 2: return launch_conv<2, 8, 2>(d, x, w, bias, scale, shift, residual, y, part, slices, per, st);
        default: return launch_conv<1, 4, 4>(d, x, w, bias, scale, shift, residual, y, part, slices, per, st);
    }
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

// ---------------------------------------------------------------------------------------------
// small-Cout weight gradient on MFMA ("rows" formulation), K = the D voxels of one (b,h,w) row.
//   G[sd][nb*8 + co][c] = sum_d' dy[(h-sh, w-sw, d')][co] * x[(h, w, d'+sd)][c]      nb = (sh+1)*3 + (sw+1)
//   i.e. the tap (sh, sw, sd) entry of  sum_u dy[u - tap][co] * x[u][c];  column c = Cin carries a 1 for
//   in-range voxels and so accumulates S = sum_{u in bounds} dy[u - tap][co].
// A workgroup owns one row at a time.  Both operands are staged ROW-major (d outermost) with 16-byte writes:
//   A image [d][9 neighbour rows x 8 dy channels]   B image [d + halo][x channels | 1 | 0]
// and read back through ds_read_b64_tr_b16 (K innermost); the three d-shifts of a tap triple are just the B
// fragment read one row up / down.  Wave w owns x-channel tile w: 3 shifts x 5 row tiles of accumulators.
// G and S are linear in the raw x: the BatchNorm affine in front of the conv, the conv weight gradient, dgamma
// and dbeta all follow from (G, S) by O(27*Cout*Cin) algebra on the host.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) bf16x4 wr_lds_bf16x4;
MIVP_DEV bf16x4 wr_tr_read(const char* smem, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wr_lds_bf16x4*)(smem + byte_off));
}
// blocks of 4 rows x 16 columns (128 B); S blocks per 4-row group, column block XOR-ed with bit 3 of the row so that
// the two 16-lane groups of a half (8 rows apart) fall into opposite halves of the 256-byte bank row
template <int S>
MIVP_DEV int wr_off(int row, int col) {
    return (((row >> 2) * S + ((col >> 4) ^ ((row >> 3) & 1))) << 7) + ((row & 3) << 5) + ((col & 15) << 1);
}
constexpr int WR_AS = 6, WR_BS = 4, WR_MT = 5;

__global__ __launch_bounds__(256) void k_conv3d_wgrad_rows(MivpConvDesc d, const bf16_t* __restrict__ x,
                                                           const bf16_t* __restrict__ dy, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int H = d.dims[0], W = d.dims[1], D = d.dims[2], Cin = d.Cin;
    const int Dp = (D + 31) / 32 * 32, G8 = Cin / 8;
    char* Aimg = smem;                                            // rows 0 .. Dp-1
    char* Bimg = smem + (size_t)(Dp / 4) * WR_AS * 128;           // rows 0 .. Dp+7  (row = d + 4)
    const long rows_total = (long)d.B * H * W;
    const bool wave_on = 16 * wave < Cin + 1;

    f32x4 acc[3][WR_MT];
#pragma unroll
    for (int sd = 0; sd < 3; ++sd)
#pragma unroll
        for (int mt = 0; mt < WR_MT; ++mt) acc[sd][mt] = fzero4();

    for (long row = blockIdx.x; row < rows_total; row += gridDim.x) {
        const long b = row / ((long)H * W);
        const int h = (int)((row / W) % H), w = (int)(row % W);
        __syncthreads();                                          // previous row's reads are done
        for (int e = tid; e < 10 * Dp; e += 256) {                // A: 9 neighbour dy rows (+ one zero piece) per d
            const int dd = e / 10, pc = e - dd * 10;
            bf16x8 v = zero8();
            if (pc < 9 && dd < D) {
                const int hh = h - (pc / 3 - 1), ww = w - (pc % 3 - 1);
                if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
                    v = ld8(dy + (((b * H + hh) * (long)W + ww) * D + dd) * 8);
            }
            *reinterpret_cast<bf16x8*>(Aimg + wr_off<WR_AS>(dd, pc * 8)) = v;
        }
        const bf16_t* xrow = x + ((b * H + h) * (long)W + w) * D * Cin;
        for (int e = tid; e < 8 * (Dp + 8); e += 256) {           // B: x row with a 4-row halo, the ones column, zeros
            const int rr = e >> 3, pc = e & 7, dd = rr - 4;
            bf16x8 v = zero8();
            if (dd >= 0 && dd < D) {
                if (pc < G8) v = ld8(xrow + (long)dd * Cin + pc * 8);
                else if (pc == G8) v[0] = (bf16_t)1.0f;
            }
            *reinterpret_cast<bf16x8*>(Bimg + wr_off<WR_BS>(rr, pc * 8)) = v;
        }
        __syncthreads();
        if (wave_on) {
            for (int ks = 0; ks < Dp / 32; ++ks) {
                const int r0 = 32 * ks + 8 * g + q;
                bf16x8 bf[3];
#pragma unroll
                for (int sd = 0; sd < 3; ++sd) {
                    const int rb = r0 + 4 + (sd - 1);
                    bf[sd] = cat44(wr_tr_read(Bimg, wr_off<WR_BS>(rb, 16 * wave + 4 * pp)),
                                   wr_tr_read(Bimg, wr_off<WR_BS>(rb + 4, 16 * wave + 4 * pp)));
                }
#pragma unroll
                for (int mt = 0; mt < WR_MT; ++mt) {
                    const bf16x8 a = cat44(wr_tr_read(Aimg, wr_off<WR_AS>(r0, 16 * mt + 4 * pp)),
                                           wr_tr_read(Aimg, wr_off<WR_AS>(r0 + 4, 16 * mt + 4 * pp)));
#pragma unroll
                    for (int sd = 0; sd < 3; ++sd) acc[sd][mt] = mfma16(a, bf[sd], acc[sd][mt]);
                }
            }
        }
    }
    // partial [block][3][80][64]
    float* my = part + (long)blockIdx.x * 3 * 80 * 64;
#pragma unroll
    for (int sd = 0; sd < 3; ++sd)
#pragma unroll
        for (int mt = 0; mt < WR_MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) my[((long)sd * 80 + 16 * mt + 4 * g + j) * 64 + 16 * wave + r] = acc[sd][mt][j];
}

static int wgrad_rows_grid(const MivpConvDesc* d) {
    const long rows = (long)d->B * d->dims[0] * d->dims[1];
    long wg = rows < 1024 ? rows : 1024;
    return (int)(wg < 1 ? 1 : wg);
}

static size_t wgrad_rows_lds(const MivpConvDesc* d) {
    const int dp = (d->dims[2] + 31) / 32 * 32;
    return (size_t)(dp / 4) * WR_AS * 128 + (size_t)((dp + 8) / 4) * WR_BS * 128;
}

extern "C" size_t mivp_conv3d_wgrad_rows_ws(const MivpConvDesc* d) {
    if (!d || d->Cout > 8 || d->Cin + 1 > 64 || d->Cin % 8 || wgrad_rows_lds(d) > 160 * 1024) return 0;
    return (size_t)wgrad_rows_grid(d) * 3 * 80 * 64;
}

extern "C" int mivp_conv3d_wgrad_rows(const MivpConvDesc* d, const void* x, const void* dy, int32_t dy_stride,
                                      float* part, float* gs, mivp_stream_t stream) {
    int rc = conv_checks(d);
    if (rc) return rc;
    MIVP_REQUIRE(x && dy && part && gs);
    if (d->Cout > 8 || dy_stride != 8 || d->Cin + 1 > 64 || d->Cin % 8) {
        mivp_set_error("conv3d_wgrad_rows: needs Cout <= 8 with dy padded to 8 channels, Cin % 8 == 0, Cin < 64");
        return MIVP_EUNSUPPORTED;
    }
    const size_t lds = wgrad_rows_lds(d);
    if (lds > 160 * 1024) { mivp_set_error("conv3d_wgrad_rows: D too long for the LDS row images"); return MIVP_EUNSUPPORTED; }
    const int grid = wgrad_rows_grid(d);
    auto kern = k_conv3d_wgrad_rows;
    MIVP_LDS_OPT_IN(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, *d, (const bf16_t*)x, (const bf16_t*)dy, part);
    rc = mivp_check_launch("conv3d_wgrad_rows");
    if (rc) return rc;
    return mivp_reduce_rows(part, (int64_t)grid, (int64_t)3 * 80 * 64, gs, stream);
}
