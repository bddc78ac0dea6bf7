// Patch merging (encoder downsample) and trilinear upsample + crop + concat (decoder).
// Reference: swin_transformer/down.py:21-53 ; swin_unetr/unet_blocks.py:31-35,72-73 and
// swin_unetr/swin_unetr.py:351-355 (nn.Upsample(trilinear, align_corners=False)).
#include "common.hpp"

// ---------------------------------------------------------------------------------------------
// K4  patch merging forward: gather 8 (or 4) neighbours -> LayerNorm(kC) -> Linear(kC -> Cout)
//   odd axes are zero-padded by one AT THE FRONT (down.py:25-28), also D when it is not merged;
//   concat order (h,w,d): 000 100 010 001 110 101 011 111   /  (h,w): 00 10 01 11
//   one wave = 16 output tokens; the gathered row goes straight into the MFMA B lane map.
// ---------------------------------------------------------------------------------------------
__device__ __constant__ int c_off8[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
__device__ __constant__ int c_off4[4][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}};

struct MergeTok { long b; int o[3]; bool live; };

MIVP_DEV MergeTok merge_token(const MivpMergeDesc& d, long t) {
    MergeTok m;
    const long ovol = (long)d.odims[0] * d.odims[1] * d.odims[2];
    const long T = (long)d.B * ovol;
    m.live = t < T;
    // token counts fit 32 bits (checked on the host): unsigned 32-bit divisions, a fraction of the cost of 64-bit ones
    const unsigned tt = m.live ? (unsigned)t : 0u;
    const unsigned ov = (unsigned)ovol, o12 = (unsigned)(d.odims[1] * d.odims[2]);
    const unsigned bb = tt / ov;
    unsigned rem = tt - bb * ov;
    m.b = bb;
    m.o[0] = (int)(rem / o12);
    rem -= (unsigned)m.o[0] * o12;
    m.o[1] = (int)(rem / (unsigned)d.odims[2]);
    m.o[2] = (int)(rem - (unsigned)m.o[1] * (unsigned)d.odims[2]);
    return m;
}

// input voxel offset of concat part `part` for output token m, or -1 when it falls in the front pad
MIVP_DEV long merge_src(const MivpMergeDesc& d, const MergeTok& m, int part) {
    int xc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int pad = d.dims[a] & 1;
        const bool merged = a < 2 || d.merge_last;
        const int off = d.merge_last ? c_off8[part][a] : c_off4[part][a];
        const int f = merged ? 2 * m.o[a] + off : m.o[a];
        xc[a] = f - pad;
    }
    if (xc[0] < 0 || xc[1] < 0 || xc[2] < 0) return -1;
    return ((m.b * d.dims[0] + xc[0]) * (long)d.dims[1] + xc[1]) * d.dims[2] + xc[2];
}

// Work split (round 3).  The first form gathered straight into the MFMA B lane map: `if (in bounds) load` per k-step, i.e.
// KS dependent memory round trips in a row (24 at kC = 768: ~25 of the 32 us of the bottleneck merge), each touching 64
// different rows per wave instruction.  Now:
//   1. lane 0..63 decode the workgroup's 64 tokens once (LDS table: first-neighbour voxel, front-pad flags);
//   2. the 64 x kC/8 sixteen-byte row pieces are loaded PIECE-MAJOR -- adjacent lanes, adjacent pieces of one source row --
//      unconditionally (clamped address, zeroed afterwards), several in flight per lane, into an LDS row image;
//   3. LayerNorm runs on the image with four lanes per row (rolled loops: a register-resident row made the compiler hoist
//      every gamma / beta load of the row, 384 registers at kC = 768);
//   4. each wave takes its 16 tokens' B fragments from the image and walks the output tiles as before (weight slabs through
//      LDS, split over gridDim.y where token groups are few).
template <int KS>
__global__ __launch_bounds__(256, KS >= 24 ? 1 : 2) void k_patch_merge_fwd(MivpMergeDesc d, const bf16_t* __restrict__ x,
                                                         const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                         const bf16_t* __restrict__ w, bf16_t* __restrict__ y) {
    constexpr int K = 32 * KS, ROWB = 2 * K + 16, PPR = K / 8, ROWS = 64;
    using WS = WeightSlabsRM<KS>;
    extern __shared__ __attribute__((aligned(16))) char smem_pm[];
    char* img = smem_pm;                                         // [64][kC] bf16 (+16 B per row): gathered, then normalised rows
    char* wsm = img + ROWS * ROWB;                               // two weight slabs
    long* tvox = reinterpret_cast<long*>(wsm + WS::BYTES);       // [64] first-neighbour voxel of the token
    int* tflag = reinterpret_cast<int*>(tvox + ROWS);            // [64] bit a: axis a starts in the front pad; bit 3: live
    float* gam = reinterpret_cast<float*>(tflag + ROWS);         // [kC] gamma | [kC] beta
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C;
    const long row0 = (long)blockIdx.x * ROWS;
    // output tiles are split over gridDim.y (the deep stages have only a few dozen token groups)
    const int n_tiles = (d.Cout + 15) / 16;
    const int per_y = (n_tiles + gridDim.y - 1) / gridDim.y;
    const int nt_lo = blockIdx.y * per_y, nt_hi = (nt_lo + per_y) < n_tiles ? (nt_lo + per_y) : n_tiles;
    // nothing below depends on the tokens: the first two weight slabs and gamma / beta travel while the rows are gathered
    WS ws, ws1;
    if (nt_lo < nt_hi) ws.fetch(w, K, 16 * nt_lo, d.Cout, K);
    if (nt_lo + 1 < nt_hi) ws1.fetch(w, K, 16 * (nt_lo + 1), d.Cout, K);
    for (int c = tid; c < K; c += 256) { gam[c] = ln_w[c]; gam[K + c] = ln_b[c]; }
    if (tid < ROWS) {
        const MergeTok m = merge_token(d, row0 + tid);
        int x0[3], flag = m.live ? 8 : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            x0[a] = ((a < 2 || d.merge_last) ? 2 * m.o[a] : m.o[a]) - (d.dims[a] & 1);
            flag |= x0[a] < 0 ? (1 << a) : 0;
        }
        tvox[tid] = ((m.b * d.dims[0] + x0[0]) * (long)d.dims[1] + x0[1]) * d.dims[2] + x0[2];
        tflag[tid] = flag;
    }
    __syncthreads();
    {   // ---- gather, piece-major ----
        // concat order (c_off8 / c_off4) as bit sets over `part`: the parts that take the second neighbour along h / w / d
        const unsigned hbits = d.merge_last ? 0xB2u : 0xAu, wbits = d.merge_last ? 0xD4u : 0xCu, dbits = d.merge_last ? 0xE8u : 0u;
        const int sw = d.dims[2], sh = d.dims[1] * d.dims[2];
        const FastDiv byC(C), byPPR(PPR);
        constexpr int NP = ROWS * PPR / 256, BATCH = NP % 12 == 0 ? 12 : (NP % 4 == 0 ? 4 : (NP % 2 == 0 ? 2 : 1));
        for (int i0 = 0; i0 < NP; i0 += BATCH) {
            bf16x8 v[BATCH];
            bool ok[BATCH];
            int dst[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int P = tid + 256 * (i0 + u);
                const int row = byPPR.div(P), c = 8 * (P - row * PPR);
                const int part = byC.div(c), ch = c - part * C;
                const int oh = (hbits >> part) & 1, ow = (wbits >> part) & 1, od = (dbits >> part) & 1;
                const int flag = tflag[row];
                // a front-pad axis is only in bounds for the second neighbour
                ok[u] = (flag & 8) && ((flag & 1) == 0 || oh) && ((flag & 2) == 0 || ow) && ((flag & 4) == 0 || od);
                const long vox = sel(ok[u], tvox[row] + oh * sh + ow * sw + od, 0L);
                v[u] = ld8(x + vox * C + ch);
                dst[u] = row * ROWB + 2 * c;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) *reinterpret_cast<bf16x8*>(img + dst[u]) = keep_if(v[u], ok[u]);
        }
    }
    if (nt_lo < nt_hi) ws.store(wsm, 0);
    if (nt_lo + 1 < nt_hi) ws1.store(wsm, 1);
    __syncthreads();
    {   // ---- LayerNorm over the kC-long rows: four lanes per row ----
        const int row = tid >> 2, sub = tid & 3;
        char* rp = img + row * ROWB + 16 * sub;
        const bool live = (tflag[row] & 8) != 0;
        float sum = 0.f;
#pragma unroll 4
        for (int i = 0; i < PPR / 4; ++i) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(rp + 64 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)raw[e];
        }
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
        const float mean = sum / (float)K;
        float var = 0.f;
#pragma unroll 4
        for (int i = 0; i < PPR / 4; ++i) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(rp + 64 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dv = (float)raw[e] - mean; var += dv * dv; }
        }
        var += __shfl_xor(var, 1); var += __shfl_xor(var, 2);
        const float rstd = rsqrtf(var / (float)K + d.ln_eps);
#pragma unroll 2
        for (int i = 0; i < PPR / 4; ++i) {
            const int c = 8 * (sub + 4 * i);
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(rp + 64 * i);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(gam + c), w1 = *reinterpret_cast<const f32x4*>(gam + c + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(gam + K + c), b1 = *reinterpret_cast<const f32x4*>(gam + K + c + 4);
            bf16x8 yv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                yv[e] = (bf16_t)(((float)raw[e] - mean) * rstd * w0[e] + b0[e]);
                yv[4 + e] = (bf16_t)(((float)raw[4 + e] - mean) * rstd * w1[e] + b1[e]);
            }
            *reinterpret_cast<bf16x8*>(rp + 64 * i) = keep_if(yv, live);     // a dead row stays zero (as the B operand)
        }
    }
    __syncthreads();
    const long t = row0 + 16 * wave + r;
    const bool live = (tflag[16 * wave + r] & 8) != 0;
    bf16x8 xb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xb[s] = *reinterpret_cast<const bf16x8*>(img + (16 * wave + r) * ROWB + 64 * s + 16 * g);
    // slab nt + 2 is requested before tile nt is multiplied and stored over slab nt after the barrier that ends the tile
    // (two tiles of flight time for a load; the products of a tile take a fraction of one memory round trip)
    for (int nt = nt_lo; nt < nt_hi; ++nt) {
        f32x4 acc = fzero4();
        const int cur = (nt - nt_lo) & 1;
        if (nt + 2 < nt_hi) ws.fetch(w, K, 16 * (nt + 2), d.Cout, K);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mfma16(WS::frag8(wsm, cur, s, r, 8 * g), xb[s], acc);
        const int n0 = 16 * nt + 4 * g;
        if (live && n0 < d.Cout) st4(y + t * d.Cout + n0, pack4(acc));
        __syncthreads();
        if (nt + 2 < nt_hi) ws.store(wsm, cur);
    }
}

extern "C" int mivp_patch_merge_fwd(const MivpMergeDesc* d, const void* x, const float* ln_w, const float* ln_b,
                                    const void* w, void* y, mivp_stream_t stream) {
    MIVP_REQUIRE(d && x && ln_w && ln_b && w && y);
    MIVP_REQUIRE(d->C % 8 == 0 && d->Cout % 4 == 0);
    for (int a = 0; a < 3; ++a) {
        const int padded = d->dims[a] + (d->dims[a] & 1);
        const bool merged = a < 2 || d->merge_last;
        MIVP_REQUIRE(d->odims[a] == (merged ? padded / 2 : padded));
    }
    const int kC = (d->merge_last ? 8 : 4) * d->C;
    const int KS = (kC + 31) / 32;
    const long T = (long)d->B * d->odims[0] * d->odims[1] * d->odims[2];
    MIVP_REQUIRE(T < (1L << 31));                               // merge_token decodes in 32 bits
    const unsigned gx = (unsigned)((T + 63) / 64);
    const int n_tiles = (d->Cout + 15) / 16;
    // every workgroup repeats the gather + LayerNorm prologue of its 64 tokens, so split the output tiles only as far as
    // ONE resident round of workgroups goes (the row image sets it: 1 per CU for KS = 24, 2 for KS = 12, ...)
    const size_t lds = (size_t)64 * (64 * KS + 16) + 2048 * KS + 64 * 12 + 256 * KS;
    long per_cu = (long)(160 * 1024 / lds);
    if (per_cu > 8) per_cu = 8;
    const long resident = 256L * per_cu;
    int ny = (int)(resident / gx);
    if (ny > n_tiles) ny = n_tiles;
    if (ny < 1) ny = 1;
    while (n_tiles % ny != 0) --ny;                              // equal shares
    const dim3 grid(gx, (unsigned)ny);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_PM(K)                                                                                                          \
    do {                                                                                                                      \
        MIVP_LDS_OPT_IN(k_patch_merge_fwd<K>, lds);                                                                           \
        hipLaunchKernelGGL((k_patch_merge_fwd<K>), grid, dim3(256), lds, st, *d, (const bf16_t*)x, ln_w, ln_b,                \
                           (const bf16_t*)w, (bf16_t*)y);                                                                     \
    } while (0)
    switch (KS) {
        case 1: LAUNCH_PM(1); break;
        case 2: LAUNCH_PM(2); break;
        case 4: LAUNCH_PM(4); break;
        case 6: LAUNCH_PM(6); break;
        case 8: LAUNCH_PM(8); break;
        case 12: LAUNCH_PM(12); break;
        case 24: LAUNCH_PM(24); break;
        default: mivp_set_error("patch_merge_fwd: k*C/32 not in {1,2,4,6,8,12,24}"); return MIVP_EUNSUPPORTED;
    }
#undef LAUNCH_PM
    return mivp_check_launch("patch_merge_fwd");
}

// ---------------------------------------------------------------------------------------------
// trilinear upsample (align_corners=False, scale 1 or 2 per axis) + crop to the skip's dims + concat
//   source index as ATen's area_pixel_compute_source_index: max(0, (o + 0.5) / scale - 0.5)
// ---------------------------------------------------------------------------------------------
struct Lerp { int i0, i1; float w0, w1; };

MIVP_DEV Lerp lerp_axis(int o, int scale, int n_in, int align = 0) {
    Lerp l;
    if (scale == 1) { l.i0 = l.i1 = o; l.w0 = 1.f; l.w1 = 0.f; return l; }
    if (!align) {
        // half-pixel centres, scale 2: source (o + 0.5) / 2 - 0.5 clamped at 0 = 0 | 0.25, 0.75 | 1.25, 1.75 | ... in integer form
        // (the same values bit for bit -- 0.25 and 0.75 are exact -- in 6 instructions instead of ~15: these kernels are bound
        // by instruction issue)
        l.i0 = max(o - 1, 0) >> 1;
        if (l.i0 > n_in - 1) l.i0 = n_in - 1;
        l.i1 = l.i0 + (l.i0 < n_in - 1 ? 1 : 0);
        l.w1 = o == 0 ? 0.f : ((o & 1) ? 0.25f : 0.75f);
        l.w0 = 1.f - l.w1;
        return l;
    }
    float src;
    if (align) {                                              // align_corners=True: o * (n_in - 1) / (n_out - 1), n_out = 2 n_in
        src = (float)o * ((float)(n_in - 1) / (float)(2 * n_in - 1));
    } else {
        src = ((float)o + 0.5f) * 0.5f - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    l.i0 = (int)src;
    if (l.i0 > n_in - 1) l.i0 = n_in - 1;
    l.i1 = l.i0 + (l.i0 < n_in - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

// One workgroup per output row (b, oh, ow).  The h/w interpolation (indices, weights, the four source rows) is uniform
// for the block, so the four rows are combined ONCE into an f32 row image in LDS -- hw[id][Cx] = sum_ij w_ij x_ij -- with
// coalesced loads, and an output item (od, channel group) is two LDS reads and one lerp.  Read per item straight from
// global memory the stencil was 8 sixteen-byte loads per 16 output bytes (680 MB of L1 traffic for the 127 MB concat tensor
// of the last decoder stage: the texture-address unit, not HBM, set the kernel's time); staged, every source piece is
// loaded once per workgroup (4x fewer vector-memory instructions).  No weight-dependent branches: clamped neighbours are
// always in bounds, their weight is simply 0.
// AFF: y = act(scale[c] * v + shift[c]) applied to the bf16-rounded concat value v (the BatchNorm affine + LeakyReLU that
// follows the concat in SwinUpBlock, unet_blocks.py:72-75) -- bit-identical to k_upcat_fwd followed by k_affine_act,
// without the round trip of the concat tensor (statistics: k_upcat_stats).
struct UpRow { int b, oh, ow; };                              // an output row; the kernels take it from their 3-D grids
// hw[id][Cx] (f32, LDS) <- the (h, w)-interpolated source row of output row u.  Two halves so that the loads of the NEXT
// row can travel while the current one is consumed: load() issues every load of the thread unconditionally (NIT pieces x
// 4 rows, clamped index), store() combines and writes them; `base` walks rows longer than 256 NIT pieces.
template <int NIT>
struct UpStage {
    bf16x8 a[NIT], b[NIT], c[NIT], e[NIT];
    float w00, w01, w10, w11;
    MIVP_DEV void load(const MivpUpcatDesc& d, const bf16_t* __restrict__ x, UpRow u, int tid, int base = 0) {
        const int n = d.idims[2] * (d.Cx / 8);
        const Lerp lh = lerp_axis(u.oh, d.scale[0], d.idims[0], d.align_corners);
        const Lerp lw = lerp_axis(u.ow, d.scale[1], d.idims[1], d.align_corners);
        // four uniform row bases (scalar registers) + one 32-bit lane offset per piece: no per-lane 64-bit address arithmetic
        const long in_row = (long)d.idims[2] * d.Cx;
        const bf16_t* xb = x + (long)u.b * d.idims[0] * d.idims[1] * in_row;
        const bf16_t* r00 = xb + (long)(lh.i0 * d.idims[1] + lw.i0) * in_row;
        const bf16_t* r01 = xb + (long)(lh.i0 * d.idims[1] + lw.i1) * in_row;
        const bf16_t* r10 = xb + (long)(lh.i1 * d.idims[1] + lw.i0) * in_row;
        const bf16_t* r11 = xb + (long)(lh.i1 * d.idims[1] + lw.i1) * in_row;
        w00 = lh.w0 * lw.w0; w01 = lh.w0 * lw.w1; w10 = lh.w1 * lw.w0; w11 = lh.w1 * lw.w1;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {                          // the row is contiguous: piece `it` at element 8 it
            const unsigned it8 = 8u * (unsigned)min(base + tid + 256 * k, n - 1);
            a[k] = ld8(r00 + it8); b[k] = ld8(r01 + it8); c[k] = ld8(r10 + it8); e[k] = ld8(r11 + it8);
        }
    }
    MIVP_DEV void store(const MivpUpcatDesc& d, float* hw, int tid, int base = 0) const {
        const int n = d.idims[2] * (d.Cx / 8);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = base + tid + 256 * k;
            if (it < n) {
                f32x4 lo, hi;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] = w00 * (float)a[k][i] + w01 * (float)b[k][i] + w10 * (float)c[k][i] + w11 * (float)e[k][i];
                    hi[i] = w00 * (float)a[k][4 + i] + w01 * (float)b[k][4 + i] + w10 * (float)c[k][4 + i] + w11 * (float)e[k][4 + i];
                }
                *reinterpret_cast<f32x4*>(hw + 8 * it) = lo;
                *reinterpret_cast<f32x4*>(hw + 8 * it + 4) = hi;
            }
        }
    }
};
// the bf16-rounded upsampled values of (od, channel group cg) from the staged row
MIVP_DEV bf16x8 upcat_item(const MivpUpcatDesc& d, const float* hw, int od, int cg) {
    const Lerp ld = lerp_axis(od, d.scale[2], d.idims[2], d.align_corners);
    const float* p0 = hw + ld.i0 * d.Cx + cg * 8;
    const float* p1 = hw + ld.i1 * d.Cx + cg * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(p0), a1 = *reinterpret_cast<const f32x4*>(p0 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p1), b1 = *reinterpret_cast<const f32x4*>(p1 + 4);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[i] = (bf16_t)(ld.w0 * a0[i] + ld.w1 * b0[i]);
        o[4 + i] = (bf16_t)(ld.w0 * a1[i] + ld.w1 * b1[i]);
    }
    return o;
}

template <bool AFF>
__global__ __launch_bounds__(256) void k_upcat_fwd(MivpUpcatDesc d, const bf16_t* __restrict__ x,
                                                   const bf16_t* __restrict__ skip, bf16_t* __restrict__ y,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   int lrelu) {
    extern __shared__ __attribute__((aligned(16))) char smem_up[];
    float* hw = reinterpret_cast<float*>(smem_up);            // [id][Cx] f32
    float* ssc = hw + d.idims[2] * d.Cx;                      // AFF: [Ct] scale | [Ct] shift
    const int Ct = d.Cx + d.Cs, G = Ct / 8, Gx = d.Cx / 8;
    const int OD = d.odims[2];
    const UpRow urow{(int)blockIdx.z, (int)blockIdx.y, (int)blockIdx.x};       // grid (OW, OH, B): no divisions
    const int row = (urow.b * d.odims[0] + urow.oh) * d.odims[1] + urow.ow;
    if (AFF)
        for (int c = threadIdx.x; c < Ct; c += 256) { ssc[c] = scale[c]; ssc[Ct + c] = shift[c]; }
    for (int base = 0; base < d.idims[2] * Gx; base += 256 * 2) {
        UpStage<2> st;
        st.load(d, x, urow, threadIdx.x, base);
        st.store(d, hw, threadIdx.x, base);
    }
    __syncthreads();
    auto affine = [&](bf16x8 v, int c0) -> bf16x8 {
        bf16x8 o;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(ssc + c0), s1 = *reinterpret_cast<const f32x4*>(ssc + c0 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(ssc + Ct + c0), h1 = *reinterpret_cast<const f32x4*>(ssc + Ct + c0 + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float f = (float)v[i] * (i < 4 ? s0[i & 3] : s1[i & 3]) + (i < 4 ? h0[i & 3] : h1[i & 3]);
            if (lrelu) f = f > 0.f ? f : 0.01f * f;
            o[i] = (bf16_t)f;
        }
        return o;
    };
    bf16_t* yrow = y + (long)row * OD * Ct;
    const bf16_t* srow = skip ? skip + (long)row * OD * d.Cs : nullptr;
    // items ordered interpolation first, skip copies last: waves are homogeneous (a mixed wave runs both paths)
    const int nI = OD * Gx, Gs = G - Gx;
    for (int it = threadIdx.x; it < OD * G; it += 256) {
        if (it >= nI) {
            const int j = it - nI, od = j / Gs, cg = j - od * Gs;
            bf16x8 sv = ld8(srow + od * d.Cs + cg * 8);
            if (AFF) sv = affine(sv, (Gx + cg) * 8);
            st8(yrow + od * Ct + (Gx + cg) * 8, sv);
            continue;
        }
        const int od = it / Gx, cg = it - od * Gx;
        bf16x8 o = upcat_item(d, hw, od, cg);
        if (AFF) o = affine(o, cg * 8);
        st8(yrow + od * Ct + cg * 8, o);
    }
}

// Per-channel sum / sum of squares of the upsample + concat tensor WITHOUT forming it (the BatchNorm statistics of
// SwinUpBlock's norm_concat): the same bf16-rounded values k_upcat_fwd would write.  Workgroups stride over output rows;
// a thread keeps ONE channel group for the whole kernel (waves 0-2:
// interpolated channels, wave 3: skip channels, so waves are homogeneous), walks the row's d positions and accumulates in
// registers; the block's partials meet in LDS in a fixed order -> part [gridDim.x][2 * Ct] as mivp_bn_stats writes them
// (mivp_bn_finalize reduces the blocks).
template <int NIT>
__global__ __launch_bounds__(256, NIT <= 2 ? 4 : (NIT <= 4 ? 3 : 2)) void k_upcat_stats(MivpUpcatDesc d, const bf16_t* __restrict__ x,
                                                     const bf16_t* __restrict__ skip, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem_up[];
    float* lsum = reinterpret_cast<float*>(smem_up);          // [256][16]
    float* hwbuf = lsum + 256 * 16;                           // [id][Cx] f32
    const int Ct = d.Cx + d.Cs, Gx = d.Cx / 8, Gs = d.Cs / 8;
    const int OD = d.odims[2];
    const int tid = threadIdx.x;
    // The kernel is bound by instruction issue (~600 instructions per wave and row: profiles/README.md), so the roles are
    // balanced by instruction count: an interpolated item costs ~75 instructions, a copied one ~25 -- three waves
    // interpolate, one copies.
    const int nI = Gs ? 192 : 256;
    const int perx = nI / Gx, pers = Gs ? 64 / Gs : 0;
    const bool interp = tid < perx * Gx, copy = tid >= nI && tid - nI < pers * Gs;
    const int cg = interp ? tid % Gx : (copy ? (tid - nI) % Gs : 0);
    const int od0 = interp ? tid / Gx : (copy ? (tid - nI) / Gs : 0);
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    // Occupancy, not software pipelining, hides the memory latency here (a row's loads held in registers across the
    // previous row's arithmetic cost 146 VGPRs = 3 workgroups per CU, and a launch then lasted as many load round trips as
    // a workgroup has rows): per row every load of the thread -- source pieces and skip pieces -- is issued at once, one
    // image buffer, two barriers.
    constexpr int NSK = 6;                                    // skip pieces per thread and batch
    float* hw = hwbuf;
    // grid (nx, OH, B): a workgroup keeps (b, oh) and walks ow = blockIdx.x, blockIdx.x + nx, ... (no divisions, the h
    // interpolation is loop-invariant); its partial row is ((b * OH + oh) * nx + blockIdx.x)
    const int OWd = d.odims[1];
    const long row_base = ((long)blockIdx.z * d.odims[0] + blockIdx.y) * OWd;
    for (int ow = blockIdx.x; ow < OWd; ow += gridDim.x) {
        const long row = row_base + ow;
        UpStage<NIT> st;
        st.load(d, x, UpRow{(int)blockIdx.z, (int)blockIdx.y, ow}, tid);
        const bf16_t* srow = skip + row * OD * d.Cs;
        bf16x8 sv[NSK];
        if (copy) {
#pragma unroll
            for (int u = 0; u < NSK; ++u) sv[u] = ld8(srow + min(od0 + u * pers, OD - 1) * d.Cs + cg * 8);
        }
        st.store(d, hw, tid);
        __syncthreads();
        if (interp) {
#pragma unroll 2
            for (int od = od0; od < OD; od += perx) {
                const bf16x8 v = upcat_item(d, hw, od, cg);
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float f = (float)v[i]; s1[i] += f; s2[i] += f * f; }
            }
        } else if (copy) {
#pragma unroll
            for (int u = 0; u < NSK; ++u) {
                const bf16x8 v = keep_if(sv[u], od0 + u * pers < OD);
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float f = (float)v[i]; s1[i] += f; s2[i] += f * f; }
            }
            for (int odb = od0 + NSK * pers; odb < OD; odb += NSK * pers) {  // (rows longer than one batch: further batches)
                bf16x8 tv[NSK];
#pragma unroll
                for (int u = 0; u < NSK; ++u) tv[u] = keep_if(ld8(srow + min(odb + u * pers, OD - 1) * d.Cs + cg * 8), odb + u * pers < OD);
#pragma unroll
                for (int u = 0; u < NSK; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float f = (float)tv[u][i]; s1[i] += f; s2[i] += f * f; }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { lsum[tid * 16 + i] = s1[i]; lsum[tid * 16 + 8 + i] = s2[i]; }
    __syncthreads();
    for (int o = tid; o < 2 * Ct; o += 256) {
        const int which = o / Ct, c = o - which * Ct, i = c & 7;
        float acc = 0.f;
        if (c < d.Cx) { for (int t = c >> 3; t < perx * Gx; t += Gx) acc += lsum[t * 16 + which * 8 + i]; }
        else { for (int t = nI + ((c - d.Cx) >> 3); t < nI + pers * Gs; t += Gs) acc += lsum[t * 16 + which * 8 + i]; }
        part[(((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2 * Ct + o] = acc;
    }
}

extern "C" int mivp_upcat_fwd(const MivpUpcatDesc* d, const void* x, const void* skip, void* y, mivp_stream_t stream) {
    MIVP_REQUIRE(d && x && y);
    MIVP_REQUIRE(d->Cx % 8 == 0 && d->Cs % 8 == 0 && (d->Cs == 0 || skip));
    for (int a = 0; a < 3; ++a) {
        MIVP_REQUIRE(d->scale[a] == 1 || d->scale[a] == 2);
        MIVP_REQUIRE(d->odims[a] > 0 && d->odims[a] <= d->scale[a] * d->idims[a]);
    }
    const long rows = (long)d->B * d->odims[0] * d->odims[1];
    MIVP_REQUIRE(rows < (1L << 31));
    const size_t lds = sizeof(float) * (size_t)d->idims[2] * d->Cx;          // the (h, w)-combined source row
    MIVP_REQUIRE(lds <= 160 * 1024);
    MIVP_LDS_OPT_IN(k_upcat_fwd<false>, lds);
    MIVP_REQUIRE(d->odims[0] < 65536 && d->B < 65536);
    hipLaunchKernelGGL(k_upcat_fwd<false>, dim3(d->odims[1], d->odims[0], d->B), dim3(256), lds, (hipStream_t)stream, *d, (const bf16_t*)x,
                       (const bf16_t*)skip, (bf16_t*)y, nullptr, nullptr, 0);
    return mivp_check_launch("upcat_fwd");
}

static int upcat_checks(const MivpUpcatDesc* d, const void* skip) {
    MIVP_REQUIRE(d->Cx % 8 == 0 && d->Cs % 8 == 0 && d->Cx > 0 && (d->Cs == 0 || skip));
    for (int a = 0; a < 3; ++a) {
        MIVP_REQUIRE(d->scale[a] == 1 || d->scale[a] == 2);
        MIVP_REQUIRE(d->odims[a] > 0 && d->odims[a] <= d->scale[a] * d->idims[a]);
    }
    MIVP_REQUIRE((long)d->B * d->odims[0] * d->odims[1] < (1L << 31));
    return MIVP_OK;
}

extern "C" int mivp_upcat_affine_fwd(const MivpUpcatDesc* d, const void* x, const void* skip, const float* scale,
                                     const float* shift, int32_t lrelu, void* y, mivp_stream_t stream) {
    MIVP_REQUIRE(d && x && y && scale && shift);
    int rc = upcat_checks(d, skip);
    if (rc) return rc;
    const size_t lds = sizeof(float) * ((size_t)d->idims[2] * d->Cx + 2 * (size_t)(d->Cx + d->Cs));
    MIVP_REQUIRE(lds <= 160 * 1024);
    MIVP_LDS_OPT_IN(k_upcat_fwd<true>, lds);
    MIVP_REQUIRE(d->odims[0] < 65536 && d->B < 65536);
    hipLaunchKernelGGL(k_upcat_fwd<true>, dim3(d->odims[1], d->odims[0], d->B), dim3(256), lds, (hipStream_t)stream, *d, (const bf16_t*)x,
                       (const bf16_t*)skip, (bf16_t*)y, scale, shift, (int)lrelu);
    return mivp_check_launch("upcat_affine_fwd");
}

extern "C" int mivp_upcat_stats(const MivpUpcatDesc* d, const void* x, const void* skip, int32_t nblk, float* part,
                                mivp_stream_t stream) {
    MIVP_REQUIRE(d && x && part && nblk > 0);
    int rc = upcat_checks(d, skip);
    if (rc) return rc;
    MIVP_REQUIRE(d->Cx / 8 <= (d->Cs ? 192 : 256) && d->Cs / 8 <= 64);
    // nblk = nx * OH * B partial rows (ops.upcat_stats picks nx): nx workgroups share an output (b, oh) line
    MIVP_REQUIRE(nblk % (d->B * d->odims[0]) == 0 && nblk / (d->B * d->odims[0]) <= d->odims[1]);
    MIVP_REQUIRE(d->odims[0] < 65536 && d->B < 65536);
    const dim3 grid3((unsigned)(nblk / (d->B * d->odims[0])), (unsigned)d->odims[0], (unsigned)d->B);
    const size_t lds = sizeof(float) * (256 * 16 + (size_t)d->idims[2] * d->Cx);
    const int pieces = d->idims[2] * (d->Cx / 8);                            // of a source row: all of them live in registers
    MIVP_REQUIRE(lds <= 160 * 1024 && pieces <= 256 * 8);
#define L_UPSTATS(N)                                                                                                           \
    do {                                                                                                                      \
        MIVP_LDS_OPT_IN(k_upcat_stats<N>, lds);                                                                               \
        hipLaunchKernelGGL(k_upcat_stats<N>, grid3, dim3(256), lds, (hipStream_t)stream, *d, (const bf16_t*)x,                 \
                           (const bf16_t*)skip, part);                                                                        \
    } while (0)
    if (pieces <= 512) L_UPSTATS(2);
    else if (pieces <= 1024) L_UPSTATS(4);
    else L_UPSTATS(8);
#undef L_UPSTATS
    return mivp_check_launch("upcat_stats");
}

// backward: dx[i] = sum over output positions o whose stencil touches i of weight(o, i) * dy[o] (first Cx
// channels of dy); dskip = channel slice copy.  Gather form: per axis at most 4 candidate outputs.
__global__ __launch_bounds__(256) void k_upcat_bwd_x(MivpUpcatDesc d, const bf16_t* __restrict__ dy,
                                                     bf16_t* __restrict__ dx) {
    const int Ct = d.Cx + d.Cs, Gx = d.Cx / 8;
    const long ivol = (long)d.idims[0] * d.idims[1] * d.idims[2];
    const long items = (long)d.B * ivol * Gx;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (long it = gtid; it < items; it += stride) {
        const unsigned vox = (unsigned)it / (unsigned)Gx;     // host checks items < 2^31: 32-bit decode
        const int cg = (int)((unsigned)it - vox * (unsigned)Gx);
        const unsigned b = vox / (unsigned)ivol;
        unsigned rem = vox - b * (unsigned)ivol;
        const unsigned i12 = (unsigned)(d.idims[1] * d.idims[2]);
        const int ih = (int)(rem / i12);
        rem -= (unsigned)ih * i12;
        const int iw = (int)(rem / (unsigned)d.idims[2]);
        const int id = (int)(rem - (unsigned)iw * (unsigned)d.idims[2]);
        const int ic[3] = {ih, iw, id};
        int cand[3][8];
        float cw[3][8];
        int ncand[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ncand[a] = 0;
            // outputs whose two-point stencil can touch input ic: 2i-1 .. 2i+2 (half-pixel centres), 2i-2 .. 2i+4 (align_corners)
            const int lo = d.scale[a] == 1 ? ic[a] : 2 * ic[a] - (d.align_corners ? 2 : 1);
            const int hi = d.scale[a] == 1 ? ic[a] : 2 * ic[a] + (d.align_corners ? 4 : 2);
            for (int o = lo; o <= hi; ++o) {
                if (o < 0 || o >= d.odims[a]) continue;
                const Lerp l = lerp_axis(o, d.scale[a], d.idims[a], d.align_corners);
                float wgt = 0.f;
                if (l.i0 == ic[a]) wgt += l.w0;
                if (l.i1 == ic[a]) wgt += l.w1;
                if (wgt != 0.f) { cand[a][ncand[a]] = o; cw[a][ncand[a]] = wgt; ++ncand[a]; }
            }
        }
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int a = 0; a < ncand[0]; ++a)
            for (int bb = 0; bb < ncand[1]; ++bb)
                for (int c = 0; c < ncand[2]; ++c) {
                    const float wgt = cw[0][a] * cw[1][bb] * cw[2][c];
                    const long ov = (((long)b * d.odims[0] + cand[0][a]) * (long)d.odims[1] + cand[1][bb]) * d.odims[2] + cand[2][c];
                    const bf16x8 v = ld8(dy + ov * Ct + cg * 8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += wgt * (float)v[i];
                }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)acc[i];
        st8(dx + (long)vox * d.Cx + cg * 8, o);
    }
}

// align_corners == 0 form of k_upcat_bwd_x: an input voxel is touched by at most the four outputs 2i-1 .. 2i+2 per axis
// (one output for a scale-1 axis).  Candidate indices and weights are static arrays, every load is unconditional (a
// candidate outside the cropped output reads a clamped index with weight 0): the 64 loads of an item are independent
// instead of 64 serial round trips behind dynamically indexed candidate lists.
__global__ __launch_bounds__(256) void k_upcat_bwd_x4(MivpUpcatDesc d, const bf16_t* __restrict__ dy,
                                                      bf16_t* __restrict__ dx) {
    const int Ct = d.Cx + d.Cs, Gx = d.Cx / 8;
    const long ivol = (long)d.idims[0] * d.idims[1] * d.idims[2];
    const long items = (long)d.B * ivol * Gx;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (long it = gtid; it < items; it += stride) {
        const unsigned vox = (unsigned)it / (unsigned)Gx;     // host checks items < 2^31: 32-bit decode
        const int cg = (int)((unsigned)it - vox * (unsigned)Gx);
        const unsigned b = vox / (unsigned)ivol;
        unsigned rem = vox - b * (unsigned)ivol;
        const unsigned i12 = (unsigned)(d.idims[1] * d.idims[2]);
        int ic[3];
        ic[0] = (int)(rem / i12);
        rem -= (unsigned)ic[0] * i12;
        ic[1] = (int)(rem / (unsigned)d.idims[2]);
        ic[2] = (int)(rem - (unsigned)ic[1] * (unsigned)d.idims[2]);
        int co[3][4];
        float cw[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool two = d.scale[a] == 2;
                const int o = two ? 2 * ic[a] - 1 + k : ic[a];
                const bool valid = (two || k == 0) && o >= 0 && o < d.odims[a];
                float wgt = 0.f;
                if (valid) {
                    const Lerp l = lerp_axis(o, d.scale[a], d.idims[a], 0);
                    wgt = (l.i0 == ic[a] ? l.w0 : 0.f) + (l.i1 == ic[a] ? l.w1 : 0.f);
                }
                co[a][k] = valid ? o : 0;
                cw[a][k] = wgt;
            }
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        const bf16_t* base = dy + cg * 8;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float w01 = cw[0][a] * cw[1][bb];
                const long row = (((long)b * d.odims[0] + co[0][a]) * d.odims[1] + co[1][bb]) * d.odims[2];
                bf16x8 v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = ld8(base + (row + co[2][c]) * Ct);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float wgt = w01 * cw[2][c];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += wgt * (float)v[c][i];
                }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)acc[i];
        st8(dx + (long)vox * d.Cx + cg * 8, o);
    }
}

__global__ __launch_bounds__(256) void k_upcat_bwd_skip(MivpUpcatDesc d, const bf16_t* __restrict__ dy,
                                                        bf16_t* __restrict__ dskip) {
    const int Ct = d.Cx + d.Cs, Gs = d.Cs / 8;
    const long ovol = (long)d.odims[0] * d.odims[1] * d.odims[2];
    const long items = (long)d.B * ovol * Gs;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (long it = gtid; it < items; it += stride) {
        const unsigned vu = (unsigned)it / (unsigned)Gs;       // host checks items < 2^31
        const int cg = (int)((unsigned)it - vu * (unsigned)Gs);
        const long vox = vu;
        st8(dskip + vox * d.Cs + cg * 8, ld8(dy + vox * Ct + d.Cx + cg * 8));
    }
}

extern "C" int mivp_upcat_bwd(const MivpUpcatDesc* d, const void* dy, void* dx, void* dskip, mivp_stream_t stream) {
    MIVP_REQUIRE(d && dy);
    MIVP_REQUIRE(d->Cx % 8 == 0 && d->Cs % 8 == 0);
    if (dx) {
        const long items = (long)d->B * d->idims[0] * d->idims[1] * d->idims[2] * (d->Cx / 8);
        MIVP_REQUIRE(items < (1L << 31));                        // 32-bit decode in the kernel
        const unsigned grid = (unsigned)((items + 255) / 256 > 8192 ? 8192 : (items + 255) / 256);
        if (d->align_corners)
            hipLaunchKernelGGL(k_upcat_bwd_x, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, (const bf16_t*)dy, (bf16_t*)dx);
        else
            hipLaunchKernelGGL(k_upcat_bwd_x4, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, (const bf16_t*)dy, (bf16_t*)dx);
        int rc = mivp_check_launch("upcat_bwd_x");
        if (rc) return rc;
    }
    if (dskip && d->Cs > 0) {
        const long items = (long)d->B * d->odims[0] * d->odims[1] * d->odims[2] * (d->Cs / 8);
        MIVP_REQUIRE(items < (1L << 31));
        const unsigned grid = (unsigned)((items + 255) / 256 > 8192 ? 8192 : (items + 255) / 256);
        hipLaunchKernelGGL(k_upcat_bwd_skip, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, (const bf16_t*)dy,
                           (bf16_t*)dskip);
        return mivp_check_launch("upcat_bwd_skip");
    }
    return MIVP_OK;
}

// ---------------------------------------------------------------------------------------------
// K4 backward: dy [T][Cout] -> dx.  dYn = dy W (A = W^T rows = concat channel) -> LayerNorm backward over the kC-long
// gathered row -> scatter to the 8 (4) source voxels.
// The two row sums of the LayerNorm backward need no GEMM: with n = gamma*xhat + beta and y = W n (the forward output),
//     s1 = sum_c dYn_c gamma_c        = dy . (W gamma)
//     s2 = sum_c dYn_c gamma_c xhat_c = dy . (y - W beta)
// so they cost O(Cout) per token from dy, the saved forward output y and two weight-only vectors; dYn itself is then
// formed ONCE, tile by tile, and the tiles are split over gridDim.y (the deep stages have only a few dozen token
// groups and hundreds of MFMAs per token).
// ---------------------------------------------------------------------------------------------
constexpr int MB_BATCH = 4;                                    // row pieces in flight per lane in the statistics pass
template <int NS>
__global__ __launch_bounds__(256) void k_patch_merge_bwd(MivpMergeDesc d, const bf16_t* __restrict__ dy,
                                                         const bf16_t* __restrict__ x, const bf16_t* __restrict__ yfwd,
                                                         const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                         const float* __restrict__ wgam, const float* __restrict__ wbet,
                                                         const bf16_t* __restrict__ w_t, bf16_t* __restrict__ dx,
                                                         bf16_t* __restrict__ wg_dn, bf16_t* __restrict__ wg_x) {
    // wg_dn / wg_x (optional, weight-gradient mode): [T][kC] gradient w.r.t. the LayerNorm output and the gathered
    // (front-padded) LayerNorm input rows, the operands of mivp_ln_wgrad / mivp_gemm_tn
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int C = d.C, Cout = d.Cout;
    const int kC = (d.merge_last ? 8 : 4) * C;
    const int n_ct = (kC + 15) / 16;
    const long t = ((long)blockIdx.x * 4 + wave) * 16 + r;
    const MergeTok m = merge_token(d, t);

    bf16x8 dyb[NS];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = 32 * s + 8 * g;
        bf16x8 v = zero8();
        if (m.live && c < Cout) {
            v = ld8(dy + t * Cout + c);
            const bf16x8 yv = ld8(yfwd + t * Cout + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gdy = (float)v[i];
                s1 += gdy * wgam[c + i];
                s2 += gdy * ((float)yv[i] - wbet[c + i]);
            }
        }
        dyb[s] = v;
    }
    const float m1 = col_sum(s1) / (float)kC, m2 = col_sum(s2) / (float)kC;

    // Row statistics in ONE pass of 16-byte loads (sum and sum of squares in fp32; the rows are at most 1536 bf16 values
    // of O(1) magnitude).  The loads are unconditional -- a dead piece reads offset 0 and is masked afterwards -- and go
    // out MB_BATCH at a time: with a branch per piece they serialised into one memory round trip each, which was most of this
    // kernel's time at the deep stages.
    auto piece_off = [&](int c0, bool& ok) -> long {             // element offset of concat channel c0 of this lane's token
        const int cc = c0 < kC ? c0 : 0;
        const int part = cc / C, ch = cc - part * C;
        const long src = merge_src(d, m, part);
        ok = m.live && c0 < kC && src >= 0;
        return ok ? src * C + ch : 0;
    };
    float sum = 0.f, sq = 0.f;
    const int n_k8 = (kC + 31) / 32;
    for (int s0 = 0; s0 < n_k8; s0 += MB_BATCH) {
        bf16x8 raw[MB_BATCH];
        bool ok[MB_BATCH];
#pragma unroll
        for (int u = 0; u < MB_BATCH; ++u) {
            const int c = 32 * (s0 + u) + 8 * g;
            const long off = piece_off(s0 + u < n_k8 ? c : kC, ok[u]);
            raw[u] = ld8(x + off);
        }
#pragma unroll
        for (int u = 0; u < MB_BATCH; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = ok[u] ? (float)raw[u][i] : 0.f;
                sum += v;
                sq += v * v;
            }
    }
    const float mean = col_sum(sum) / (float)kC;
    float var = col_sum(sq) / (float)kC - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + d.ln_eps);

    const int per_y = (n_ct + gridDim.y - 1) / gridDim.y;
    const int ct_lo = blockIdx.y * per_y, ct_hi = (ct_lo + per_y) < n_ct ? (ct_lo + per_y) : n_ct;
    // weight slabs shared by the four waves through LDS (WeightSlabs, common.hpp)
    using WS = WeightSlabsRM<NS>;
    __shared__ __attribute__((aligned(16))) char wsm[WS::BYTES];
    WS ws;
    if (ct_lo < ct_hi) { ws.fetch(w_t, Cout, 16 * ct_lo, kC, Cout); ws.store(wsm, 0); __syncthreads(); }
    for (int ct = ct_lo; ct < ct_hi; ++ct) {
        f32x4 gy = fzero4();
        const int cur = (ct - ct_lo) & 1;
        const int c0 = 16 * ct + 4 * g;
        bool okx;
        const long xo = piece_off(c0, okx);
        const bf16x4 xraw = ld4(x + xo);                          // in flight under the MFMAs
        const long so = okx ? xo : -1;
        if (ct + 1 < ct_hi) ws.fetch(w_t, Cout, 16 * (ct + 1), kC, Cout);
#pragma unroll
        for (int s = 0; s < NS; ++s) gy = mfma16(WS::frag8(wsm, cur, s, r, 8 * g), dyb[s], gy);
        if (ct + 1 < ct_hi) ws.store(wsm, cur ^ 1);
        __syncthreads();
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = okx ? (float)xraw[j] : 0.f;
        if (wg_dn && m.live && c0 < kC) {
            st4(wg_dn + t * kC + c0, pack4(gy));
            st4(wg_x + t * kC + c0, pack4(v));
        }
        if (c0 < kC && so >= 0) {
            f32x4 out;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = rstd * (gy[j] * ln_w[c0 + j] - m1 - (v[j] - mean) * rstd * m2);
            st4(dx + so, pack4(out));
        }
    }
}

extern "C" int mivp_patch_merge_bwd(const MivpMergeDesc* d, const void* dy, const void* x, const void* yfwd,
                                    const float* ln_w, const float* ln_b, const float* wgam, const float* wbet,
                                    const void* w_t, void* dx, void* wg_dn, void* wg_x, mivp_stream_t stream) {
    MIVP_REQUIRE(d && dy && x && yfwd && ln_w && ln_b && wgam && wbet && w_t && dx);
    MIVP_REQUIRE((wg_dn == nullptr) == (wg_x == nullptr));
    MIVP_REQUIRE(d->C % 8 == 0 && d->Cout % 8 == 0);
    const int NS = (d->Cout + 31) / 32;
    const long T = (long)d->B * d->odims[0] * d->odims[1] * d->odims[2];
    MIVP_REQUIRE(T < (1L << 31));                               // merge_token decodes in 32 bits
    const unsigned gx = (unsigned)((T + 63) / 64);
    const int kC = (d->merge_last ? 8 : 4) * d->C, n_ct = (kC + 15) / 16;
    int ny = (int)((1024 + gx - 1) / gx);
    if (ny > n_ct / 3) ny = n_ct / 3;                           // at least three tiles behind each statistics prologue
    if (ny < 1) ny = 1;
    const dim3 grid(gx, (unsigned)ny);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_MB(K) hipLaunchKernelGGL((k_patch_merge_bwd<K>), grid, dim3(256), 0, st, *d, (const bf16_t*)dy, (const bf16_t*)x, \
                                         (const bf16_t*)yfwd, ln_w, ln_b, wgam, wbet, (const bf16_t*)w_t, (bf16_t*)dx,            \
                                         (bf16_t*)wg_dn, (bf16_t*)wg_x)
    switch (NS) {
        case 1: LAUNCH_MB(1); break;
        case 2: LAUNCH_MB(2); break;
        case 3: LAUNCH_MB(3); break;
        case 4: LAUNCH_MB(4); break;
        case 6: LAUNCH_MB(6); break;
        case 12: LAUNCH_MB(12); break;
        default: mivp_set_error("patch_merge_bwd: Cout/32 not in {1,2,3,4,6,12}"); return MIVP_EUNSUPPORTED;
    }
#undef LAUNCH_MB
    return mivp_check_launch("patch_merge_bwd");
}
