// 3x3x3 stride-1 pad-1 convolution, "halo brick" form for LARGE volumes with few output channels
// (the last decoder stage's conv_concat, swin_unetr/unet_blocks.py:46-56,74: 144 -> 48 channels at 48^3).
//
// The im2col kernel (conv3d.hip) fetches every input voxel 27 times through L1 -- once per tap -- and that
// path, not the MFMA pipe, bounds it when Cout is small (few MACs per fetched byte).  Here a workgroup owns a
// brick of 4 x 8 x 16 = 512 output voxels and stages the brick's 6 x 10 x 18 input halo ONCE per 16-channel
// chunk in LDS; the 27 taps are then just LDS row offsets:
//
//   for chunk c of 16 input channels:                       (global -> registers one chunk ahead, double-buffered LDS)
//       halo  [1080 rows][16 ch]        34.5 KB             rows = halo voxels, d fastest
//       wts   [14 k-steps][Cout][32 k]  Cout * 0.9 KB       k-step j = taps (2j, 2j+1) x 16 channels (tap 27 = zeros)
//       for j in 0..13:  every wave: 4 voxel-tile fragments + NTN weight fragments -> 4*NTN MFMAs
//
// One barrier per chunk (14 k-steps) instead of one per k-step; global traffic per MFMA is ~1/8 of the im2col
// kernel's.  MFMA roles as everywhere (common.hpp): weight tile on operand A, voxel tile on operand B, so a lane
// ends with one voxel and four consecutive output channels.
//
// LDS bank behaviour: ds_read_b128 is served in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... i.e.
// lanes (r in 0-3|12-15, g) together with lanes (r in 4-11, g^1).  A voxel-tile fragment read takes halo rows
// base + r (32 B apart) and the 16-byte half g&1, so inside a group the rows 8 apart -- the ones that share a
// 32-byte bank window -- always take OPPOSITE halves: plain rows are conflict-free, no swizzle (an XOR on the row
// index measured 32 % conflict cycles).  Both tap selections (g>>1) never meet in one group.  Weight rows are 64 B
// with the chunk swizzle of conv3d.hip.
#include "common.hpp"

namespace {

// output brick BH x BW x 16 voxels; a wave owns TPW tiles of 16 voxels along d (consecutive along w):
//   4 x 8 x 16, 8 waves x 4 tiles   (512 voxels)
//   4 x 4 x 16, 4 waves x 4 tiles   (256 voxels: the mid-size decoder stages, whose 12- and 24-voxel axes an 8-wide
//                                    brick would mostly pad)
//   6 x 6 x 16, 12 waves x 3 tiles  (576 voxels: 48^3 x 4 volumes are 768 such bricks = exactly three per CU, where the
//                                    864 4x8x16 bricks need a fourth, 3/8-full round)
constexpr int KSTEPS = 14;                                    // 28 taps (27 + one zero tap) x 16 channels / 32
// Deep decoder stages (round 3): 12 x 12 x 24 and 6 x 6 x 24 volumes.  With 16 voxels of a tile along d a 24-deep axis is
// covered by two tiles (25 % padding) and the 4 x 8 / 4 x 4 bricks pad h and w as well: the dec2 conv ran 44 % of its
// MFMAs on padding, on 192 of 256 CUs.  BD = 8 bricks use tiles of 2 (w) x 8 (d) voxels -- the halo rows of a tile are then
// two runs of 8 -- and divide those volumes exactly:
//   6 x 6 x 8, 6 waves x 3 tiles    (288 voxels; dec2: 48 bricks x 4 channel groups)
//   3 x 6 x 8, 3 waves x 3 tiles    (144 voxels; bottleneck: 24 bricks x 8 channel groups)
template <int BH, int BW, int TPW, int BD = 16>
struct HaloGeom {
    // HD: halo rows along d as laid out in LDS.  The 2 x 8 tiles read two runs of 8 rows HD apart; with HD = 16 (10 used) the
    // second run falls in the bank windows the first one leaves free (rows r and r + 8 take opposite 16-byte halves in a
    // ds_read_b128 service group, see the top of the file); packed at 10 the two runs collide two-way
    static constexpr int HB_D = BD, HD_USED = BD + 2, HD = BD == 8 ? 16 : BD + 2;
    static constexpr int TW = BD == 8 ? 2 : 1;                  // w columns of a voxel tile
    static constexpr int HH = BH + 2, HW = BW + 2;
    static constexpr int HROWS = HH * HW * HD;                 // 1080 (4x8) / 648 (4x4) / 1152 (6x6)
    static constexpr int HALO_BYTES = HROWS * 32;
    static constexpr int WAVES = BH * (BW / TW) / TPW, TILES = TPW;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int HPIECES = (HROWS * 2 + THREADS - 1) / THREADS;   // 16-byte halo pieces per thread
    static_assert(BW % (TW * TPW) == 0, "a wave's tiles stay in one brick row");
};
template <int BRICK> struct BrickOf;
template <> struct BrickOf<8> { using G = HaloGeom<4, 8, 4>; };
template <> struct BrickOf<4> { using G = HaloGeom<4, 4, 4>; };
template <> struct BrickOf<6> { using G = HaloGeom<6, 6, 3>; };
template <> struct BrickOf<66> { using G = HaloGeom<6, 6, 3, 8>; };
template <> struct BrickOf<36> { using G = HaloGeom<3, 6, 3, 8>; };

// 16 zero bytes in global memory: the LDS-DMA source of halo pieces outside the volume
__device__ __attribute__((aligned(16))) unsigned int g_halo_zero[4] = {0u, 0u, 0u, 0u};

// one 16-byte global -> LDS DMA per lane (global_load_lds_dwordx4): no VGPR destination, no ds_write; the LDS address is
// wave-uniform base + lane * 16, so the images below are laid out lane-linearly and masked lanes write nothing
MIVP_DEV void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

MIVP_DEV int halo_off(int row, int half) { return row * 32 + (half << 4); }
MIVP_DEV int wswz(int row, int chunk) { return chunk ^ ((0 - (row >> 2)) & 3); }

template <int NTN, int BRICK, bool PRO>
__global__ __launch_bounds__(BrickOf<BRICK>::G::THREADS) void k_conv3d_halo(MivpConvDesc d, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ wh, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const bf16_t* __restrict__ residual, bf16_t* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using GEO = typename BrickOf<BRICK>::G;
    constexpr int HB_H = GEO::HH - 2, HB_W = GEO::HW - 2, TPW = GEO::TILES, HB_D = GEO::HB_D, HD = GEO::HD, TW = GEO::TW;
    constexpr int HW = GEO::HW, HROWS = GEO::HROWS, HALO_BYTES = GEO::HALO_BYTES, HTHREADS = GEO::THREADS, HPIECES = GEO::HPIECES;
    constexpr int BN = 16 * NTN;
    constexpr int WBYTES = KSTEPS * BN * 64;
    constexpr int WPIECES = (WBYTES / 16 + HTHREADS - 1) / HTHREADS;
    auto Hs = [&](int buf) -> char* { return smem + buf * (HALO_BYTES + WBYTES); };
    auto Ws = [&](int buf) -> char* { return smem + buf * (HALO_BYTES + WBYTES) + HALO_BYTES; };
    // optional prologue (BatchNorm affine + LeakyReLU of the layer in front, unet_blocks.py:41-45,74): applied ONCE per
    // staged input piece -- not once per tap as in the im2col kernel -- and only to voxels inside the volume (the conv's
    // zero padding comes after the activation).  scale | shift live behind the tile buffers: [2][Cin] f32.
    float* aff = reinterpret_cast<float*>(smem + 2 * (HALO_BYTES + WBYTES));

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave index
    const int r = lane & 15, g = lane >> 4;
    const int H = d.dims[0], W = d.dims[1], D = d.dims[2], Cin = d.Cin;
    // blockIdx.y = group of BN output channels (convolutions with many output channels re-stage the halo per group:
    // the halo is small next to the weights); the packed weights are [group][chunk][k-step][BN][32]
    const int co_base = blockIdx.y * BN;
    wh += (long)blockIdx.y * (Cin / 16) * (WBYTES / 2);
    const int nbh = (H + HB_H - 1) / HB_H, nbw = (W + HB_W - 1) / HB_W, nbd = (D + HB_D - 1) / HB_D;
    // XCD-aware brick order: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of bricks (shared halos in L2)
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned qb = nb >> 3, rb = nb & 7;
    unsigned brick = (xcd < rb ? xcd * (qb + 1) : rb * (qb + 1) + (xcd - rb) * qb) + idx;
    const int bd = brick % nbd; brick /= nbd;
    const int bw = brick % nbw; brick /= nbw;
    const int bh = brick % nbh;
    const long b = brick / nbh;
    const int h0 = bh * HB_H, w0 = bw * HB_W, d0 = bd * HB_D;

    // ---- per-thread constants of the halo staging: element offset of each 16-byte piece (or -1: outside the volume)
    long hsrc[HPIECES];
    int hdst[HPIECES];
#pragma unroll
    for (int u = 0; u < HPIECES; ++u) {
        const int p = tid + HTHREADS * u;
        hsrc[u] = -1;
        hdst[u] = -1;
        if (p < HROWS * 2) {
            const int row = p >> 1, half = p & 1;
            const int hd = row % HD, hw = (row / HD) % HW, hh = row / (HD * HW);
            const int gh = h0 + hh - 1, gw = w0 + hw - 1, gd = d0 + hd - 1;
            hdst[u] = halo_off(row, half);
            if ((unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W && (unsigned)gd < (unsigned)D && hd < GEO::HD_USED)
                hsrc[u] = ((((long)b * H + gh) * W + gw) * D + gd) * Cin + 8 * half;
        }
    }
    const int nchunks = Cin / 16;
    if (PRO) {
        for (int c = tid; c < Cin; c += HTHREADS) { aff[c] = scale[c]; aff[Cin + c] = shift[c]; }
        __syncthreads();
    }
    bf16x8 hreg[HPIECES], wreg[WPIECES];
    int fetched_chunk = 0;
    auto fetch = [&](int c) {
        // every load is unconditional (pieces outside the volume read a clamped, valid address and are zeroed when they are
        // written to LDS): a select on the loaded value here would make the wave wait for its own loads in every chunk
#pragma unroll
        for (int u = 0; u < HPIECES; ++u) hreg[u] = ld8(x + (hsrc[u] >= 0 ? hsrc[u] : 0) + 16 * c);
        fetched_chunk = c;
        const bf16_t* wsrc = wh + (long)c * (WBYTES / 2);
#pragma unroll
        for (int u = 0; u < WPIECES; ++u) {
            const int p = tid + HTHREADS * u;
            wreg[u] = ld8(wsrc + 8 * (p < WBYTES / 16 ? p : 0));
        }
    };
    auto stage = [&](int buf) {
        char* hs = Hs(buf);
        char* ws = Ws(buf);
#pragma unroll
        for (int u = 0; u < HPIECES; ++u) {
            if (hdst[u] < 0) continue;
            bf16x8 v = hsrc[u] >= 0 ? hreg[u] : zero8();
            if (PRO && hsrc[u] >= 0) {
                const int c0 = 16 * fetched_chunk + 8 * ((tid + HTHREADS * u) & 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float f = (float)v[i] * aff[c0 + i] + aff[Cin + c0 + i];
                    if (d.pro_lrelu) f = f > 0.f ? f : 0.01f * f;
                    v[i] = (bf16_t)f;
                }
            }
            *reinterpret_cast<bf16x8*>(hs + hdst[u]) = v;
        }
#pragma unroll
        for (int u = 0; u < WPIECES; ++u) {
            const int p = tid + HTHREADS * u;
            if (p < WBYTES / 16) {
                const int row = p >> 2, ch = p & 3;                    // row = kstep*BN + co
                *reinterpret_cast<bf16x8*>(ws + row * 64 + 16 * wswz(row, ch)) = wreg[u];
            }
        }
    };

    // LDS-DMA staging (every call without the fused prologue): piece p of the halo image lives at byte 16 p, so the 64
    // pieces of one wave-instruction are contiguous; the weight image's per-row chunk swizzle moves to the SOURCE
    // address (slot s of row holds chunk s ^ f(row)).  No staging registers, no VGPR -> LDS pass, no vmcnt waits inside
    // the chunk: the DMAs of chunk c+1 are issued at the top of chunk c and retired by the wait + barrier at its end.
    constexpr bool use_dma = !PRO;                             // PRO: the fused BatchNorm + LeakyReLU prologue needs the register path
    auto dma = [&](int c, int buf) {
        char* hs = Hs(buf);
        char* ws = Ws(buf);
#pragma unroll
        for (int u = 0; u < HPIECES; ++u) {
            const int p = tid + HTHREADS * u;
            if (p < HROWS * 2) {
                const void* src = hsrc[u] >= 0 ? (const void*)(x + hsrc[u] + 16 * c) : (const void*)g_halo_zero;
                glds16(src, hs + (size_t)(p - lane) * 16);
            }
        }
        const bf16_t* wsrc = wh + (long)c * (WBYTES / 2);
#pragma unroll
        for (int u = 0; u < WPIECES; ++u) {
            const int p = tid + HTHREADS * u;
            if (p < WBYTES / 16) {
                const int row = p >> 2, ch = (p & 3) ^ ((0 - (row >> 2)) & 3);
                glds16(wsrc + 8 * (row * 4 + ch), ws + (size_t)(p - lane) * 16);
            }
        }
    };

    // ---- this wave's TPW voxel tiles: brick rows (th, tw_i), 16 voxels along d; byte offset of voxel r of tile i at tap 0
    constexpr int WPR = HB_W / (TW * TPW);                        // waves per brick row
    const int th = wave / WPR, tw0 = (wave % WPR) * TPW * TW;
    const int half = g & 1, tsel = g >> 1;
    // voxel r of a tile: (w, d) = (0, r) for the 1 x 16 tiles, (r >> 3, r & 7) for the 2 x 8 ones
    const int rw = TW == 2 ? (r >> 3) : 0, rd = TW == 2 ? (r & 7) : r;
    int vbyte[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) vbyte[i] = halo_off(((th * HW) + (tw0 + TW * i + rw)) * HD + rd, half);
    // weight fragment: row (16nt + r) of k-step j, chunk g under the row swizzle (constant per lane: j*BN and 16nt are multiples of 16)
    const int wbyte = r * 64 + 16 * wswz(r, g);

    f32x4 acc[TPW][NTN];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) acc[i][nt] = fzero4();

    if (use_dma) {
        dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) { if (use_dma) dma(c + 1, buf ^ 1); else fetch(c + 1); }
        const char* hs = Hs(buf);
        const char* ws = Ws(buf) + wbyte;
        // fragments of k-step j+1 are read while the MFMAs of k-step j run (two register sets, loop fully unrolled)
        bf16x8 vf[2][TPW], wf[2][NTN];
        auto read_frags = [&](int j, int set) {
            // taps 2j (lanes g < 2) and 2j+1 (g >= 2); tap 27 has zero weights, keep its rows in range
            const int t0 = 2 * j, t1 = 2 * j + 1 < 27 ? 2 * j + 1 : 26;
            const int o0 = (((t0 / 9) * HW + (t0 / 3) % 3) * HD + t0 % 3) * 32;
            const int o1 = (((t1 / 9) * HW + (t1 / 3) % 3) * HD + t1 % 3) * 32;
            const int toff = tsel ? o1 : o0;
#pragma unroll
            for (int i = 0; i < TPW; ++i) vf[set][i] = *reinterpret_cast<const bf16x8*>(hs + vbyte[i] + toff);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) wf[set][nt] = *reinterpret_cast<const bf16x8*>(ws + (j * BN + 16 * nt) * 64);
        };
        read_frags(0, 0);
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            const int cur = j & 1;
            if (j + 1 < KSTEPS) read_frags(j + 1, cur ^ 1);
            // the next chunk's tiles go to the other LDS buffer during this chunk (that buffer was last read in chunk c-1),
            // late in this chunk: the global loads issued at the top have landed by then and the writes overlap the
            // remaining MFMAs.  (Spreading the pieces over k-steps 2..12 instead measured 4-20 % SLOWER: the early
            // writes wait for their loads and an in-order wave stalls its MFMAs behind them.)
            if (j == KSTEPS - 4 && c + 1 < nchunks && !use_dma) stage(buf ^ 1);
            // Without the fences the machine scheduler sinks every ds_read to just above its first use (read, wait, MFMA):
            // with two waves per SIMD nothing hides the LDS latency then and the matrix pipe sat at 37 % busy.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i][nt] = mfma16(wf[cur][nt], vf[cur][i], acc[i][nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the DMAs into the other buffer are ordered for the next chunk's ds_reads only by every issuing wave's vmcnt
        // wait followed by the barrier
        if (use_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: bias, residual, bf16 store (lane: voxel r of tile i, channels 16nt + 4g .. +3).  All bias and residual
    //      loads are issued before the first use: one exposed memory latency per workgroup instead of one per (tile, nt).
    f32x4 bv[NTN];
    bf16x4 rv[TPW][NTN];
    bool ok[TPW];
    long voff[TPW];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int co = co_base + 16 * nt + 4 * g;
        bv[nt] = (bias && co < d.Cout) ? *reinterpret_cast<const f32x4*>(bias + co) : fzero4();
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int gh = h0 + th, gw = w0 + tw0 + TW * i + rw, gd = d0 + rd;
        ok[i] = gh < H && gw < W && gd < D;
        voff[i] = ((((long)b * H + gh) * W + gw) * D + gd) * d.Cout;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const int co = co_base + 16 * nt + 4 * g;
            rv[i][nt] = (residual && ok[i] && co < d.Cout) ? ld4(residual + voff[i] + co) : zero4();
        }
    }
    // Wide stores: a lane's natural piece is 8 bytes (4 channels of one voxel), so one store instruction would write a
    // third of every 96-byte voxel row and leave the L2 to merge partial lines (measured: up to 1.56x the output bytes
    // in 64-byte write requests).  When the workgroup's channel slice is whole, each wave stages its TPW x 16 voxels x
    // BN channels in its own corner of the (now idle) LDS and writes 16 bytes per lane, whole rows at a time.
    const bool wide = (co_base + BN <= d.Cout) && (d.Cout % 8 == 0);
    if (wide) {
        char* ost = smem + (size_t)wave * (TPW * 16 * BN * 2);    // the last chunk's barrier has passed: LDS is free
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                f32x4 v = acc[i][nt] + bv[nt];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)rv[i][nt][j];
                *reinterpret_cast<bf16x4*>(ost + ((i * 16 + r) * BN + 16 * nt + 4 * g) * 2) = pack4(v);
            }
        // the wave reads back only what it wrote itself: no barrier
        constexpr int SEG = BN * 2 / 16;                          // 16-byte pieces per voxel
        const long row0 = (((long)b * H + (h0 + th)) * W + (w0 + tw0)) * D + d0;
#pragma unroll
        for (int p0 = 0; p0 < TPW * 16 * SEG; p0 += 64) {
            const int p = p0 + lane;
            if (p < TPW * 16 * SEG) {
                const int vox = p / SEG, seg = p - vox * SEG, i = vox >> 4, rr = vox & 15;
                const int ow = TW * i + (TW == 2 ? (rr >> 3) : 0), od = TW == 2 ? (rr & 7) : rr;
                const bool inside = (h0 + th) < H && (w0 + tw0 + ow) < W && (d0 + od) < D;
                if (inside) {
                    const bf16x8 piece = *reinterpret_cast<const bf16x8*>(ost + (size_t)vox * (BN * 2) + 16 * seg);
                    st8(y + (row0 + (long)ow * D + od) * d.Cout + co_base + 8 * seg, piece);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        if (!ok[i]) continue;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const int co = co_base + 16 * nt + 4 * g;
            if (co < d.Cout) {
                f32x4 v = acc[i][nt] + bv[nt];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)rv[i][nt][j];
                st4(y + voff[i] + co, pack4(v));
            }
        }
    }
}

}  // namespace

// Shapes the halo kernel takes: Cin a multiple of 16, Cout a multiple of 4; up to 48 output channels per workgroup
// (two halo buffers + two weight buffers of 14 * 48 * 64 B must fit the 160 KB of LDS), more through blockIdx.y groups
// of 48; plain bf16 output with optional bias.  Whether it PAYS (enough bricks x groups to fill 256 CUs, bricks not
// mostly padding) is the caller's call: see mivp_amd/ops.py.
extern "C" int mivp_conv3d_halo_supported(const MivpConvDesc* d) {
    if (!d || d->out_f32) return 0;
    if (d->pro_affine && d->Cin > 1024) return 0;
    if (d->Cin % 16 || d->Cout % 4 || d->Cout < 1) return 0;
    if (d->Cout > 48 && d->Cout % 48) return 0;
    return 1;
}

/* wh: bf16 [groups][Cin/16][14][BN][32] with BN = 16*ceil(min(Cout, 48)/16), groups = ceil(Cout/48); element
 * (grp, c, j, co, kk):  kk < 16 : weight[48 grp + co][16c + kk][tap 2j]   kk >= 16 : ...[16c + kk - 16][tap 2j + 1]
 * (tap 27 and rows past Cout: zero) */
extern "C" int mivp_conv3d_halo_fwd(const MivpConvDesc* d, const void* x, const void* wh, const float* bias,
                                    const float* scale, const float* shift, const void* residual, void* y, int32_t brick_w,
                                    mivp_stream_t stream) {
    MIVP_REQUIRE(d && x && wh && y);
    MIVP_REQUIRE(!d->pro_affine || (scale && shift));
    MIVP_REQUIRE((d->add_residual != 0) == (residual != nullptr));
    MIVP_REQUIRE(brick_w == 4 || brick_w == 8 || brick_w == 6 || brick_w == 66 || brick_w == 36);
    if (!mivp_conv3d_halo_supported(d)) { mivp_set_error("conv3d_halo_fwd: shape outside the halo kernel's window"); return MIVP_EUNSUPPORTED; }
    const int groups = (d->Cout + 47) / 48;
    const int ntn = groups > 1 ? 3 : (d->Cout + 15) / 16;
    // brick codes: 8 = 4 x 8 x 16, 4 = 4 x 4 x 16, 6 = 6 x 6 x 16; 66 = 6 x 6 x 8 and 36 = 3 x 6 x 8 (tiles of 2 x 8 voxels)
    const int bh = brick_w == 6 || brick_w == 66 ? 6 : (brick_w == 36 ? 3 : 4);
    const int bwid = brick_w == 66 || brick_w == 36 ? 6 : brick_w;
    const int bd = brick_w == 66 || brick_w == 36 ? 8 : 16;
    const long bricks = (long)d->B * ((d->dims[0] + bh - 1) / bh) * ((d->dims[1] + bwid - 1) / bwid) *
                        ((d->dims[2] + bd - 1) / bd);
    const size_t halo_bytes = brick_w == 8 ? BrickOf<8>::G::HALO_BYTES : brick_w == 4 ? BrickOf<4>::G::HALO_BYTES
                            : brick_w == 6 ? BrickOf<6>::G::HALO_BYTES : brick_w == 66 ? BrickOf<66>::G::HALO_BYTES
                                                                                       : BrickOf<36>::G::HALO_BYTES;
    const size_t lds = 2 * (halo_bytes + (size_t)KSTEPS * 16 * ntn * 64) + (d->pro_affine ? (size_t)2 * d->Cin * sizeof(float) : 0);
    if (lds > 160 * 1024) { mivp_set_error("conv3d_halo_fwd: LDS budget exceeded"); return MIVP_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
#define HALO_LAUNCH(N, W)                                                                                                \
    do {                                                                                                                 \
        auto kern = d->pro_affine ? k_conv3d_halo<N, W, true> : k_conv3d_halo<N, W, false>;                             \
        MIVP_LDS_OPT_IN(kern, lds);                                                                                      \
        hipLaunchKernelGGL(kern, dim3((unsigned)bricks, (unsigned)groups), dim3(BrickOf<W>::G::THREADS), lds, st, *d,    \
                           (const bf16_t*)x, (const bf16_t*)wh, bias, scale, shift, (const bf16_t*)residual, (bf16_t*)y); \
    } while (0)
#define HALO_BRICK(W)                                                                                                    \
    do {                                                                                                                 \
        if (ntn == 1) HALO_LAUNCH(1, W);                                                                                 \
        else if (ntn == 2) HALO_LAUNCH(2, W);                                                                            \
        else HALO_LAUNCH(3, W);                                                                                          \
    } while (0)
    if (brick_w == 8) HALO_BRICK(8);
    else if (brick_w == 4) HALO_BRICK(4);
    else if (brick_w == 6) HALO_BRICK(6);
    else if (brick_w == 66) HALO_BRICK(66);
    else HALO_BRICK(36);
#undef HALO_BRICK
#undef HALO_LAUNCH
    return mivp_check_launch("conv3d_halo_fwd");
}
