"""Host wrappers of the non-attention kernels (conv, norm, merge, upsample) around the C ABI.

Everything here is plumbing: shape bookkeeping, buffer allocation through torch,
weight re-layout (done once per weight version), one HIP launch per op.
"""
import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L
from .geometry import round_up

BF16 = torch.bfloat16


def _nblk(items: int, groups: int, cap: int = 1024) -> int:
    """Number of 256-thread blocks for a grid-stride kernel whose threads keep a fixed
    channel group: (nblk*256) % groups == 0."""
    want = max(1, min(cap, (items + 255) // 256))
    import math
    step = groups // math.gcd(groups, 256)
    return max(step, (want // step) * step)


# ------------------------------------------------------------------------------------------
# conv 3x3x3
# ------------------------------------------------------------------------------------------
def pack_conv_weight(w: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    """nn.Conv3d weight [Cout, Cin, 3,3,3] -> bf16 [Cout_p, Kp], k = tap*Cin_p + ci, tap = (kh*3+kw)*3+kd."""
    cout, cin = w.shape[0], w.shape[1]
    cin_p = cin_pad or round_up(cin, 8)
    wt = w.detach().float().permute(0, 2, 3, 4, 1).reshape(cout, 27, cin)
    if cin_p != cin:
        wt = torch.nn.functional.pad(wt, (0, cin_p - cin))
    k = 27 * cin_p
    kp = round_up(k, 32)
    out = torch.zeros((round_up(cout, 16), kp), dtype=torch.float32, device=w.device)
    out[:cout, :k] = wt.reshape(cout, k)
    return out.to(BF16).contiguous()


def pack_conv_weight_dgrad(w: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """Weights of the data-gradient convolution: dgrad of a stride-1 pad-1 3^3 conv is the same conv of
    dy with taps flipped and Cin/Cout swapped.  dy is expected with channels padded to a multiple of 8."""
    wf = w.detach().float().flip(2, 3, 4).permute(1, 0, 2, 3, 4).contiguous()     # [Cin, Cout, 3,3,3]
    cpad = round_up(w.shape[0], 8)
    return pack_conv_weight(wf, cin_pad=cpad), cpad


def pack_conv_weight_dgrad16(w: torch.Tensor):
    """The same data-gradient pack for dy padded to 16 channels, or None when pack_conv_weight_dgrad already is one.  The
    halo-brick kernel stages 16-channel chunks: a segmentation head's 2-5 class gradient padded to 8 channels falls to the
    im2col kernel (675 us at 4 x 96^3), padded to 16 it runs as one halo chunk."""
    if round_up(w.shape[0], 8) >= 16:
        return None
    wf = w.detach().float().flip(2, 3, 4).permute(1, 0, 2, 3, 4).contiguous()
    return pack_conv_weight(wf, cin_pad=16)


def conv_desc(B, dims, cin, cout, affine, lrelu, residual, out_f32) -> L.ConvDesc:
    d = L.ConvDesc()
    d.B = B
    for a in range(3):
        d.dims[a] = int(dims[a])
    d.Cin, d.Cout = cin, cout
    d.Kp = round_up(27 * cin, 32)
    d.pro_affine = 1 if affine else 0
    d.pro_lrelu = 1 if lrelu else 0
    d.add_residual = 1 if residual else 0
    d.out_f32 = 1 if out_f32 else 0
    return d


def halo_pack(wp: torch.Tensor, cin: int, cout: int) -> torch.Tensor:
    """Re-lay the im2col pack [Cout_p][tap*Cin + ci] for the halo-brick kernel: bf16 [groups][Cin/16][14][BN][32] where
    a group is 48 output channels (BN = 48, or Cout rounded to 16 for a single group) and k-step j holds taps
    (2j, 2j+1) x 16 channels (tap 27 = zeros).  Cached on the pack tensor (rebuilt with it)."""
    wh = getattr(wp, "_mivp_halo", None)
    if wh is None:
        groups = (cout + 47) // 48
        bn = 48 if groups > 1 else round_up(cout, 16)
        wt = torch.zeros((groups * bn, 28, cin), dtype=BF16, device=wp.device)
        rows = min(wp.shape[0], groups * bn)
        wt[:rows, :27] = wp[:rows, :27 * cin].view(rows, 27, cin)
        wh = wt.view(groups, bn, 14, 2, cin // 16, 16).permute(0, 4, 2, 1, 3, 5).reshape(groups, cin // 16, 14, bn, 32).contiguous()
        wp._mivp_halo = wh
    return wh


# halo-brick kernel geometries: code -> (brick h, brick w, brick d, voxel tiles of the busiest SIMD per k-step)
_HALO_BRICKS = {8: (4, 8, 16, 8), 4: (4, 4, 16, 4), 6: (6, 6, 16, 9), 66: (6, 6, 8, 6), 36: (3, 6, 8, 3)}


def halo_brick(B, dims, cout) -> int:
    """Brick code (8: 4x8x16, 4: 4x4x16, 6: 6x6x16; 66: 6x6x8 and 36: 3x6x8 with voxel tiles of 2 x 8 instead of 1 x 16) for
    the halo-brick kernel, 0 for the im2col kernel (tiny volumes).
    One 128-160 KB workgroup fits a CU, so the cost model is rounds = ceil(workgroups / 256) times the work a SIMD does
    per k-step of one workgroup (+1 for the per-chunk staging / barrier cost); ties go to the smaller brick.  At 48^3 x 4
    the 6x6x16 brick gives 768 workgroups = exactly three rounds where 4x8x16 needs 864 = a fourth, 3/8-full one; the
    8-deep bricks divide the 12 x 12 x 24 and 6 x 6 x 24 volumes of the deep decoder stages exactly (145 -> 110 us and
    70 -> 54 us there, tools/ab_conv_bricks.py); measured on the decoder shapes with tools/bench_conv.py."""
    H, W, D = dims
    if B * H * W * D < 1024:
        return 0
    groups = (cout + 47) // 48
    best, best_cost = 0, None
    for code in (4, 8, 6, 36, 66):
        bh, bw, bd, tiles = _HALO_BRICKS[code]
        wgs = B * ((H + bh - 1) // bh) * ((W + bw - 1) // bw) * ((D + bd - 1) // bd) * groups
        cost = ((wgs + 255) // 256) * (tiles + 1)
        if best_cost is None or cost < best_cost:
            best, best_cost = code, cost
    return best


def conv3d_bn_act(x, wp, bias, cout, scale, shift, lrelu, out_f32=False, fuse_prologue=False):
    """conv3x3x3(act(x * scale + shift)).  Default: one elementwise pass applies the BatchNorm affine + activation and
    the conv streams plain operands.  The halo-brick kernel can also apply the prologue while staging its input
    (``fuse_prologue``; tested): at the decoder shapes that saves the 18 us pass but costs the MFMA-bound conv 12 us,
    a wash end to end, so the product keeps the conv kernel clean.  (Fused into the im2col kernel the prologue would be
    re-applied per tap -- ~100 VALU ops per k-step -- which is why it was split off in the first place.)"""
    if fuse_prologue and not out_f32:
        B, H, W, D, cin = x.shape
        d = conv_desc(B, (H, W, D), cin, cout, True, lrelu, False, out_f32)
        if halo_brick(B, (H, W, D), cout) and L.lib().mivp_conv3d_halo_supported(C.byref(d)):
            return conv3d(x, wp, bias, cout, scale, shift, lrelu, None, False)
    return conv3d(affine_act(x, scale, shift, lrelu), wp, bias, cout, None, None, False, None, out_f32)


def conv3d(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
           scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, lrelu: bool = False,
           residual: Optional[torch.Tensor] = None, out_f32: bool = False, force_halo=False) -> torch.Tensor:
    B, H, W, D, cin = x.shape
    d = conv_desc(B, (H, W, D), cin, cout, scale is not None, lrelu, residual is not None, out_f32)
    if wp.shape != (round_up(cout, 16), d.Kp):
        raise RuntimeError(f"conv3d: packed weight shape {tuple(wp.shape)} does not match Cout={cout}, Cin={cin}")
    bw = int(force_halo) if force_halo not in (False, True) else (8 if force_halo else halo_brick(B, (H, W, D), cout))
    if bw and L.lib().mivp_conv3d_halo_supported(C.byref(d)):
        # large volume, few output channels: the halo-brick kernel (each input voxel fetched once per workgroup)
        y = torch.empty((B, H, W, D, cout), dtype=BF16, device=x.device)
        L.call("mivp_conv3d_halo_fwd", C.byref(d), L.ptr(x), L.ptr(halo_pack(wp, cin, cout)), L.ptr(bias), L.ptr(scale),
               L.ptr(shift), L.ptr(residual), L.ptr(y), C.c_int32(bw), L.stream())
        return y
    y = torch.empty((B, H, W, D, cout), dtype=torch.float32 if out_f32 else BF16, device=x.device)
    ws_bytes = L.lib().mivp_conv3d_fwd_ws(C.byref(d))
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device) if ws_bytes else None
    L.call("mivp_conv3d_fwd", C.byref(d), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(scale), L.ptr(shift),
           L.ptr(residual), L.ptr(y), L.ptr(ws), C.c_size_t(ws_bytes), L.stream())
    return y


def head_conv_supported(cin: int, cout: int) -> bool:
    return 27 * cout <= 60 and cin + 1 <= 64 and cin % 8 == 0


def head_conv(x, conv_w, conv_b, scale, shift):
    """Segmentation head: conv3x3x3(BatchNorm-affine(x)) + bias -> f32 logits, one fused kernel."""
    B, H, W, D, cin = x.shape
    cout = conv_w.shape[0]
    d = conv_desc(B, (H, W, D), cin, cout, True, False, False, True)
    y = torch.empty((B, H, W, D, cout), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(L.lib().mivp_head_conv_ws()) // 2, dtype=BF16, device=x.device)     # the folded weight tile (hi | lo)
    L.call("mivp_head_conv_fwd", C.byref(d), L.ptr(x), L.ptr(conv_w.detach().float().contiguous()),
           L.ptr(conv_b.detach().float().contiguous()), L.ptr(scale), L.ptr(shift), L.ptr(ws), L.ptr(y), L.stream())
    return y


# ------------------------------------------------------------------------------------------
# downstream head on the low-resolution decoder output (csrc/uphead.hip)
# ------------------------------------------------------------------------------------------
def uphead_supported(cin: int, cout: int) -> bool:
    return cin % 8 == 0 and cin + 1 <= 64 and 1 <= cout <= 2


def uphead_batch_stats(x, weight, bias, eps, running_mean=None, running_var=None, momentum=0.1, keep_gx=False):
    """Training-mode BatchNorm statistics of upsample_x2(x) computed from x: (scale, shift, mean_rstd).
    ``keep_gx`` appends gx f32 [B,h,w,d,C] = U^T U x (the pass forms it anyway; uphead_dx reads it back)."""
    B, h, w, d, Cc = x.shape
    gx = torch.empty(x.shape, dtype=torch.float32, device=x.device) if keep_gx else None
    nblk = L.lib().mivp_uphead_nblk(C.c_int32(B), C.c_int32(h), C.c_int32(w), C.c_int32(d), C.c_int32(Cc))
    part = torch.empty((nblk, 2 * Cc), dtype=torch.float32, device=x.device)
    L.call("mivp_uphead_stats", L.ptr(x), C.c_int32(B), C.c_int32(h), C.c_int32(w), C.c_int32(d), C.c_int32(Cc), L.ptr(part),
           L.ptr(gx), L.stream())
    out = bn_finalize(part, nblk, Cc, 8 * B * h * w * d, weight, bias, eps, running_mean, running_var, momentum)
    return (*out, gx) if keep_gx else out


def uphead_fold(conv_w, scale, shift):
    """BatchNorm affine folded into the head conv: bf16 [2][Mp][64] (hi | lo halves of the f32 value), row tap*Cout + co =
    (w*scale | sum_c w*shift | 0)."""
    cout, cin = conv_w.shape[0], conv_w.shape[1]
    wf = torch.empty((2, round_up(27 * cout, 16), 64), dtype=BF16, device=conv_w.device)
    L.call("mivp_uphead_fold", L.ptr(conv_w.detach().float().contiguous()), L.ptr(scale), L.ptr(shift), C.c_int32(cout),
           C.c_int32(cin), L.ptr(wf), L.stream())
    return wf


def uphead_forward(x, wf, conv_b, cout):
    """f32 logits [B,2h,2w,2d,cout] = bias + conv3x3x3(BN-affine(upsample_x2(x))) from the low-res x."""
    B, h, w, d, Cc = x.shape
    ws_bytes = L.lib().mivp_uphead_fwd_ws(C.c_int32(B), C.c_int32(h), C.c_int32(w), C.c_int32(d), C.c_int32(cout))
    ws = torch.empty(ws_bytes // 2, dtype=torch.float16, device=x.device)
    y = torch.empty((B, 2 * h, 2 * w, 2 * d, cout), dtype=torch.float32, device=x.device)
    L.call("mivp_uphead_fwd", L.ptr(x), L.ptr(wf), L.ptr(conv_b.detach().float().contiguous()), C.c_int32(B), C.c_int32(h),
           C.c_int32(w), C.c_int32(d), C.c_int32(Cc), C.c_int32(cout), L.ptr(ws), L.ptr(y), L.stream())
    return y


def uphead_gs(x, dy, cout, keep_d=False, raw=False):
    """(G, S) of the head as conv3d_wgrad_rows defines them, from the low-res x and dy [B,2h,2w,2d,cout] f32:
    G[co, tap, c] = sum_u dy[u - tap][co] * upsample(x)[u][c],  S[co, tap] = sum_{u in bounds} dy[u - tap][co].
    ``keep_d`` also returns the adjoint tensor D [T, 64] (for uphead_dx)."""
    from .swin_ops import _colsum_bf16
    B, h, w, d, Cc = x.shape
    T = B * h * w * d
    ld = 64 if keep_d else round_up(27 * cout, 8)
    dy = dy.contiguous().float()
    D = torch.empty((T, ld), dtype=BF16, device=x.device)
    L.call("mivp_uphead_adjoint", L.ptr(dy), C.c_int32(dy.shape[-1]), C.c_int32(B), C.c_int32(h), C.c_int32(w), C.c_int32(d),
           C.c_int32(cout), L.ptr(D), C.c_int32(ld), L.stream())
    G = gemm_tn(D, operand_rows(ld), x, operand_rows(Cc), T, 27 * cout, Cc)                 # [tap*cout + co][c]
    S = _colsum_bf16(D)
    if raw:                                                   # tap-major, as produced: for head_grads_fused
        return (G, S, D) if keep_d else (G, S)
    G, S = G.view(27, cout, Cc).permute(1, 0, 2).contiguous(), S[:27 * cout].view(27, cout).t().contiguous()
    return (G, S, D) if keep_d else (G, S)


def head_grads_fused(G, S, conv_w, scale, shift, mean_rstd):
    """head_grads_from_gs on the tap-major (G [27*Cout][Cin], S [>= 27*Cout]) of ``uphead_gs(..., raw=True)``: one
    single-workgroup kernel instead of ~12 tiny tensor ops."""
    cout, cin = conv_w.shape[0], conv_w.shape[1]
    dev = conv_w.device
    dW = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device=dev)
    db = torch.empty(cout, dtype=torch.float32, device=dev)
    dgamma = torch.empty(cin, dtype=torch.float32, device=dev)
    dbeta = torch.empty(cin, dtype=torch.float32, device=dev)
    L.call("mivp_head_grads", L.ptr(G), C.c_int64(G.shape[1]), C.c_int64(cout * G.shape[1]), L.ptr(S), C.c_int64(1),
           C.c_int64(cout), L.ptr(conv_w.detach().float().contiguous()), L.ptr(scale), L.ptr(shift), L.ptr(mean_rstd),
           C.c_int32(cout), C.c_int32(cin), L.ptr(dW), L.ptr(db), L.ptr(dgamma), L.ptr(dbeta), L.stream())
    return dW, db, dgamma, dbeta


def uphead_dx(x, D, conv_w, scale, mean_rstd, dgamma, dbeta, training, gx=None):
    """Gradient w.r.t. the low-res x through upsample -> BatchNorm -> conv (see k_uphead_dx); ``gx`` from
    uphead_batch_stats(keep_gx=True) spares the kernel its 27-point gather."""
    B, h, w, d, Cc = x.shape
    cout = conv_w.shape[0]
    wc = torch.empty((round_up(Cc, 16), 64), dtype=BF16, device=x.device)
    coef = torch.empty((4, Cc), dtype=torch.float32, device=x.device)
    L.call("mivp_uphead_dx_prep", L.ptr(conv_w.detach().float().contiguous()), L.ptr(scale), L.ptr(mean_rstd),
           L.ptr(dgamma if training else None), L.ptr(dbeta if training else None), C.c_double(8.0 * B * h * w * d),
           C.c_int32(1 if training else 0), C.c_int32(cout), C.c_int32(Cc), L.ptr(wc), L.ptr(coef), L.stream())
    dx = torch.empty_like(x)
    L.call("mivp_uphead_dx", L.ptr(D), L.ptr(wc), L.ptr(x), L.ptr(coef), L.ptr(gx), C.c_int32(B), C.c_int32(h),
           C.c_int32(w), C.c_int32(d), C.c_int32(Cc), L.ptr(dx), L.stream())
    return dx


def conv3d_wgrad_small(x, scale, shift, lrelu, dy, cout):
    """dw [Cout,Cin,3,3,3] f32 and db [Cout] f32 for a small-Cout conv (segmentation heads)."""
    B, H, W, D, cin = x.shape
    d = conv_desc(B, (H, W, D), cin, cout, scale is not None, lrelu, False, False)
    ws = L.lib().mivp_conv3d_wgrad_small_ws(C.byref(d))
    part = torch.empty(ws, dtype=torch.float32, device=x.device)
    out = torch.empty(cout * 27 * cin + cout, dtype=torch.float32, device=x.device)
    L.call("mivp_conv3d_wgrad_small", C.byref(d), L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(dy),
           C.c_int32(dy.shape[-1]), L.ptr(part), L.ptr(out), L.stream())
    dw = out[:cout * 27 * cin].reshape(cout, 3, 3, 3, cin).permute(0, 4, 1, 2, 3).contiguous()
    return dw, out[cout * 27 * cin:].clone()


def conv3d_wgrad_rows_supported(cin: int, cout: int, depth: int) -> bool:
    """Shape window of mivp_conv3d_wgrad_rows (csrc/conv3d.hip): Cout <= 8, Cin + 1 <= 64, row images within LDS."""
    if cout > 8 or cin + 1 > 64 or cin % 8:
        return False
    dp = (depth + 31) // 32 * 32
    return (dp // 4) * 6 * 128 + ((dp + 8) // 4) * 4 * 128 <= 160 * 1024


def conv3d_wgrad_rows(x, dy, cout):
    """MFMA weight-gradient core for small-Cout convs: returns (G, S) with
    G[co, tap, ci] = sum_u dy[u - tap][co] * x[u][ci]  (raw x) and S[co, tap] = sum_{u in bounds} dy[u - tap][co].
    dy must carry 8 channels (zero padded)."""
    B, H, W, D, cin = x.shape
    if dy.shape[-1] != 8:
        raise RuntimeError("conv3d_wgrad_rows: dy must be padded to 8 channels")
    d = conv_desc(B, (H, W, D), cin, cout, False, False, False, False)
    ws = L.lib().mivp_conv3d_wgrad_rows_ws(C.byref(d))
    if ws == 0:
        raise RuntimeError("conv3d_wgrad_rows: shape not supported (needs Cout <= 8, Cin % 8 == 0, Cin < 64)")
    part = torch.empty(ws, dtype=torch.float32, device=x.device)
    gs = torch.empty((3, 80, 64), dtype=torch.float32, device=x.device)
    L.call("mivp_conv3d_wgrad_rows", C.byref(d), L.ptr(x), L.ptr(dy), C.c_int32(8), L.ptr(part), L.ptr(gs), L.stream())
    # gs[sd][nb*8 + co][c]  ->  [co][tap = nb*3 + sd][c]
    full = gs[:, :72].reshape(3, 9, 8, 64).permute(2, 1, 0, 3).reshape(8, 27, 64)
    return full[:cout, :, :cin], full[:cout, :, cin]


def head_grads_from_gs(G, S, conv_w, scale, shift, mean_rstd):
    """Everything the (BatchNorm -> conv, no activation) head needs from one pass over the data:
    conv dW [Cout,Cin,3,3,3], conv db [Cout], BatchNorm dgamma [Cin], dbeta [Cin].  Tiny tensor algebra."""
    cout, _, cin = G.shape
    dW = (G * scale.view(1, 1, cin) + S.unsqueeze(-1) * shift.view(1, 1, cin))        # [co, tap, ci]
    dW = dW.permute(0, 2, 1).reshape(cout, cin, 3, 3, 3).contiguous()
    db = S[:, 13].contiguous()
    w = conv_w.detach().float().reshape(cout, cin, 27).permute(0, 2, 1)               # [co, tap, ci]
    mean, rstd = mean_rstd[:cin], mean_rstd[cin:]
    dbeta = (w * S.unsqueeze(-1)).sum((0, 1))
    dgamma = rstd * (w * (G - S.unsqueeze(-1) * mean.view(1, 1, cin))).sum((0, 1))
    return dW, db, dgamma, dbeta


# ------------------------------------------------------------------------------------------
# batch norm (training mode)
# ------------------------------------------------------------------------------------------
def bn_finalize(part, nblk, Cc, count, weight, bias, eps, running_mean, running_var, momentum=0.1):
    dev = part.device
    scale = torch.empty(Cc, dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    mean_rstd = torch.empty(2 * Cc, dtype=torch.float32, device=dev)
    L.call("mivp_bn_finalize", L.ptr(part), C.c_int32(nblk), C.c_int32(Cc), C.c_double(float(count)), L.ptr(weight),
           L.ptr(bias), C.c_float(eps), C.c_float(momentum), L.ptr(running_mean), L.ptr(running_var), L.ptr(scale),
           L.ptr(shift), L.ptr(mean_rstd), L.stream())
    return scale, shift, mean_rstd


def bn_batch_stats(x, weight, bias, eps, running_mean=None, running_var=None, momentum=0.1):
    """Batch statistics of a bf16 channels-last tensor -> (scale, shift, mean_rstd); running stats updated in place."""
    Cc = x.shape[-1]
    n_vox = x.numel() // Cc
    nblk = _nblk(n_vox * (Cc // 8), Cc // 8)
    part = torch.empty((nblk, 2 * Cc), dtype=torch.float32, device=x.device)
    L.call("mivp_bn_stats", L.ptr(x), C.c_int64(n_vox), C.c_int32(Cc), C.c_int32(nblk), L.ptr(part), L.stream())
    return bn_finalize(part, nblk, Cc, n_vox, weight, bias, eps, running_mean, running_var, momentum)


def bn_eval_affine(weight, bias, running_mean, running_var, eps):
    """Eval-mode BatchNorm as a per-channel affine (parameter-only arithmetic)."""
    rstd = torch.rsqrt(running_var.float() + eps)
    scale = (weight.float() * rstd).contiguous()
    shift = (bias.float() - running_mean.float() * scale).contiguous()
    return scale, shift, torch.cat([running_mean.float(), rstd]).contiguous()


def affine_act(x, scale, shift, lrelu=False):
    y = torch.empty_like(x)
    Cc = x.shape[-1]
    L.call("mivp_affine_act", L.ptr(x), C.c_int64(x.numel() // Cc), C.c_int32(Cc), L.ptr(scale), L.ptr(shift),
           C.c_int32(1 if lrelu else 0), L.ptr(y), L.stream())
    return y


def bn_backward(x, dy, scale, shift, mean_rstd, lrelu):
    """Training-mode BN (+ optional LeakyReLU) backward: returns dx (bf16), dgamma, dbeta (f32)."""
    Cc = x.shape[-1]
    n_vox = x.numel() // Cc
    nblk = _nblk(n_vox * (Cc // 8), Cc // 8)
    part = torch.empty((nblk, 2 * Cc), dtype=torch.float32, device=x.device)
    L.call("mivp_bn_bwd_stats", L.ptr(x), L.ptr(dy), C.c_int64(n_vox), C.c_int32(Cc), L.ptr(scale), L.ptr(shift),
           L.ptr(mean_rstd), C.c_int32(1 if lrelu else 0), C.c_int32(nblk), L.ptr(part), L.stream())
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=x.device)
    L.call("mivp_reduce_rows", L.ptr(part), C.c_int64(nblk), C.c_int64(2 * Cc), L.ptr(sums), L.stream())
    dx = torch.empty_like(x)
    L.call("mivp_bn_bwd_apply", L.ptr(x), L.ptr(dy), C.c_int64(n_vox), C.c_int32(Cc), L.ptr(scale), L.ptr(shift),
           L.ptr(mean_rstd), L.ptr(sums), C.c_int32(1 if lrelu else 0), L.ptr(dx), L.stream())
    return dx, sums[Cc:], sums[:Cc]


def bn_backward_eval(x, dy, scale, shift, mean_rstd, lrelu):
    """Eval-mode BN (running statistics are constants): dx = scale * act'(z) * dy; dgamma/dbeta as in training."""
    Cc = x.shape[-1]
    n_vox = x.numel() // Cc
    nblk = _nblk(n_vox * (Cc // 8), Cc // 8)
    part = torch.empty((nblk, 2 * Cc), dtype=torch.float32, device=x.device)
    L.call("mivp_bn_bwd_stats", L.ptr(x), L.ptr(dy), C.c_int64(n_vox), C.c_int32(Cc), L.ptr(scale), L.ptr(shift),
           L.ptr(mean_rstd), C.c_int32(1 if lrelu else 0), C.c_int32(nblk), L.ptr(part), L.stream())
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=x.device)
    L.call("mivp_reduce_rows", L.ptr(part), C.c_int64(nblk), C.c_int64(2 * Cc), L.ptr(sums), L.stream())
    zeros = torch.zeros_like(sums)
    dx = torch.empty_like(x)
    L.call("mivp_bn_bwd_apply", L.ptr(x), L.ptr(dy), C.c_int64(n_vox), C.c_int32(Cc), L.ptr(scale), L.ptr(shift),
           L.ptr(mean_rstd), L.ptr(zeros), C.c_int32(1 if lrelu else 0), L.ptr(dx), L.stream())
    return dx, sums[Cc:], sums[:Cc]


# ------------------------------------------------------------------------------------------
# patch embedding (+ its BatchNorm)
# ------------------------------------------------------------------------------------------
def patch_embed(x, w, bias, bn_w, bn_b, eps, running_mean, running_var, training=True, momentum=0.1,
                return_stats=False):
    """x f32 [B, Cin, H, W, D] -> bf16 [B, H/2, W/2, D/2, C] = BN(conv_k2s2(x))."""
    B, cin, H, W, D = x.shape
    Cc = w.shape[0]
    d = L.EmbedDesc()
    d.B, d.Cin, d.C = B, cin, Cc
    for a, v in enumerate((H, W, D)):
        d.dims[a] = v
    n_out = B * (H // 2) * (W // 2) * (D // 2)
    d.nblk = _nblk(n_out * (Cc // 8), Cc // 8, cap=2048)
    x = x.contiguous().float()
    w = w.detach().float().contiguous()
    bias = bias.detach().float().contiguous()
    st = L.stream()
    if training:
        part = torch.empty((d.nblk, 2 * Cc), dtype=torch.float32, device=x.device)
        L.call("mivp_patch_embed", C.byref(d), C.c_int(0), L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(None), L.ptr(None),
               L.ptr(part), L.ptr(None), st)
        scale, shift, mean_rstd = bn_finalize(part, d.nblk, Cc, n_out, bn_w, bn_b, eps, running_mean, running_var, momentum)
    else:
        scale, shift, mean_rstd = bn_eval_affine(bn_w, bn_b, running_mean, running_var, eps)
    y = torch.empty((B, H // 2, W // 2, D // 2, Cc), dtype=BF16, device=x.device)
    L.call("mivp_patch_embed", C.byref(d), C.c_int(1), L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(scale), L.ptr(shift),
           L.ptr(None), L.ptr(y), st)
    if return_stats:
        return y, (scale, shift, mean_rstd)
    return y


def patch_embed_backward(x, w, bias, dy, stats, training=True):
    """Parameter gradients of BN(conv_k2s2(x)): (dW [C, Cin, 2,2,2], dbias, dgamma, dbeta), all f32.
    The conv output is recomputed (bf16) instead of having been kept, the BatchNorm backward is the generic one,
    and dW = dz^T patches runs on the TN GEMM."""
    scale, shift, mean_rstd = stats
    B, cin, H, W, D = x.shape
    Cc = w.shape[0]
    d = L.EmbedDesc()
    d.B, d.Cin, d.C = B, cin, Cc
    for a, v in enumerate((H, W, D)):
        d.dims[a] = v
    n_out = B * (H // 2) * (W // 2) * (D // 2)
    d.nblk = _nblk(n_out * (Cc // 8), Cc // 8, cap=2048)
    x = x.contiguous().float()
    wf = w.detach().float().contiguous()
    bf = bias.detach().float().contiguous()
    st = L.stream()
    z = torch.empty((B, H // 2, W // 2, D // 2, Cc), dtype=BF16, device=x.device)
    one = torch.ones(Cc, dtype=torch.float32, device=x.device)
    L.call("mivp_patch_embed", C.byref(d), C.c_int(1), L.ptr(x), L.ptr(wf), L.ptr(bf), L.ptr(one), L.ptr(torch.zeros_like(one)),
           L.ptr(None), L.ptr(z), st)
    dz, dgamma, dbeta = (bn_backward if training else bn_backward_eval)(z, dy.contiguous(), scale, shift, mean_rstd, False)
    patches = torch.empty((n_out, cin * 8), dtype=BF16, device=x.device)
    L.call("mivp_patch_im2col", C.byref(d), L.ptr(x), L.ptr(patches), st)
    dw = gemm_tn(dz, operand_rows(Cc), patches, operand_rows(cin * 8), n_out, Cc, cin * 8)
    from .swin_ops import _colsum_bf16
    db = _colsum_bf16(dz.view(n_out, Cc))
    return dw.view(Cc, cin, 2, 2, 2), db, dgamma, dbeta


# ------------------------------------------------------------------------------------------
# patch merging
# ------------------------------------------------------------------------------------------
def merge_desc(B, dims, Cc, cout, merge_last) -> L.MergeDesc:
    d = L.MergeDesc()
    d.B, d.C, d.Cout = B, Cc, cout
    d.merge_last = 1 if merge_last else 0
    d.ln_eps = 1e-6
    for a in range(3):
        d.dims[a] = int(dims[a])
        padded = dims[a] + (dims[a] & 1)
        d.odims[a] = padded // 2 if (a < 2 or merge_last) else padded
    return d


def patch_merge(x, ln_w, ln_b, w_bf16, merge_last):
    B, H, W, D, Cc = x.shape
    cout = w_bf16.shape[0]
    d = merge_desc(B, (H, W, D), Cc, cout, merge_last)
    y = torch.empty((B, d.odims[0], d.odims[1], d.odims[2], cout), dtype=BF16, device=x.device)
    L.call("mivp_patch_merge_fwd", C.byref(d), L.ptr(x), L.ptr(ln_w), L.ptr(ln_b), L.ptr(w_bf16), L.ptr(y), L.stream())
    return y


# ------------------------------------------------------------------------------------------
# trilinear upsample + crop + concat
# ------------------------------------------------------------------------------------------
def upcat_desc(B, idims, odims, scale, cx, cs, align_corners=False) -> L.UpcatDesc:
    d = L.UpcatDesc()
    d.B, d.Cx, d.Cs = B, cx, cs
    d.align_corners = 1 if align_corners else 0
    for a in range(3):
        d.idims[a], d.odims[a], d.scale[a] = int(idims[a]), int(odims[a]), int(scale[a])
    return d


def upcat(x, skip, scale: Sequence[int], odims: Optional[Sequence[int]] = None, align_corners: bool = False):
    B, ih, iw, id_, cx = x.shape
    if skip is not None:
        odims = skip.shape[1:4]
        cs = skip.shape[-1]
    else:
        cs = 0
        odims = odims or (ih * scale[0], iw * scale[1], id_ * scale[2])
    d = upcat_desc(B, (ih, iw, id_), odims, scale, cx, cs, align_corners)
    y = torch.empty((B, odims[0], odims[1], odims[2], cx + cs), dtype=BF16, device=x.device)
    L.call("mivp_upcat_fwd", C.byref(d), L.ptr(x), L.ptr(skip), L.ptr(y), L.stream())
    return y


def _upcat_args(x, skip, scale, align_corners=False):
    B, ih, iw, id_, cx = x.shape
    odims = skip.shape[1:4] if skip is not None else (ih * scale[0], iw * scale[1], id_ * scale[2])
    cs = skip.shape[-1] if skip is not None else 0
    return upcat_desc(B, (ih, iw, id_), odims, scale, cx, cs, align_corners), tuple(odims), cx + cs


def upcat_stats(x, skip, scale: Sequence[int]):
    """Partial per-channel sums of the (never materialised) upsample + concat tensor: (part [nblk, 2*Ct], nblk, n_vox)."""
    d, odims, ct = _upcat_args(x, skip, scale)
    rows = x.shape[0] * odims[0] * odims[1]
    lines = x.shape[0] * odims[0]     # nx workgroups walk one (b, oh) line: ~1024 workgroups = one resident round (4 per CU)
    nblk = lines * max(1, min(odims[1], 1024 // lines))
    part = torch.empty((nblk, 2 * ct), dtype=torch.float32, device=x.device)
    L.call("mivp_upcat_stats", C.byref(d), L.ptr(x), L.ptr(skip), C.c_int32(nblk), L.ptr(part), L.stream())
    return part, nblk, rows * odims[2]


def upcat_affine(x, skip, scale: Sequence[int], bn_scale, bn_shift, lrelu: bool):
    """act(BatchNorm-affine(cat(upsample(x), skip))) in one pass (bit-equal to upcat + affine_act)."""
    d, odims, ct = _upcat_args(x, skip, scale)
    y = torch.empty((x.shape[0],) + odims + (ct,), dtype=BF16, device=x.device)
    L.call("mivp_upcat_affine_fwd", C.byref(d), L.ptr(x), L.ptr(skip), L.ptr(bn_scale), L.ptr(bn_shift),
           C.c_int32(1 if lrelu else 0), L.ptr(y), L.stream())
    return y


def upcat_backward(dy, idims, scale, cx, cs, need_skip=True, align_corners=False):
    B = dy.shape[0]
    odims = dy.shape[1:4]
    d = upcat_desc(B, idims, odims, scale, cx, cs, align_corners)
    dx = torch.empty((B, idims[0], idims[1], idims[2], cx), dtype=BF16, device=dy.device)
    dskip = torch.empty((B, odims[0], odims[1], odims[2], cs), dtype=BF16, device=dy.device) if (cs and need_skip) else None
    L.call("mivp_upcat_bwd", C.byref(d), L.ptr(dy), L.ptr(dx), L.ptr(dskip), L.stream())
    return dx, dskip


def cast_bf16(t: torch.Tensor) -> torch.Tensor:
    t = t.detach().float().contiguous()
    out = torch.empty(t.shape, dtype=BF16, device=t.device)
    L.call("mivp_cast_f32_bf16", L.ptr(t), C.c_int64(t.numel()), L.ptr(out), L.stream())
    return out


def add_bf16(a, b):
    y = torch.empty_like(a)
    L.call("mivp_add_bf16", L.ptr(a), L.ptr(b), C.c_int64(a.numel()), L.ptr(y), L.stream())
    return y


def patch_merge_backward(dy, x, ln_w, ln_b, w_t_bf16, merge_last, need_w=False, y_fwd=None, wgam=None, wbet=None):
    """dx, and with ``need_w`` also (d reduction.weight [Cout, kC], d norm.weight, d norm.bias) in f32.
    y_fwd = the forward output; wgam / wbet = W gamma, W beta (recomputed here when not handed in)."""
    B, H, W, D, Cc = x.shape
    cout = dy.shape[-1]
    d = merge_desc(B, (H, W, D), Cc, cout, merge_last)
    dx = torch.empty_like(x)
    T = dy.numel() // cout
    kC = (8 if merge_last else 4) * Cc
    wg_dn = torch.empty((T, kC), dtype=BF16, device=x.device) if need_w else None
    wg_x = torch.empty((T, kC), dtype=BF16, device=x.device) if need_w else None
    if wgam is None:
        wf = w_t_bf16.float()                                   # [kC, Cout]
        wgam, wbet = (ln_w @ wf).contiguous(), (ln_b @ wf).contiguous()
    if y_fwd is None:
        y_fwd = patch_merge(x, ln_w, ln_b, w_t_bf16.t().contiguous(), merge_last)
    L.call("mivp_patch_merge_bwd", C.byref(d), L.ptr(dy), L.ptr(x), L.ptr(y_fwd), L.ptr(ln_w), L.ptr(ln_b), L.ptr(wgam),
           L.ptr(wbet), L.ptr(w_t_bf16), L.ptr(dx), L.ptr(wg_dn), L.ptr(wg_x), L.stream())
    if not need_w:
        return dx
    from .swin_ops import ln_wgrad
    n, dgamma, dbeta = ln_wgrad(wg_x, wg_dn, ln_w, ln_b, float(d.ln_eps), T, kC)
    dw = gemm_tn(dy, operand_rows(cout), n, operand_rows(kC), T, cout, kC)
    return dx, dw, dgamma, dbeta


# ---------------------------------------------------------------------------------------------
# weight gradients (csrc/wgrad.hip): out[M][N] (+)= alpha * sum_t A[t][m] B[t][n]
# ---------------------------------------------------------------------------------------------
def conv3d_wgrad(x: torch.Tensor, dy: torch.Tensor, cout: int, cin: Optional[int] = None):
    """Weight / bias gradient of a 3x3x3 'same' convolution: x bf16 [B,H,W,D,Cin_p] (the conv's input as it was fed,
    channel padding included), dy bf16 [B,H,W,D,>=cout].  Returns (dW [cout, cin, 3,3,3], dbias [cout]) in f32."""
    B, H, W, D, cin_p = x.shape
    cin = cin or cin_p
    ld = dy.shape[-1]
    if ld % 8 or cin_p % 4:
        raise RuntimeError("conv3d_wgrad: dy channels must be padded to 8, x channels to 4")
    vox = B * H * W * D
    from .swin_ops import _colsum_bf16
    db = _colsum_bf16(dy.view(vox, ld))[:cout]
    if ld <= 16 and cin_p > ld:
        # few output channels (segmentation heads): let x supply the rows and the shifted dy the columns,
        #   dW[co][ci][k] = sum_u x[u][ci] * dy[u - (k-1)][co]  =  out[ci][(26 - tap)*ld + co]
        # so the 64x64 output blocks are full instead of 5/64 occupied
        out = gemm_tn(x, operand_rows(cin_p), dy, operand_conv_taps((H, W, D), ld, ld), vox, cin_p, 27 * ld)
        dw = out.view(cin_p, 27, ld)[:, :, :cout].flip(1).permute(2, 0, 1).contiguous()
        return dw.view(cout, cin_p, 3, 3, 3)[:, :cin], db
    dw = gemm_tn(dy, operand_rows(ld), x, operand_conv_taps((H, W, D), cin_p, cin_p), vox, cout, 27 * cin_p,
                 perm_cin=cin_p)
    return dw.view(cout, cin_p, 3, 3, 3)[:, :cin], db


def operand_rows(ld: int) -> L.OperandDesc:
    return L.OperandDesc(0, ld, 0, 0, (L.i32 * 3)(0, 0, 0), 0)


def operand_heads(rows: int, hd: int) -> L.OperandDesc:
    """[T/rows][C/hd][rows][hd]: the layout of the attention kernels' q/k/v (and their gradients)."""
    return L.OperandDesc(1, 0, rows, hd, (L.i32 * 3)(0, 0, 0), 0)


def operand_conv_taps(dims: Sequence[int], cin: int, ld: int) -> L.OperandDesc:
    return L.OperandDesc(2, ld, 0, 0, (L.i32 * 3)(*dims), cin)


def gemm_tn(a: torch.Tensor, a_desc: L.OperandDesc, b: torch.Tensor, b_desc: L.OperandDesc, T: int, M: int, N: int,
            out: Optional[torch.Tensor] = None, alpha: float = 1.0, accumulate: bool = False,
            perm_cin: int = 0) -> torch.Tensor:
    """fp32 [M, N] = alpha * A^T B over T tokens (bf16 operands described by a_desc / b_desc)."""
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.numel() == M * N
    d = L.GemmTnDesc(T, M, N, a_desc, b_desc, alpha, 1 if accumulate else 0, perm_cin)
    ws_bytes = L.lib().mivp_gemm_tn_ws(C.byref(d))
    ws = torch.empty(ws_bytes // 4, device=a.device, dtype=torch.float32)
    L.call("mivp_gemm_tn", C.byref(d), L.ptr(a), L.ptr(b), L.ptr(ws), C.c_size_t(ws_bytes), L.ptr(out), L.stream())
    return out


# ---------------------------------------------------------------------------------------------
# pieces of the phase-1 proxy heads (swin_unetr.py:64-83,180-222)
# ---------------------------------------------------------------------------------------------
def pointwise_conv(x, w, bias):
    """1x1x1 conv, <= 4 output channels: x bf16 [B,H,W,D,C], w [Cout, C(,1,1,1)] -> f32 [B,H,W,D,Cout]."""
    B, H, W, D, Cc = x.shape
    cout = w.shape[0]
    wf = w.detach().float().reshape(cout, Cc).contiguous()
    y = torch.empty((B, H, W, D, cout), dtype=torch.float32, device=x.device)
    L.call("mivp_pointwise_fwd", L.ptr(x), L.ptr(wf), L.ptr(None if bias is None else bias.detach().float().contiguous()),
           C.c_int64(B * H * W * D), C.c_int32(Cc), C.c_int32(cout), L.ptr(y), L.stream())
    return y


def pointwise_conv_backward(x, w, dy, need_dx=True):
    """(dx bf16, dW [Cout, C], db [Cout]) of pointwise_conv; dy f32 [B,H,W,D,Cout]."""
    B, H, W, D, Cc = x.shape
    cout = w.shape[0]
    n = B * H * W * D
    wf = w.detach().float().reshape(cout, Cc).contiguous()
    dy = dy.contiguous().float()
    dx = torch.empty_like(x)
    dyb = torch.empty((n, 4), dtype=BF16, device=x.device)
    L.call("mivp_pointwise_bwd", L.ptr(dy), L.ptr(wf), C.c_int64(n), C.c_int32(Cc), C.c_int32(cout), L.ptr(dx), L.ptr(dyb),
           L.stream())
    dw = gemm_tn(dyb, operand_rows(4), x, operand_rows(Cc), n, cout, Cc)
    db = dy.view(n, cout).sum(0)
    return (dx if need_dx else None), dw, db


def instance_norm_act(x, eps=1e-5, lrelu=True):
    """nn.InstanceNorm3d (no affine, batch statistics per sample) + LeakyReLU(0.01): the per-sample form of the training
    BatchNorm kernels.  Returns (y, saved) with saved = per-sample (scale, shift, mean_rstd)."""
    B, Cc = x.shape[0], x.shape[-1]
    one = torch.ones(Cc, dtype=torch.float32, device=x.device)
    zero = torch.zeros(Cc, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    saved = []
    for b in range(B):
        scale, shift, mean_rstd = bn_batch_stats(x[b:b + 1], one, zero, eps)
        y[b:b + 1] = affine_act(x[b:b + 1], scale, shift, lrelu)
        saved.append((scale, shift, mean_rstd))
    return y, saved


def instance_norm_act_backward(x, dy, saved, lrelu=True):
    dx = torch.empty_like(x)
    for b, (scale, shift, mean_rstd) in enumerate(saved):
        dxb, _, _ = bn_backward(x[b:b + 1], dy[b:b + 1].contiguous(), scale, shift, mean_rstd, lrelu)
        dx[b:b + 1] = dxb
    return dx


def global_avg_pool(x):
    """nn.AdaptiveAvgPool3d((1,1,1)) on channels-last bf16: f32 [B, C]."""
    from .swin_ops import _colsum_bf16
    B, Cc = x.shape[0], x.shape[-1]
    n = x.numel() // (B * Cc)
    return torch.stack([_colsum_bf16(x[b].reshape(n, Cc)) for b in range(B)]) / float(n)


# ---------------------------------------------------------------------------------------------
# transposed convolution, kernel == stride (MONAI UnetrUpBlock's up-sampling step; csrc/convt.hip)
# ---------------------------------------------------------------------------------------------
def pack_convt_weight(w: torch.Tensor):
    """nn.ConvTranspose3d weight [Cin, Cout, s0, s1, s2] -> (w1 bf16 [taps*Cout][Cin] for the forward GEMM,
    w2 bf16 [Cin][taps*Cout] for the data gradient); column / row (tap, co) with tap = (a*s1 + b)*s2 + c."""
    cin, cout = w.shape[0], w.shape[1]
    wf = w.detach().float()
    w1 = wf.permute(2, 3, 4, 1, 0).reshape(-1, cin).to(BF16).contiguous()
    w2 = wf.permute(0, 2, 3, 4, 1).reshape(cin, -1).to(BF16).contiguous()
    return w1, w2


def convt_forward(x: torch.Tensor, w1: torch.Tensor, stride, cout: int) -> torch.Tensor:
    B, h, w, d, cin = x.shape
    y = torch.empty((B, h * stride[0], w * stride[1], d * stride[2], cout), dtype=BF16, device=x.device)
    L.call("mivp_convt_fwd", C.c_int32(B), (C.c_int32 * 3)(h, w, d), (C.c_int32 * 3)(*stride), C.c_int32(cin), C.c_int32(cout),
           L.ptr(x), L.ptr(w1), L.ptr(y), L.stream())
    return y


def convt_dgrad(dy: torch.Tensor, w2: torch.Tensor, stride, cin: int) -> torch.Tensor:
    B, H, W, D, cout = dy.shape
    h, w, d = H // stride[0], W // stride[1], D // stride[2]
    dx = torch.empty((B, h, w, d, cin), dtype=BF16, device=dy.device)
    L.call("mivp_convt_dgrad", C.c_int32(B), (C.c_int32 * 3)(h, w, d), (C.c_int32 * 3)(*stride), C.c_int32(cin), C.c_int32(cout),
           L.ptr(dy), L.ptr(w2), L.ptr(dx), L.stream())
    return dx


def convt_wgrad(x: torch.Tensor, dy: torch.Tensor, stride) -> torch.Tensor:
    """dW [Cin, Cout, s0, s1, s2] f32 = TN GEMM over the low-resolution tokens of x and the space-to-depth view of dy."""
    B, h, w, d, cin = x.shape
    cout = dy.shape[-1]
    s0, s1, s2 = stride
    dyg = dy.view(B, h, s0, w, s1, d, s2, cout).permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, s0 * s1 * s2 * cout)
    T = B * h * w * d
    g = gemm_tn(x.reshape(T, cin), operand_rows(cin), dyg, operand_rows(dyg.shape[1]), T, cin, dyg.shape[1])
    return g.view(cin, s0, s1, s2, cout).permute(0, 4, 1, 2, 3).contiguous()
