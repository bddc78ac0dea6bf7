"""Host wrappers of the Swin-block kernels (K1-K3) around the C ABI.

Mirrors ``SwinTransformerBlock.forward_attn_mlp`` (swin_transformer/swin_block.py:145-255):
``swin_block_forward`` = gather+LN+QKV -> prompt K/V -> bias augmentation ->
window attention -> proj+MLP+scatter, each one HIP launch.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from .geometry import BlockTables, block_tables, round_up

BF16 = torch.bfloat16
USE_FUSED_ATTN_BWD = True      # one-pass attention backward where it applies (tests flip it to compare with the two-pass form)
import os as _os
USE_FP8_ATTN_FWD = _os.environ.get("MIVP_FP8_ATTN_FWD") == "1"   # EXPERIMENT (BASELINE configs[4]): E4M3 Q'/K'/V/P in the forward attention of head_dim < 16 blocks;
#                                measured slower-or-equal and 20x less accurate than bf16 (profiles/r02_fp8_attention.json): off
#                                unless MIVP_FP8_ATTN_FWD=1 / bench.py --fp8-attn (configs[4] with its named arithmetic)


@dataclass
class SwinBlockWeights:
    """Device-side, kernel-ready copies of one block's parameters."""
    heads: int
    ln1_w: torch.Tensor
    ln1_b: torch.Tensor
    wqkv: torch.Tensor          # bf16 [3C, C] row-major (prompt K/V kernels)
    wproj: Optional[torch.Tensor]   # (unused: the kernels read the fragment images below)
    bproj: torch.Tensor
    ln2_w: torch.Tensor
    ln2_b: torch.Tensor
    wmlp: Optional[torch.Tensor]
    bmlp: torch.Tensor
    t_h: torch.Tensor           # f32 [heads, 2w-1], already * embed_dim**-0.5 / 3
    t_w: torch.Tensor
    t_d: torch.Tensor
    ts: Optional[torch.Tensor]  # f32 [heads, Np], already * embed_dim**-0.5
    wqkv_f: Optional[torch.Tensor] = None    # fragment images (pack_weight_frags) read by the token kernels: forward
    wproj_f: Optional[torch.Tensor] = None
    wmlp_f: Optional[torch.Tensor] = None    # paired k order
    wqkv_t: Optional[torch.Tensor] = None    # ... backward: images of the transposes ([C, 3C], [C, C], [C, C] paired)
    wproj_t: Optional[torch.Tensor] = None
    wmlp_t: Optional[torch.Tensor] = None
    aug_cache: Optional[dict] = None         # (Nqp, Nkp, augp) -> (qa, ka) for prompt-free calls (constants of the weights)


def pack_weight_frags(wm: torch.Tensor, paired: bool = False, k_steps: int = 0) -> torch.Tensor:
    """Row-major [R, K] bf16 -> the MFMA-fragment-major image the token kernels read (mivp.h "weight fragment images"):
    [R/16 row tiles][K/32 k-steps][64 lanes][8], rows padded to 16 and K to 32 with zeros.  Lane (r, g) = 16 g + r of
    (row tile nt, k-step s) holds W[16 nt + r][32 s + 8 g .. + 8], or with ``paired`` the two 4-element groups
    W[..][32 s + 4 g .. + 4] | W[..][32 s + 16 + 4 g .. + 4] (the k order of a GEMM whose B operand is the previous GEMM's
    accumulator).  A wave's A fragment is then 1 KB contiguous: one fully coalesced load instead of 64 scattered 16-byte
    ones (the texture-address unit spends about a cycle per cache line touched -- DESIGN.md, r02 TA counters)."""
    R, K = wm.shape
    ks = max((K + 31) // 32, k_steps)                    # k_steps: the kernel's count when it exceeds ceil(K / 32)
    src = wm.contiguous()
    out = torch.empty(((R + 15) // 16, ks, 64, 8), dtype=BF16, device=wm.device)
    L.call("mivp_pack_weight_frags", L.ptr(src), C.c_int32(R), C.c_int32(K), C.c_int32(ks), C.c_int32(1 if paired else 0),
           L.ptr(out), L.stream())
    return out


WIDE_C = (48, 96, 192, 384)                              # csrc/swin_tok_wide.hip (proj / MLP pair)


def paired_and_natural(wm: torch.Tensor) -> torch.Tensor:
    """Image of the second GEMM's weight of the proj/MLP kernels: the paired form; for the wide stages followed by the
    natural form (the column-split kernels feed that GEMM from an LDS row image, in natural k order; which kernel runs is
    decided per call -- dropout and weight-gradient outputs stay with the 16-token kernel)."""
    img = pack_weight_frags(wm, paired=True)
    if wm.shape[0] in WIDE_C:
        img = torch.cat([img.reshape(-1), pack_weight_frags(wm).reshape(-1)])
    return img


def weights_from_state(sd, prefix, heads, embed_dim, n_prompt, device, need_bwd=False) -> SwinBlockWeights:
    f = lambda k: sd[prefix + k].detach().to(device=device, dtype=torch.float32).contiguous()
    scale = embed_dim ** -0.5
    tabs = []
    for name in "hwd":
        tabs.append(((f(f"pe.weights_content_{name}") @ f(f"pe.enc_content_{name}").t()) * (scale / 3.0)).contiguous())
    ts = None
    if n_prompt:
        ts = ((f("pe.weights_token") @ f("pe.enc_token.0").t())[:, :n_prompt] * scale).contiguous()
    mats = [f("attn.to_q.weight"), f("attn.to_k.weight"), f("attn.to_v.weight"), f("attn.proj.weight"), f("mlp.weight")]
    Cc = int(mats[0].shape[0])
    ct, ks = (Cc + 15) // 16, (Cc + 31) // 32
    img = ct * ks * 512                                  # elements of one [C][C] fragment image
    two = 2 if Cc in WIDE_C else 1                       # paired (+ natural) images, see paired_and_natural
    new = lambda n: torch.empty(n, dtype=BF16, device=device)
    w = SwinBlockWeights(
        heads=heads, ln1_w=f("attn_norm.weight"), ln1_b=f("attn_norm.bias"), wqkv=new(3 * Cc * Cc).view(3 * Cc, Cc),
        wproj=None, bproj=f("attn.proj.bias"), ln2_w=f("mlp_norm.weight"), ln2_b=f("mlp_norm.bias"), wmlp=None,
        bmlp=f("mlp.bias"), t_h=tabs[0], t_w=tabs[1], t_d=tabs[2], ts=ts)
    w.wqkv_f, w.wproj_f, w.wmlp_f = new(((3 * Cc + 15) // 16) * ks * 512), new(img), new(two * img)
    if need_bwd:                                         # k_swin_qkv_bwd<CT> walks ceil(3 * 16 CT / 32) k-steps (C = 8: two)
        w.wqkv_t, w.wmlp_t, w.wproj_t = new(ct * ((3 * 16 * ct + 31) // 32) * 512), new(img), new(two * img)
    # one launch for all of them (a step that changes every parameter rebuilt these with ~25 launches per block)
    L.call("mivp_pack_block_weights", C.c_int32(Cc), *[L.ptr(m) for m in mats], C.c_int32(1 if two == 2 else 0),
           L.ptr(w.wqkv), L.ptr(w.wqkv_f), L.ptr(w.wproj_f), L.ptr(w.wmlp_f), L.ptr(w.wqkv_t), L.ptr(w.wmlp_t),
           L.ptr(w.wproj_t), L.stream())
    return w


def make_desc(B, C, heads, tb: BlockTables, n_prompt: int) -> L.SwinDesc:
    d = L.SwinDesc()
    d.B, d.C, d.heads = B, C, heads
    d.vol_in = d.vol_out = tb.dims[0] * tb.dims[1] * tb.dims[2]
    d.P, d.Nq, d.Nqp = tb.P, tb.Nq, tb.Nqp
    d.Np = n_prompt
    d.Npp = round_up(n_prompt, 16) if n_prompt else 0
    d.Nkp = round_up(d.Nqp + d.Npp, 32)
    w = tb.window
    d.aug = w[0] + w[1] + w[2] - 1
    d.augp = round_up(d.aug, 4)
    d.has_mask = 1 if tb.has_mask else 0
    for a in range(3):
        d.win[a] = w[a]
    d.q_scale = float((C // heads) ** -0.5)
    d.ln_eps = 1e-6
    return d


def prompt_desc(Cc: int, heads: int, window, n_prompt: int) -> L.SwinDesc:
    """The descriptor fields the prompt-side kernels read (no volume geometry: B = P = 1)."""
    d = L.SwinDesc()
    d.B, d.C, d.heads = 1, int(Cc), int(heads)
    w = [int(v) for v in window]
    d.P = 1
    d.Nq = w[0] * w[1] * w[2]
    d.Nqp = round_up(d.Nq, 16)
    d.vol_in = d.vol_out = d.Nq
    d.Np = int(n_prompt)
    d.Npp = round_up(n_prompt, 16)
    d.Nkp = round_up(d.Nqp + d.Npp, 32)
    d.aug = w[0] + w[1] + w[2] - 1
    d.augp = round_up(d.aug, 4)
    d.has_mask = 0
    for a in range(3):
        d.win[a] = w[a]
    d.q_scale = float((Cc // heads) ** -0.5)
    d.ln_eps = 1e-6
    return d


def prompt_aug_image(w: SwinBlockWeights, d: L.SwinDesc):
    """(qa, ka) of a prompted block for ts = 0, cached on the weights object: with frozen content tables only the prompt
    rows' i0 columns of ka change from step to step, and mivp_prompt_kv_fwd_multi rewrites those."""
    key = ("prompt", int(d.Nqp), int(d.Nkp), int(d.augp), int(d.Np))
    if w.aug_cache is None:
        w.aug_cache = {}
    if key not in w.aug_cache:
        dev = w.t_h.device
        qa = torch.empty((d.Nqp, d.augp), dtype=BF16, device=dev)
        ka = torch.empty((w.heads, d.Nkp, d.augp), dtype=BF16, device=dev)
        zero = torch.zeros((w.heads, d.Np), dtype=torch.float32, device=dev)
        L.call("mivp_relbias_aug", C.byref(d), L.ptr(w.t_h), L.ptr(w.t_w), L.ptr(w.t_d), L.ptr(zero), L.ptr(qa), L.ptr(ka), L.stream())
        w.aug_cache[key] = (qa, ka)
    return w.aug_cache[key]


@dataclass
class SwinSaved:
    desc: object
    tb: BlockTables
    x: torch.Tensor
    q: torch.Tensor
    k: torch.Tensor
    v: torch.Tensor
    kp: Optional[torch.Tensor]
    vp: Optional[torch.Tensor]
    qa: torch.Tensor
    ka: torch.Tensor
    o: torch.Tensor
    lse: torch.Tensor
    t1: torch.Tensor


def set_dropout(d: L.SwinDesc, dropout):
    """dropout = (p_attn, p_proj, seed_attn, seed_proj[, epoch_ptr]) or None: fills the descriptor's counter-hash dropout
    fields; ``epoch_ptr`` = address of a device int32 word that the kernels fold into both seeds (recorded graphs: the
    graph increments it, so every replay draws new masks -- mivp.h ``seed_epoch``)."""
    if not dropout:
        return
    p_attn, p_proj, seed_attn, seed_proj = dropout[:4]
    d.seed_epoch = int(dropout[4]) if len(dropout) > 4 and dropout[4] else None
    for name, p, seed in (("attn", p_attn, seed_attn), ("proj", p_proj, seed_proj)):
        thr = int(round(float(p) * 65536.0))
        if not 0 <= thr < 65536:
            raise ValueError("dropout probability must be in [0, 1)")
        setattr(d, f"{name}_drop_thr", thr)
        setattr(d, f"{name}_drop_scale", 65536.0 / (65536.0 - thr))
        setattr(d, f"{name}_seed", int(seed) & 0xFFFFFFFF)


def swin_block_forward(x: torch.Tensor, prompt: Optional[torch.Tensor], w: SwinBlockWeights,
                       ts: Optional[torch.Tensor], window, shift_cfg, save: bool = False, dropout=None, pre=None):
    """x: bf16 [B, H, W, D, C] channels-last; prompt: f32 [Np, C] or None; ts: f32 [heads, Np] prompt-token
    bias scores (``None`` -> ``w.ts``); dropout: None or (p_attn, p_proj, seed_attn, seed_proj) for a training
    forward with ``attn_drop`` / ``proj_drop`` (window_attention.py:57,60).  Returns y (same shape) and, if
    ``save``, what backward needs (the descriptor carries the dropout seeds to the backward kernels)."""
    if ts is None:
        ts = w.ts
    if ts is not None:
        ts = ts.detach().float().contiguous()
    if x.dtype != BF16 or x.dim() != 5:
        raise RuntimeError("swin_block_forward expects a bf16 [B,H,W,D,C] tensor")
    B, H, W_, D, Cc = x.shape
    tb = block_tables((H, W_, D), tuple(int(v) for v in window), tuple(int(v) for v in shift_cfg), str(x.device))
    n_prompt = 0 if prompt is None else int(prompt.shape[0])
    d = make_desc(B, Cc, w.heads, tb, n_prompt)
    set_dropout(d, dropout)
    hd = Cc // w.heads
    dev = x.device
    BP = B * tb.P
    st = L.stream()
    q = torch.empty((BP, w.heads, d.Nqp, hd), dtype=BF16, device=dev)
    k = torch.empty_like(q)
    v = torch.empty_like(q)
    L.call("mivp_swin_qkv_fwd", C.byref(d), L.ptr(x), L.ptr(tb.tok_src), L.ptr(w.ln1_w), L.ptr(w.ln1_b), L.ptr(w.wqkv_f),
           L.ptr(q), L.ptr(k), L.ptr(v), st)
    kp = vp = None
    if pre is not None and pre[0] != (int(d.Nqp), int(d.Nkp), int(d.augp), n_prompt):
        pre = None                               # prepared for another window geometry: recompute here
    if pre is not None:                          # functional.prepare_prompted_blocks: prompt K / V and the bias columns are ready
        _, kp, vp, qa, ka = pre
    else:
        if n_prompt:
            kp = torch.empty((w.heads, d.Npp, hd), dtype=BF16, device=dev)
            vp = torch.empty_like(kp)
            pr = prompt.detach().to(torch.float32).contiguous()
            L.call("mivp_prompt_kv_fwd", C.byref(d), L.ptr(pr), L.ptr(w.ln1_w), L.ptr(w.ln1_b), L.ptr(w.wqkv),
                   L.ptr(kp), L.ptr(vp), L.ptr(None), st)
        aug_key = (d.Nqp, d.Nkp, d.augp)
        cached = None if (n_prompt or w.aug_cache is None) else w.aug_cache.get(aug_key)
        if cached is not None:
            qa, ka = cached
        else:
            qa = torch.empty((d.Nqp, d.augp), dtype=BF16, device=dev)
            ka = torch.empty((w.heads, d.Nkp, d.augp), dtype=BF16, device=dev)
            L.call("mivp_relbias_aug", C.byref(d), L.ptr(w.t_h), L.ptr(w.t_w), L.ptr(w.t_d), L.ptr(ts), L.ptr(qa), L.ptr(ka), st)
            if not n_prompt:                     # without prompt columns the tables depend on the (frozen) weights only
                if w.aug_cache is None:
                    w.aug_cache = {}
                w.aug_cache[aug_key] = (qa, ka)
    o = torch.empty((BP, d.Nqp, Cc), dtype=BF16, device=dev)
    fp8 = USE_FP8_ATTN_FWD and hd < 16 and hd + d.augp <= 32 and not dropout
    # the log-sum-exp rows feed the backward passes only (the fp8 experiment's kernel always writes them)
    lse = torch.empty((BP, w.heads, d.Nqp), dtype=torch.float32, device=dev) if (save or fp8) else None
    if fp8:
        L.call("mivp_win_attn_fwd_fp8", C.byref(d), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(kp), L.ptr(vp),
               L.ptr(qa), L.ptr(ka), L.ptr(tb.tok_rid), L.ptr(o), L.ptr(lse), st)
    else:
        L.call("mivp_win_attn_fwd", C.byref(d), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(kp), L.ptr(vp),
               L.ptr(qa), L.ptr(ka), L.ptr(tb.tok_rid), L.ptr(o), L.ptr(lse), L.ptr(tb.mask_words), L.ptr(tb.cut_flags), st)
    t1 = torch.empty((BP, d.Nqp, Cc), dtype=BF16, device=dev) if save else None
    y = torch.empty_like(x)
    L.call("mivp_swin_proj_mlp_fwd", C.byref(d), L.ptr(o), L.ptr(x), L.ptr(tb.tok_src), L.ptr(tb.tok_dst), L.ptr(w.wproj_f),
           L.ptr(w.bproj), L.ptr(w.ln2_w), L.ptr(w.ln2_b), L.ptr(w.wmlp_f), L.ptr(w.bmlp), L.ptr(t1), L.ptr(y), st)
    if save:
        return y, SwinSaved(d, tb, x, q, k, v, kp, vp, qa, ka, o, lse, t1)
    return y, None


def _colsum_bf16(x2d: torch.Tensor) -> torch.Tensor:
    """Column sums of a bf16 [rows, C] matrix (f32 [C]) with the BatchNorm statistics kernel (fixed-order sums)."""
    from . import ops
    rows, Cc = x2d.shape
    nblk = ops._nblk(rows * (Cc // 8), Cc // 8)
    part = torch.empty((nblk, 2 * Cc), dtype=torch.float32, device=x2d.device)
    L.call("mivp_bn_stats", L.ptr(x2d), C.c_int64(rows), C.c_int32(Cc), C.c_int32(nblk), L.ptr(part), L.stream())
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=x2d.device)
    L.call("mivp_reduce_rows", L.ptr(part), C.c_int64(nblk), C.c_int64(2 * Cc), L.ptr(sums), L.stream())
    return sums[:Cc]


def ln_wgrad(x: torch.Tensor, dn: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, T: int, Cc: int,
             tok_src: Optional[torch.Tensor] = None, Nqp: int = 0, P: int = 0, vol: int = 0):
    """LayerNorm parameter gradients.  Returns (n [T, C] bf16 = LN(x) rows, dgamma, dbeta)."""
    from . import ops
    dev = dn.device
    stats = torch.empty((T, 2), dtype=torch.float32, device=dev)
    n = torch.empty((T, Cc), dtype=BF16, device=dev)
    nblk = ops._nblk(T * (Cc // 8), Cc // 8)
    part = torch.empty((nblk, 2 * Cc), dtype=torch.float32, device=dev)
    L.call("mivp_ln_wgrad", L.ptr(x), L.ptr(tok_src), L.ptr(dn), C.c_int64(T), C.c_int32(Cc), C.c_int32(Nqp), C.c_int32(P),
           C.c_int64(vol), C.c_float(eps), L.ptr(gamma), L.ptr(beta), L.ptr(stats), L.ptr(n), C.c_int32(nblk), L.ptr(part),
           L.stream())
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=dev)
    L.call("mivp_reduce_rows", L.ptr(part), C.c_int64(nblk), C.c_int64(2 * Cc), L.ptr(sums), L.stream())
    return n, sums[Cc:], sums[:Cc]


def swin_block_backward(sv: SwinSaved, w: SwinBlockWeights, prompt: Optional[torch.Tensor], dy: torch.Tensor,
                        need_dx: bool, need_prompt: bool, need_w: bool = False):
    """Backward of ``swin_block_forward``: returns (dx | None, dprompt | None, dts | None[, wg]).

    ``need_w``: also the gradients of the block's own parameters, returned as a dict ``wg`` (f32):
    ln1_w, ln1_b, wq, wk, wv, wproj, bproj, ln2_w, ln2_b, wmlp, bmlp and the three relative-position tables
    t_h, t_w, t_d ``[heads, 2w-1]`` (autograd carries those into ``pe.weights_content_*`` / ``pe.enc_content_*``).

    ``dts`` is the gradient of the ``[heads, Np]`` prompt-token bias scores; autograd carries it into
    ``pe.weights_token`` / ``pe.enc_token``.  When ``need_dx`` is False (first prompted block behind a frozen
    stem) only the prompt key columns of the attention backward are computed."""
    d, tb = sv.desc, sv.tb
    if w.wqkv_t is None:
        raise RuntimeError("swin_block_backward: weights were prepared without transposed copies")
    dev = dy.device
    st = L.stream()
    Cc, heads = d.C, d.heads
    hd = Cc // heads
    BP = d.B * d.P
    from . import ops
    T = BP * d.Nqp
    if need_w:
        need_dx = True                                   # the weight gradients need the full attention backward
        need_prompt = d.Np > 0
    d_o = torch.empty((BP, d.Nqp, Cc), dtype=BF16, device=dev)
    d_t1 = torch.empty_like(d_o)
    dn2 = torch.empty_like(d_o) if need_w else None
    dyw = torch.empty_like(d_o) if need_w else None
    d_pj = torch.empty_like(d_o) if (need_w and d.proj_drop_thr) else None
    L.call("mivp_swin_proj_mlp_bwd", C.byref(d), L.ptr(dy), L.ptr(tb.tok_dst), L.ptr(sv.t1), L.ptr(w.ln2_w), L.ptr(w.ln2_b),
           L.ptr(w.wmlp_t), L.ptr(w.wproj_t), L.ptr(d_o), L.ptr(d_t1), L.ptr(dn2), L.ptr(dyw), L.ptr(d_pj), st)
    wg = None
    if need_w:
        wg = {}
        rows = ops.operand_rows(Cc)
        n2, wg["ln2_w"], wg["ln2_b"] = ln_wgrad(sv.t1, dn2, w.ln2_w, w.ln2_b, d.ln_eps, T, Cc)
        wg["wmlp"] = ops.gemm_tn(dyw, rows, n2, rows, T, Cc, Cc)
        wg["bmlp"] = _colsum_bf16(dyw.view(T, Cc))
        g_pj = d_pj if d_pj is not None else d_t1          # gradient w.r.t. the proj output
        wg["wproj"] = ops.gemm_tn(g_pj, rows, sv.o, rows, T, Cc, Cc)
        wg["bproj"] = _colsum_bf16(g_pj.view(T, Cc))
        del n2, dn2, dyw, d_pj
    dx = dprompt = dts = None
    dk = dv = None
    has_prompt = d.Np > 0
    dkp_part = dvp_part = dtok_part = None
    if has_prompt and (need_prompt or need_dx):          # the kernels always emit the prompt partials when prompts exist
        dkp_part = torch.empty((BP * heads, d.Npp, hd), dtype=torch.float32, device=dev)
        dvp_part = torch.empty_like(dkp_part)
        dtok_part = torch.empty((BP * heads, d.Npp), dtype=torch.float32, device=dev)
    fused = need_dx and not need_w and USE_FUSED_ATTN_BWD and bool(L.lib().mivp_win_attn_bwd_fused_supported(C.byref(d)))
    if fused:
        # one pass: S, dP and the exponentials once per (query, key) pair (csrc/swin_bwd_fused.hip)
        dq = torch.empty_like(sv.q)
        dk = torch.empty_like(sv.k)
        dv = torch.empty_like(sv.v)
        L.call("mivp_win_attn_bwd_fused", C.byref(d), L.ptr(sv.q), L.ptr(sv.k), L.ptr(sv.v), L.ptr(sv.kp), L.ptr(sv.vp),
               L.ptr(sv.qa), L.ptr(sv.ka), L.ptr(tb.tok_rid), L.ptr(sv.o), L.ptr(d_o), L.ptr(sv.lse), L.ptr(dq), L.ptr(dk),
               L.ptr(dv), L.ptr(dkp_part), L.ptr(dvp_part), L.ptr(dtok_part), st)
    elif (not need_dx and has_prompt and need_prompt and USE_FUSED_ATTN_BWD
          and bool(L.lib().mivp_win_attn_bwd_prompt_supported(C.byref(d)))):
        # first prompted block behind a frozen stem: only the prompt keys' partials (one launch, forms delta itself)
        L.call("mivp_win_attn_bwd_prompt", C.byref(d), L.ptr(sv.q), L.ptr(sv.kp), L.ptr(sv.vp), L.ptr(sv.qa), L.ptr(sv.ka),
               L.ptr(sv.o), L.ptr(d_o), L.ptr(sv.lse), L.ptr(dkp_part), L.ptr(dvp_part), L.ptr(dtok_part), st)
    else:
        delta = torch.empty((BP, heads, d.Nqp), dtype=torch.float32, device=dev)
        if need_dx:
            dq = torch.empty_like(sv.q)                      # the dq pass also writes delta for the dkv pass
            L.call("mivp_win_attn_bwd_dq", C.byref(d), L.ptr(sv.q), L.ptr(sv.k), L.ptr(sv.v), L.ptr(sv.kp), L.ptr(sv.vp),
                   L.ptr(sv.qa), L.ptr(sv.ka), L.ptr(tb.tok_rid), L.ptr(sv.o), L.ptr(d_o), L.ptr(sv.lse), L.ptr(delta),
                   L.ptr(dq), st)
            dk = torch.empty_like(sv.k)
            dv = torch.empty_like(sv.v)
        else:
            L.call("mivp_win_attn_delta", C.byref(d), L.ptr(sv.o), L.ptr(d_o), L.ptr(delta), st)
        if need_dx or (has_prompt and need_prompt):
            dka_part = torch.empty((BP * heads, d.Nkp, 32), dtype=torch.float32, device=dev) if need_w else None
            L.call("mivp_win_attn_bwd_dkv", C.byref(d), L.ptr(sv.q), L.ptr(sv.k), L.ptr(sv.v), L.ptr(sv.kp), L.ptr(sv.vp),
                   L.ptr(sv.qa), L.ptr(sv.ka), L.ptr(tb.tok_rid), L.ptr(d_o), L.ptr(sv.lse), L.ptr(delta), L.ptr(dk), L.ptr(dv),
                   L.ptr(dkp_part), L.ptr(dvp_part), L.ptr(dtok_part), L.ptr(dka_part), st)
            if need_w:
                dka = torch.empty((heads, d.Nkp, 32), dtype=torch.float32, device=dev)
                L.call("mivp_reduce_rows", L.ptr(dka_part), C.c_int64(BP), C.c_int64(heads * d.Nkp * 32), L.ptr(dka), st)
                del dka_part
                win = [int(d.win[a]) for a in range(3)]
                tabs = [torch.empty((heads, 2 * win[a] - 1), dtype=torch.float32, device=dev) for a in range(3)]
                L.call("mivp_relbias_grad", C.byref(d), L.ptr(dka), L.ptr(tabs[0]), L.ptr(tabs[1]), L.ptr(tabs[2]), st)
                wg["t_h"], wg["t_w"], wg["t_d"] = tabs
    if need_dx:
        dx = torch.empty_like(sv.x)
        dn1 = torch.empty((BP, d.Nqp, Cc), dtype=BF16, device=dev) if need_w else None
        L.call("mivp_swin_qkv_bwd", C.byref(d), L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(sv.x), L.ptr(tb.tok_src),
               L.ptr(w.ln1_w), L.ptr(w.ln1_b), L.ptr(w.wqkv_t), L.ptr(d_t1), L.ptr(dx), L.ptr(dn1), st)
        if need_w:
            n1, wg["ln1_w"], wg["ln1_b"] = ln_wgrad(sv.x, dn1, w.ln1_w, w.ln1_b, d.ln_eps, T, Cc, tok_src=tb.tok_src,
                                                    Nqp=d.Nqp, P=d.P, vol=d.vol_in)
            rows = ops.operand_rows(Cc)
            hs = ops.operand_heads(d.Nqp, hd)
            wg["wq"] = ops.gemm_tn(dq, hs, n1, rows, T, Cc, Cc, alpha=float(d.q_scale))
            wg["wk"] = ops.gemm_tn(dk, hs, n1, rows, T, Cc, Cc)
            wg["wv"] = ops.gemm_tn(dv, hs, n1, rows, T, Cc, Cc)
            del n1, dn1
    if has_prompt and need_prompt:
        rows = heads * d.Npp * hd
        dkp = torch.empty((heads, d.Npp, hd), dtype=torch.float32, device=dev)
        dvp = torch.empty_like(dkp)
        dtok = torch.empty((heads, d.Npp), dtype=torch.float32, device=dev)
        ins = (C.c_void_p * 3)(dkp_part.data_ptr(), dvp_part.data_ptr(), dtok_part.data_ptr())
        outs = (C.c_void_p * 3)(dkp.data_ptr(), dvp.data_ptr(), dtok.data_ptr())
        nrows = (C.c_int64 * 3)(rows, rows, heads * d.Npp)
        L.call("mivp_reduce_rows_multi", C.c_int32(3), ins, nrows, outs, C.c_int64(BP), st)     # one launch for the three
        pr = prompt.detach().to(torch.float32).contiguous()
        dprompt = torch.empty_like(pr)
        wg_a = wg_n = wg_ln = None
        if need_w:
            wg_a = torch.empty((2, d.Np, Cc), dtype=BF16, device=dev)
            wg_n = torch.empty((d.Np, Cc), dtype=BF16, device=dev)
            wg_ln = torch.empty((2, d.Np, Cc), dtype=torch.float32, device=dev)
        L.call("mivp_prompt_kv_bwd", C.byref(d), L.ptr(dkp), L.ptr(dvp), L.ptr(pr), L.ptr(w.ln1_w), L.ptr(w.ln1_b),
               L.ptr(w.wqkv), L.ptr(dprompt), L.ptr(wg_a), L.ptr(wg_n), L.ptr(wg_ln), st)
        dts = dtok[:, :d.Np].contiguous()
        if need_w:                                       # the prompt rows also pass through attn_norm, to_k and to_v
            rows = ops.operand_rows(Cc)
            ops.gemm_tn(wg_a[0], rows, wg_n, rows, d.Np, Cc, Cc, out=wg["wk"], accumulate=True)
            ops.gemm_tn(wg_a[1], rows, wg_n, rows, d.Np, Cc, Cc, out=wg["wv"], accumulate=True)
            ln_rows = torch.empty((2, Cc), dtype=torch.float32, device=dev)
            L.call("mivp_reduce_rows", L.ptr(wg_ln[0]), C.c_int64(d.Np), C.c_int64(Cc), L.ptr(ln_rows[0]), st)
            L.call("mivp_reduce_rows", L.ptr(wg_ln[1]), C.c_int64(d.Np), C.c_int64(Cc), L.ptr(ln_rows[1]), st)
            wg["ln1_b"] = wg["ln1_b"] + ln_rows[0]
            wg["ln1_w"] = wg["ln1_w"] + ln_rows[1]
    if need_w:
        return dx, dprompt, dts, wg
    return dx, dprompt, dts
