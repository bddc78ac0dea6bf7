"""Autograd glue between the nn.Module surface (swin_unetr.py) and the HIP kernels.

Each fused stage is one ``torch.autograd.Function`` whose forward/backward launch kernels through
``ops`` / ``swin_ops``.  Activations between stages are channels-last bf16 tensors
``[B, H, W, D, C]``; parameters stay fp32 masters in the modules and are re-laid-out for the
kernels by a small per-module cache that is refreshed when a parameter's version counter or storage
changes (``load_state_dict``, ``.to()``, foreach optimizers) or when ANY optimizer has stepped since the
copy was made (fused optimizers do not bump version counters: see ``WeightCache``).

Every stage has two backward flavours, chosen at forward time from ``requires_grad`` of its parameters:
  * frozen parameters (``--training-mode downstream``: swin_unetr.py:33-40, segmentation.py:25-39): data
    gradients, prompt-token / prompt-bias gradients and the segmentation-head gradients only;
  * trainable parameters (the ``*_all`` / ``*_decoder`` modes): additionally every weight gradient, from the
    TN-GEMM / LayerNorm / BatchNorm kernels of csrc/wgrad.hip and csrc/norm_embed.hip.
"""
from __future__ import annotations

from typing import Optional

import torch

import ctypes as C

from . import _lib as L
from . import ops, swin_ops

BF16 = torch.bfloat16


def require_device(x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError(
            "mivp_amd.SwinUnetR runs only on the GPU (hand-written HIP kernels, no CPU or eager fallback). "
            "Move the model and the input to 'cuda'. The CPU oracle in oracle/ is test infrastructure.")


def to_channels_first(y: torch.Tensor) -> torch.Tensor:
    return y.permute(0, 4, 1, 2, 3)


def to_channels_last(x: torch.Tensor, pad_to: int = 8) -> torch.Tensor:
    """float [B,C,H,W,D] -> bf16 [B,H,W,D,Cp] with channels zero-padded to a multiple of 8."""
    y = x.permute(0, 2, 3, 4, 1)
    c = y.shape[-1]
    cp = (c + pad_to - 1) // pad_to * pad_to
    if cp != c:
        y = torch.nn.functional.pad(y, (0, cp - c))
    return y.to(BF16).contiguous()


# BatchNorm's ``num_batches_tracked += 1`` is a 5 us kernel per layer: the increments of one forward are collected and
# issued as ONE multi-tensor add when the model's forward returns (SwinUnetR.forward calls flush_counters()).
_pending_counters = []


def bump_counter(t: torch.Tensor):
    _pending_counters.append(t)


def flush_counters():
    if _pending_counters:
        torch._foreach_add_(_pending_counters, 1)
        _pending_counters.clear()


# ``torch.optim.AdamW(fused=True)`` (and any other optimizer that updates parameters through an op without an
# in-place-version bump) changes a parameter's values while ``p._version`` and ``p.data_ptr()`` stay the same, so the
# version counter alone cannot tell a cache that its packed copies are stale.  Two process-wide counters close that gap:
#   * ``_param_epoch``: advanced by every optimizer step (a post-step hook registered for ALL torch optimizers, plus our own
#     fused optimizer); part of the stamp of parameters that can be stepped (``requires_grad``).  Frozen parameters do not
#     carry it, so the frozen backbone of ``--training-mode downstream`` is packed once;
#   * ``_raw_epoch``: advanced by ``invalidate_weight_caches()`` only -- raw kernels that write parameters behind autograd's
#     back (the EMA teacher update, a replayed graph) -- and part of EVERY stamp: the teacher is frozen
#     (``requires_grad=False`` after ``copy_state_dict``, students_teacher.py:136) yet rewritten every step, and round 2's
#     cache kept serving its first packed weights (an effective tau of 1).
_param_epoch = [0]
_raw_epoch = [0]


def invalidate_weight_caches():
    """Call after changing parameters behind autograd's back (``p.data`` arithmetic through raw kernels, graph replays):
    every packed copy, of trainable and of frozen parameters, is rebuilt at its next use."""
    _param_epoch[0] += 1
    _raw_epoch[0] += 1


def _optimizer_post_step(optimizer, args, kwargs):
    _param_epoch[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_post_step  # noqa: E402

_register_post_step(_optimizer_post_step)


_dropout_epochs = {}


def dropout_epoch(device) -> torch.Tensor:
    """The device's dropout epoch word (int32 [1]): recorded steps increment it, the dropout kernels of a recording fold it
    into their seeds (mivp.h ``MivpSwinDesc.seed_epoch``)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    if key not in _dropout_epochs:
        _dropout_epochs[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _dropout_epochs[key]


class WeightCache:
    """key -> kernel-ready tensors, rebuilt when any source parameter changed."""

    def __init__(self):
        self._store = {}

    @staticmethod
    def _stamp(params):
        ep, raw = _param_epoch[0], _raw_epoch[0]
        return tuple((p._version, p.data_ptr(), str(p.device), ep if p.requires_grad else -1, raw)
                     for p in params if p is not None)

    def get(self, key, params, builder):
        stamp = self._stamp(params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            val = builder()
        self._store[key] = (stamp, val)
        return val


def bn_momentum(bn) -> float:
    """``momentum=None`` means a cumulative moving average in PyTorch (factor 1 / num_batches_tracked); the kernels take
    one factor per call, so hand them that factor for THIS call (the counter is bumped after the forward)."""
    if bn.momentum is not None:
        return float(bn.momentum)
    return 1.0 / float(int(bn.num_batches_tracked) + 1)


# ----------------------------------------------------------------------------------------------
# patch embedding + BatchNorm
# ----------------------------------------------------------------------------------------------
class _PatchEmbedFn(torch.autograd.Function):
    """BN(conv_k2s2(x)); backward = parameter gradients only (the input volume is data)."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, bn_w, bn_b, bn):
        training = bn.training
        y, stats = ops.patch_embed(x, conv_w, conv_b, bn_w.detach().float().contiguous(),
                                   bn_b.detach().float().contiguous(), bn.eps, bn.running_mean, bn.running_var,
                                   training=training, momentum=bn_momentum(bn),
                                   return_stats=True)
        if training:
            bump_counter(bn.num_batches_tracked)
        ctx.save_for_backward(x, conv_w, conv_b, *stats)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, conv_w, conv_b, scale, shift, mean_rstd = ctx.saved_tensors
        dw, db, dgamma, dbeta = ops.patch_embed_backward(x, conv_w, conv_b, dy, (scale, shift, mean_rstd), ctx.training)
        g = ctx.needs_input_grad
        return (None, dw if g[1] else None, db if g[2] else None, dgamma if g[3] else None, dbeta if g[4] else None, None)


def patch_embed(owner, conv, bn, x):
    if x.requires_grad:
        raise NotImplementedError("mivp_amd: gradient w.r.t. the input volume is not built")
    params = [conv.weight, conv.bias, bn.weight, bn.bias]
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _PatchEmbedFn.apply(x.detach(), conv.weight, conv.bias, bn.weight, bn.bias, bn)
    training = bn.training
    with torch.no_grad():
        y = ops.patch_embed(x.detach(), conv.weight, conv.bias, bn.weight.detach().float().contiguous(),
                            bn.bias.detach().float().contiguous(), bn.eps, bn.running_mean, bn.running_var,
                            training=training, momentum=bn_momentum(bn))
        if training:
            bump_counter(bn.num_batches_tracked)
    return y


# ----------------------------------------------------------------------------------------------
# Swin block
# ----------------------------------------------------------------------------------------------
class _SwinBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, prompt, ts, w, window, shift, dropout, pre=None):
        need = (ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        y, saved = swin_ops.swin_block_forward(x, prompt, w, ts, window, shift, save=need, dropout=dropout, pre=pre)
        ctx.saved = saved
        ctx.w = w
        ctx.has_prompt = prompt is not None
        if prompt is not None:
            ctx.save_for_backward(prompt)
        return y

    @staticmethod
    def backward(ctx, dy):
        need_dx = ctx.needs_input_grad[0]
        need_p = ctx.has_prompt and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        prompt = ctx.saved_tensors[0] if ctx.has_prompt else None
        dx, dprompt, dts = swin_ops.swin_block_backward(ctx.saved, ctx.w, prompt, dy.contiguous(), need_dx, need_p)
        ctx.saved = None
        return dx, dprompt, dts, None, None, None, None, None


_SWIN_WG_ORDER = ("ln1_w", "ln1_b", "wq", "wk", "wv", "wproj", "bproj", "ln2_w", "ln2_b", "wmlp", "bmlp")


class _SwinBlockTrainFn(torch.autograd.Function):
    """The same block with trainable weights: the 11 body parameters and the three relative-position tables are
    inputs, so autograd routes the kernels' weight gradients to them."""

    @staticmethod
    def forward(ctx, x, prompt, ts, t_h, t_w, t_d, w, window, shift, dropout, *body):
        y, saved = swin_ops.swin_block_forward(x, prompt, w, ts, window, shift, save=True, dropout=dropout)
        ctx.saved = saved
        ctx.w = w
        ctx.has_prompt = prompt is not None
        if prompt is not None:
            ctx.save_for_backward(prompt)
        return y

    @staticmethod
    def backward(ctx, dy):
        prompt = ctx.saved_tensors[0] if ctx.has_prompt else None
        dx, dprompt, dts, wg = swin_ops.swin_block_backward(ctx.saved, ctx.w, prompt, dy.contiguous(), True,
                                                            ctx.has_prompt, need_w=True)
        ctx.saved = None
        g = ctx.needs_input_grad
        body = tuple(wg[k] if g[10 + i] else None for i, k in enumerate(_SWIN_WG_ORDER))
        return (dx if g[0] else None, dprompt if g[1] else None, dts if g[2] else None,
                wg["t_h"] if g[3] else None, wg["t_w"] if g[4] else None, wg["t_d"] if g[5] else None,
                None, None, None, None) + body


class _TokenScoresFn(torch.autograd.Function):
    """ts [heads, Np] = scale * weights_token @ enc_token[:Np].T and both parameter gradients, one tiny kernel each way."""

    @staticmethod
    def forward(ctx, W, E, n_prompt, scale):
        Wc, Ec = W.detach().float().contiguous(), E.detach().float().contiguous()
        heads, e = Wc.shape
        ts = torch.empty((heads, n_prompt), dtype=torch.float32, device=W.device)
        L.call("mivp_token_scores_fwd", L.ptr(Wc), L.ptr(Ec), C.c_int32(heads), C.c_int32(n_prompt), C.c_int32(e),
               C.c_float(scale), L.ptr(ts), L.stream())
        ctx.save_for_backward(Wc, Ec)
        ctx.meta = (n_prompt, scale)
        return ts

    @staticmethod
    def backward(ctx, dts):
        Wc, Ec = ctx.saved_tensors
        n_prompt, scale = ctx.meta
        heads, e = Wc.shape
        dW = torch.empty_like(Wc)
        dE = torch.zeros_like(Ec) if Ec.shape[0] != n_prompt else torch.empty_like(Ec)
        L.call("mivp_token_scores_bwd", L.ptr(dts.contiguous().float()), L.ptr(Wc), L.ptr(Ec), C.c_int32(heads),
               C.c_int32(n_prompt), C.c_int32(e), C.c_float(scale), L.ptr(dW), L.ptr(dE), L.stream())
        return dW, dE, None, None


class _TokenScoresMultiFn(torch.autograd.Function):
    """The token scores of several blocks in one launch each way (mivp_token_scores_{fwd,bwd}_multi).  ``meta`` = tuple of
    (n_prompt, scale) per block; tensors = W_0, E_0, W_1, E_1, ..."""

    @staticmethod
    def forward(ctx, meta, *WE):
        n = len(meta)
        Ws = [WE[2 * i].detach().float().contiguous() for i in range(n)]
        Es = [WE[2 * i + 1].detach().float().contiguous() for i in range(n)]
        e = int(Ws[0].shape[1])
        ts = [torch.empty((Ws[i].shape[0], meta[i][0]), dtype=torch.float32, device=Ws[i].device) for i in range(n)]
        vp = lambda ts_: (C.c_void_p * n)(*[t.data_ptr() for t in ts_])
        L.call("mivp_token_scores_fwd_multi", C.c_int32(n), vp(Ws), vp(Es), (C.c_int32 * n)(*[int(w_.shape[0]) for w_ in Ws]),
               (C.c_int32 * n)(*[m[0] for m in meta]), C.c_int32(e), (C.c_float * n)(*[m[1] for m in meta]), vp(ts), L.stream())
        ctx.save_for_backward(*Ws, *Es)
        ctx.meta = meta
        return tuple(ts)

    @staticmethod
    def backward(ctx, *dts):
        meta = ctx.meta
        n = len(meta)
        Ws, Es = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        e = int(Ws[0].shape[1])
        dts = [None if g is None else g.contiguous().float() for g in dts]
        dW = [torch.empty_like(w_) for w_ in Ws]
        dE = [torch.zeros_like(E_) if E_.shape[0] != meta[i][0] else torch.empty_like(E_) for i, E_ in enumerate(Es)]
        vp = lambda ts_: (C.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in ts_])
        L.call("mivp_token_scores_bwd_multi", C.c_int32(n), vp(dts), vp(Ws), vp(Es),
               (C.c_int32 * n)(*[int(w_.shape[0]) for w_ in Ws]), (C.c_int32 * n)(*[m[0] for m in meta]), C.c_int32(e),
               (C.c_float * n)(*[m[1] for m in meta]), vp(dW), vp(dE), L.stream())
        out = [None]
        for i in range(n):
            out += [dW[i], dE[i]]
        return tuple(out)


def _block_param_lists(block):
    """(body, content) parameter lists of a Swin block, cached on the block: every ``block.attn_norm.weight`` goes through
    nn.Module.__getattr__ (881 such lookups per cfg1 step were 0.2 ms of a 1.9 ms host enqueue time per step).  The cache is
    dropped when a parameter OBJECT was replaced (first / last entries are checked by identity)."""
    cached = block.__dict__.get("_plists")
    if cached is not None and cached[0][0] is block.attn_norm._parameters["weight"] \
            and cached[1][-1] is block.pe._parameters["weights_content_d"]:
        return cached
    pe, attn = block.pe, block.attn
    body = [block.attn_norm.weight, block.attn_norm.bias, attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
            attn.proj.weight, attn.proj.bias, block.mlp_norm.weight, block.mlp_norm.bias, block.mlp.weight,
            block.mlp.bias]
    content = [pe.enc_content_h, pe.enc_content_w, pe.enc_content_d, pe.weights_content_h, pe.weights_content_w,
               pe.weights_content_d]
    block.__dict__["_plists"] = (body, content)
    return body, content


def _block_weights(block, device):
    body, content = _block_param_lists(block)

    def build():
        sd = {k: v for k, v in block.state_dict().items()}
        return swin_ops.weights_from_state(sd, "", block.num_heads, block.embed_dim, 0, device, need_bwd=True)

    return block._wcache.get("w", body + content, build)


def prepare_prompted_blocks(pairs):
    """Prompt-side operands of several Swin blocks at once: ``pairs`` = [(block, prompt parameter [Np, C]), ...] in the order the
    blocks will run.  A prompt-tuning step used to launch token scores, mivp_relbias_aug and mivp_prompt_kv_fwd per prompted
    block -- three latency-bound 5-8 us kernels -- and the token-score gradient kernel per block on the way back (12 prompted
    blocks: ~0.4 ms of a 5.5 ms cfg3 step).  Here the token scores of all blocks are one launch each way, prompt K / V of all
    blocks one launch (which also drops ts into the prompt rows of each block's cached K'-augmentation image), and
    mivp_relbias_aug leaves the per-step path.  The results are parked on the blocks and picked up by ``swin_block``; blocks
    with trainable body / bias-table parameters, several prompt slots or CPU tensors are left to the per-block path."""
    elig = []
    for blk, prm in pairs:
        pe = blk.pe
        if prm is None or prm.dim() != 2 or not prm.is_cuda or not getattr(pe, "use_token_params", False):
            continue
        if len(pe.enc_token) != 1 or int(prm.shape[0]) > int(pe.enc_token[0].shape[0]):
            continue
        body, content = _block_param_lists(blk)
        if torch.is_grad_enabled() and any(q.requires_grad for q in body + content):
            continue
        elig.append((blk, prm))
    for c0 in range(0, len(elig), 16):
        chunk = elig[c0:c0 + 16]
        if len(chunk) < 2:
            break
        n = len(chunk)
        dev = chunk[0][1].device
        ws = [_block_weights(blk, dev) for blk, _ in chunk]
        nps = [int(prm.shape[0]) for _, prm in chunk]
        descs = [swin_ops.prompt_desc(blk.attn_norm.weight.shape[0], blk.num_heads, blk.window_size, nps[i])
                 for i, (blk, _) in enumerate(chunk)]
        meta = tuple((nps[i], float(blk.pe.scale)) for i, (blk, _) in enumerate(chunk))
        we = []
        for blk, _ in chunk:
            we += [blk.pe.weights_token, blk.pe.enc_token[0]]
        ts = _TokenScoresMultiFn.apply(meta, *we)
        aug = [swin_ops.prompt_aug_image(ws[i], descs[i]) for i in range(n)]           # cached per weights object
        prs = [prm.detach().to(torch.float32).contiguous() for _, prm in chunk]
        kps = [torch.empty((ws[i].heads, descs[i].Npp, descs[i].C // ws[i].heads), dtype=torch.bfloat16, device=dev) for i in range(n)]
        vps = [torch.empty_like(t) for t in kps]
        darr = (L.SwinDesc * n)(*descs)
        vp = lambda ts_: (C.c_void_p * n)(*[t.data_ptr() for t in ts_])
        L.call("mivp_prompt_kv_fwd_multi", C.c_int32(n), darr, vp(prs), vp([w_.ln1_w for w_ in ws]), vp([w_.ln1_b for w_ in ws]),
               vp([w_.wqkv for w_ in ws]), vp([t.detach() for t in ts]), vp(kps), vp(vps), vp([a[1] for a in aug]), L.stream())
        for i, (blk, prm) in enumerate(chunk):
            key = (int(descs[i].Nqp), int(descs[i].Nkp), int(descs[i].augp), nps[i])
            blk.__dict__["_pre"] = (prm, ts[i], (key, kps[i], vps[i], aug[i][0], aug[i][1]))


def token_scores(pe, n_prompt):
    """Bias of the prompt-token key columns (RelativePE.token_scores) through the fused kernels when the block holds
    one prompt (max_prompts == 1, the reference's configuration); the torch form otherwise."""
    if len(pe.enc_token) == 1 and pe.enc_token[0].is_cuda and n_prompt <= pe.enc_token[0].shape[0]:
        return _TokenScoresFn.apply(pe.weights_token, pe.enc_token[0], n_prompt, float(pe.scale))
    return pe.token_scores(n_prompt)


def content_tables(pe):
    """RelativePE.content_tables() -- three [heads, 2w-1] tables -- through the same fused kernels (one launch per table
    and direction; the torch form is a matmul + a scaling forward and two matmuls + a scaling backward per table, ~220
    tiny launches per all-weights step)."""
    heads, e = pe.weights_content_h.shape
    rows = max(getattr(pe, f"enc_content_{a}").shape[0] for a in "hwd")
    if not pe.weights_content_h.is_cuda or ((heads + rows) * e + heads * rows) * 4 > 64 * 1024:
        return pe.content_tables()
    s3 = float(pe.scale) / 3.0
    return tuple(_TokenScoresFn.apply(getattr(pe, f"weights_content_{a}"), getattr(pe, f"enc_content_{a}"),
                                      int(getattr(pe, f"enc_content_{a}").shape[0]), s3) for a in "hwd")


def swin_block(block, x, prompt: Optional[torch.Tensor]):
    pe, attn = block.pe, block.attn
    body, content = _block_param_lists(block)
    n_prompt = 0 if prompt is None else int(prompt.shape[0])
    train_w = torch.is_grad_enabled() and any(p.requires_grad for p in body + content)
    w = _block_weights(block, x.device)
    parked = block.__dict__.pop("_pre", None)          # prepare_prompted_blocks: (prompt, ts, operands) for this forward
    ts = pre = None
    if n_prompt:
        if not pe.use_token_params:
            raise RuntimeError("prompt tokens passed to a block built without token bias parameters")
        if parked is not None and parked[0] is prompt and not train_w:
            ts, pre = parked[1], parked[2]
        else:
            ts = token_scores(pe, n_prompt)       # autograd carries d(ts) into weights_token / enc_token
    dropout = None
    p_attn, p_proj = float(attn.attn_drop.p), float(attn.proj_drop.p)
    if block.training and (p_attn > 0 or p_proj > 0):
        # counter-hash dropout inside the kernels; the two seeds come from torch's CPU generator, so
        # torch.manual_seed() reproduces a run (the random STREAM differs from nn.Dropout's by construction)
        # Under a graph recording the two host-drawn seeds are frozen with the descriptor (a kernel argument); the kernels
        # then fold a device-resident epoch word into them, which the recorded step increments at its start
        # (train.GraphedStep): every replay draws fresh masks, forward and backward of one step agree.
        seeds = torch.randint(0, 2 ** 31 - 1, (2,))
        epoch = dropout_epoch(x.device).data_ptr() if torch.cuda.is_current_stream_capturing() else None
        dropout = (p_attn, p_proj, int(seeds[0]), int(seeds[1]), epoch)
    if train_w:
        t_h, t_w, t_d = content_tables(pe)        # same for the content tables [heads, 2w-1]
        return _SwinBlockTrainFn.apply(x, prompt, ts, t_h, t_w, t_d, w, block.window_size, block.shift_size, dropout,
                                       *body)
    return _SwinBlockFn.apply(x, prompt, ts, w, block.window_size, block.shift_size, dropout, pre)


# ----------------------------------------------------------------------------------------------
# patch merging
# ----------------------------------------------------------------------------------------------
class _PatchMergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, norm_w, norm_b, red_w, ln_w, ln_b, w, w_t, wgam, wbet, merge_last):
        y = ops.patch_merge(x, ln_w, ln_b, w, merge_last)
        ctx.save_for_backward(x, ln_w, ln_b, w_t, wgam, wbet, y)
        ctx.merge_last = merge_last
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ln_w, ln_b, w_t, wgam, wbet, y = ctx.saved_tensors
        g = ctx.needs_input_grad
        kw = dict(y_fwd=y, wgam=wgam, wbet=wbet)
        if g[1] or g[2] or g[3]:
            dx, dw, dgamma, dbeta = ops.patch_merge_backward(dy.contiguous(), x, ln_w, ln_b, w_t, ctx.merge_last, need_w=True, **kw)
            return (dx if g[0] else None, dgamma if g[1] else None, dbeta if g[2] else None, dw if g[3] else None,
                    None, None, None, None, None, None, None)
        dx = ops.patch_merge_backward(dy.contiguous(), x, ln_w, ln_b, w_t, ctx.merge_last, **kw)
        return dx, None, None, None, None, None, None, None, None, None, None


def patch_merge(mod, x):
    params = [mod.norm.weight, mod.norm.bias, mod.reduction.weight]

    def build():
        w = mod.reduction.weight.detach().float()
        ln_w, ln_b = mod.norm.weight.detach().float().contiguous(), mod.norm.bias.detach().float().contiguous()
        wb = w.to(BF16)
        wf = wb.float()                                         # the values the kernels multiply with
        return (ln_w, ln_b, wb.contiguous(), wb.t().contiguous(), (wf @ ln_w).contiguous(), (wf @ ln_b).contiguous())

    ln_w, ln_b, w, w_t, wgam, wbet = mod._wcache.get("w", params, build)
    return _PatchMergeFn.apply(x, mod.norm.weight, mod.norm.bias, mod.reduction.weight, ln_w, ln_b, w, w_t, wgam, wbet,
                               mod.merge_last_dim)


# ----------------------------------------------------------------------------------------------
# convolutions
# ----------------------------------------------------------------------------------------------
class _ConvFn(torch.autograd.Function):
    """Plain conv3d 3^3 (+bias) with an optional residual add: y = conv(x) + residual."""

    @staticmethod
    def forward(ctx, x, residual, conv_w, conv_b, wp, wd, bias, cout):
        ctx.wd = wd
        ctx.cin = x.shape[-1]
        ctx.cin_w = conv_w.shape[1]
        ctx.cout = cout
        ctx.has_res = residual is not None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            ctx.save_for_backward(x)
        return ops.conv3d(x, wp, bias, cout, residual=residual)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        g = ctx.needs_input_grad
        wd, cpad = ctx.wd
        dy_p = dy if cpad == dy.shape[-1] else torch.nn.functional.pad(dy, (0, cpad - dy.shape[-1]))
        dx = dw = db = None
        if g[0]:
            dx = ops.conv3d(dy_p, wd, None, ctx.cin)
        if g[2] or g[3]:
            (x,) = ctx.saved_tensors
            dw, db = ops.conv3d_wgrad(x, dy_p, ctx.cout, ctx.cin_w)
        dres = dy if (ctx.has_res and g[1]) else None
        return dx, dres, dw if g[2] else None, db if g[3] else None, None, None, None, None


def _conv_weights(cache: WeightCache, key, conv):
    def build():
        wp = ops.pack_conv_weight(conv.weight)
        wd = ops.pack_conv_weight_dgrad(conv.weight)
        b = conv.bias.detach().float().contiguous() if conv.bias is not None else None
        return wp, wd, b
    return cache.get(key, [conv.weight, conv.bias], build)


def conv3d_plain(owner, key, conv, x, residual=None):
    wp, wd, b = _conv_weights(owner._wcache, key, conv)
    if x.shape[-1] != wp.shape[1] // 27 and x.shape[-1] * 27 > wp.shape[1]:
        raise RuntimeError("conv3d_plain: channel mismatch")
    return _ConvFn.apply(x, residual, conv.weight, conv.bias, wp, wd, b, conv.out_channels)


class _BnActConvFn(torch.autograd.Function):
    """y = conv3x3x3( act( BatchNorm(x) ) ) with the BatchNorm affine + activation fused into the conv's
    operand load.  Training mode: batch statistics (+ running-stat update); eval: running statistics."""

    @staticmethod
    def forward(ctx, x, bn_w, bn_b, conv_w, conv_b, bn, wp, wd, lrelu, out_f32):
        if bn.training:
            scale, shift, mean_rstd = ops.bn_batch_stats(
                x, bn_w.detach().float().contiguous(), bn_b.detach().float().contiguous(), bn.eps,
                bn.running_mean, bn.running_var, bn_momentum(bn))
            bump_counter(bn.num_batches_tracked)
        else:
            scale, shift, mean_rstd = ops.bn_eval_affine(bn_w.detach(), bn_b.detach(), bn.running_mean, bn.running_var, bn.eps)
        cout = conv_w.shape[0]
        # one elementwise pass applies BatchNorm affine + activation; the conv then streams plain bf16 operands
        # (applying them inside the conv's operand load costs ~100 VALU ops per k-step, 27x per voxel)
        if out_f32 and not lrelu and ops.head_conv_supported(x.shape[-1], cout):
            y = ops.head_conv(x, conv_w, conv_b, scale, shift)          # segmentation head: fused BN-affine + conv
        else:
            y = ops.conv3d_bn_act(x, wp, conv_b.detach().float().contiguous(), cout, scale, shift, lrelu, out_f32)
        ctx.save_for_backward(x, scale, shift, mean_rstd)
        ctx.meta = (bn.training, lrelu, wd, cout, x.shape[-1])
        ctx.conv_w = conv_w
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean_rstd = ctx.saved_tensors
        training, lrelu, wd_pack, cout, cin = ctx.meta
        wd, cpad = wd_pack[0], wd_pack[1]
        wd16 = wd_pack[2] if len(wd_pack) > 2 else None
        need_x, need_bw, need_bb, need_cw, need_cb = ctx.needs_input_grad[:5]
        dy_p = dy
        if dy.dtype != BF16 or dy.shape[-1] != cpad or not dy.is_contiguous():
            dy_p = torch.zeros(dy.shape[:-1] + (cpad,), dtype=BF16, device=dy.device)
            dy_p[..., :cout] = dy
        dx = dgamma = dbeta = dw = db = None
        rows_ok = (not lrelu) and ops.conv3d_wgrad_rows_supported(cin, cout, x.shape[3])
        if need_x:
            if wd16 is not None and ops.halo_brick(x.shape[0], tuple(x.shape[1:4]), cin):
                dy16 = torch.zeros(dy.shape[:-1] + (16,), dtype=BF16, device=dy.device)     # one 16-channel halo chunk
                dy16[..., :cout] = dy
                dz = ops.conv3d(dy16, wd16, None, cin)
            else:
                dz = ops.conv3d(dy_p, wd, None, cin)        # gradient w.r.t. the conv operand act(BN(x))
            if training:
                dx, dgamma, dbeta = ops.bn_backward(x, dz, scale, shift, mean_rstd, lrelu)
            else:
                dx, dgamma, dbeta = ops.bn_backward_eval(x, dz, scale, shift, mean_rstd, lrelu)
        if need_cw or need_cb or ((need_bw or need_bb) and not need_x):
            if rows_ok:
                # one MFMA pass over (x, dy) gives every parameter gradient of the BN -> conv head
                G, S = ops.conv3d_wgrad_rows(x, dy_p, cout)
                dw, db, dg2, db2 = ops.head_grads_from_gs(G, S, ctx.conv_w, scale, shift, mean_rstd)
                if not need_x:
                    dgamma, dbeta = dg2, db2
            else:
                if (need_bw or need_bb) and not need_x:
                    dz = ops.conv3d(dy_p, wd, None, cin)
                    _, dgamma, dbeta = (ops.bn_backward if training else ops.bn_backward_eval)(x, dz, scale, shift, mean_rstd, lrelu)
                if need_cw or need_cb:              # recompute act(BN(x)), then the TN GEMM over voxels
                    dw, db = ops.conv3d_wgrad(ops.affine_act(x, scale, shift, lrelu), dy_p, cout)
        return (dx if need_x else None, dgamma if need_bw else None, dbeta if need_bb else None,
                dw if need_cw else None, db if need_cb else None, None, None, None, None, None)


def _bn_conv_weights(owner, key, conv):
    def build():
        wd, cpad = ops.pack_conv_weight_dgrad(conv.weight)
        return ops.pack_conv_weight(conv.weight), (wd, cpad, ops.pack_conv_weight_dgrad16(conv.weight))

    return owner._wcache.get(key, [conv.weight], build)


def bn_act_conv(owner, bn, conv, x, lrelu, out_f32=False, key=None):
    wp, wd = _bn_conv_weights(owner, key or "bn_conv", conv)
    return _BnActConvFn.apply(x, bn.weight, bn.bias, conv.weight, conv.bias, bn, wp, wd, lrelu, out_f32)


USE_FUSED_UPCAT_BN = True


def upcat_bn_act_conv(owner, bn, conv, x, skip, strides, lrelu=True, key=None):
    """SwinUpBlock's up -> cat -> norm_concat -> act -> conv_concat (unet_blocks.py:72-75).  When nothing on the way needs
    a gradient (frozen decoder without prompts: BASELINE configs[1]; evaluation) the concat tensor is never stored
    un-normalised: its BatchNorm statistics come from the sources (``mivp_upcat_stats``) and one pass writes
    act(BN(cat(up(x), skip))) for the conv (``mivp_upcat_affine_fwd``) -- three passes over the largest tensor of the
    stage fewer than upcat -> statistics -> affine_act.  Otherwise the differentiable pair below."""
    tensors = (x, skip, bn.weight, bn.bias, conv.weight, conv.bias)
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
    fits = skip is not None and x.shape[3] * (x.shape[-1] // 8) <= 2048 and x.shape[-1] <= 1536 and skip.shape[-1] <= 512
    if needs_grad or not USE_FUSED_UPCAT_BN or not fits:
        return bn_act_conv(owner, bn, conv, upcat(x, skip, strides), lrelu, key=key)
    require_device(x)
    wp, _ = _bn_conv_weights(owner, key or "bn_conv", conv)
    strides = tuple(int(s) for s in strides)
    with torch.no_grad():
        if bn.training:
            part, nblk, n_vox = ops.upcat_stats(x, skip, strides)
            scale, shift, _ = ops.bn_finalize(part, nblk, x.shape[-1] + skip.shape[-1], n_vox,
                                              bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous(),
                                              bn.eps, bn.running_mean, bn.running_var, bn_momentum(bn))
            bump_counter(bn.num_batches_tracked)
        else:
            scale, shift, _ = ops.bn_eval_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        z = ops.upcat_affine(x, skip, strides, scale, shift, lrelu)
        return ops.conv3d(z, wp, conv.bias.detach().float().contiguous(), conv.weight.shape[0])


class _UpHeadFn(torch.autograd.Function):
    """Segmentation head on the LOW-resolution decoder output: logits = conv(BatchNorm(upsample_x2(x))) without ever
    forming the upsampled tensor (csrc/uphead.hip): forward, the head's four parameter gradients and the gradient
    w.r.t. the low-resolution input."""

    @staticmethod
    def forward(ctx, x, bn_w, bn_b, conv_w, conv_b, bn):
        gx = None
        if bn.training:
            keep_gx = bool(ctx.needs_input_grad[0])               # the backward w.r.t. x reuses U^T U x from this pass
            res = ops.uphead_batch_stats(
                x, bn_w.detach().float().contiguous(), bn_b.detach().float().contiguous(), bn.eps,
                bn.running_mean, bn.running_var, bn_momentum(bn), keep_gx)
            scale, shift, mean_rstd = res[:3]
            gx = res[3] if keep_gx else None
            bump_counter(bn.num_batches_tracked)
        else:
            scale, shift, mean_rstd = ops.bn_eval_affine(bn_w.detach(), bn_b.detach(), bn.running_mean, bn.running_var, bn.eps)
        cout = conv_w.shape[0]
        y = ops.uphead_forward(x, ops.uphead_fold(conv_w, scale, shift), conv_b, cout)
        ctx.save_for_backward(x, scale, shift, mean_rstd, conv_w)
        ctx.gx = gx
        ctx.cout = cout
        ctx.training = bn.training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean_rstd, conv_w = ctx.saved_tensors
        g = ctx.needs_input_grad
        if g[0]:
            G, S, D = ops.uphead_gs(x, dy, ctx.cout, keep_d=True, raw=True)
        else:
            G, S = ops.uphead_gs(x, dy, ctx.cout, raw=True)
        dw, db, dgamma, dbeta = ops.head_grads_fused(G, S, conv_w, scale, shift, mean_rstd)
        dx = ops.uphead_dx(x, D, conv_w, scale, mean_rstd, dgamma, dbeta, ctx.training, ctx.gx) if g[0] else None
        ctx.gx = None
        return (dx, dgamma if g[1] else None, dbeta if g[2] else None, dw if g[3] else None, db if g[4] else None, None)


def uphead_applicable(x, bn, conv) -> bool:
    """The low-res head path: x2 trilinear upsample straight into (BatchNorm -> conv 3^3) with few classes."""
    return ops.uphead_supported(x.shape[-1], conv.out_channels)


def uphead(bn, conv, x):
    return _UpHeadFn.apply(x, bn.weight, bn.bias, conv.weight, conv.bias, bn)


# ----------------------------------------------------------------------------------------------
# upsample + crop + concat
# ----------------------------------------------------------------------------------------------
class _UpcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip, scale, align_corners):
        ctx.meta = (tuple(x.shape[1:4]), tuple(scale), x.shape[-1], 0 if skip is None else skip.shape[-1], align_corners)
        return ops.upcat(x, skip, scale, align_corners=align_corners)

    @staticmethod
    def backward(ctx, dy):
        idims, scale, cx, cs, align = ctx.meta
        dx, dskip = ops.upcat_backward(dy.contiguous(), idims, scale, cx, cs,
                                       need_skip=cs > 0 and ctx.needs_input_grad[1], align_corners=align)
        return (dx if ctx.needs_input_grad[0] else None), dskip, None, None


def upcat(x, skip, scale, align_corners=False):
    return _UpcatFn.apply(x, skip, tuple(int(s) for s in scale), bool(align_corners))


# ----------------------------------------------------------------------------------------------
# phase-1 proxy heads (--training-mode self_supervised_learning_encoder, swin_unetr.py:64-83,180-222)
# ----------------------------------------------------------------------------------------------
class _InstanceNormActFn(torch.autograd.Function):
    """nn.InstanceNorm3d (no affine) [+ nn.LeakyReLU(0.01)] on channels-last bf16."""

    @staticmethod
    def forward(ctx, x, eps, lrelu=True):
        y, saved = ops.instance_norm_act(x, eps, lrelu)
        ctx.save_for_backward(x)
        ctx.saved = saved
        ctx.lrelu = lrelu
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.instance_norm_act_backward(x, dy.contiguous(), ctx.saved, ctx.lrelu), None, None


def _conv1x1_weights(cache: WeightCache, key, conv):
    """A 1^3 convolution run by the 3^3 kernels: its weight sits on the centre tap (autograd slices the gradient back)."""
    def build():
        w27 = torch.nn.functional.pad(conv.weight.detach(), (1, 1, 1, 1, 1, 1))
        return ops.pack_conv_weight(w27), ops.pack_conv_weight_dgrad(w27)
    return cache.get(key, [conv.weight], build)


def unetr_basic_block(owner, key, block, x, layer=None):
    """MONAI ``UnetrBasicBlock`` (stride 1, norm 'instance', LeakyReLU 0.01) on channels-last bf16:
    conv 3^3 -> IN -> LReLU -> conv 3^3 -> IN [-> + (IN(conv 1^3(x)) | x)] -> LReLU.  ``layer``: the UnetResBlock /
    UnetBasicBlock module when it is not ``block.layer`` (UnetrUpBlock calls it ``conv_block``)."""
    lay = layer if layer is not None else block.layer
    cout = lay.conv1.conv.out_channels
    out = conv3d_plain(owner, f"{key}.c1", lay.conv1.conv, x)
    out = _InstanceNormActFn.apply(out, float(lay.norm1.eps), True)
    out = conv3d_plain(owner, f"{key}.c2", lay.conv2.conv, out)
    if not block.res_block:
        return _InstanceNormActFn.apply(out, float(lay.norm2.eps), True)
    out = _InstanceNormActFn.apply(out, float(lay.norm2.eps), False)
    if hasattr(lay, "conv3"):
        wp, wd = _conv1x1_weights(owner._wcache, f"{key}.c3", lay.conv3.conv)
        w27 = torch.nn.functional.pad(lay.conv3.conv.weight, (1, 1, 1, 1, 1, 1))
        res = _ConvFn.apply(x, None, w27, None, wp, wd, None, cout)
        res = _InstanceNormActFn.apply(res, float(lay.norm3.eps), False)
    else:
        res = x[..., :cout]
    return torch.nn.functional.leaky_relu(out + res, 0.01)


class _ConvTransposeFn(torch.autograd.Function):
    """nn.ConvTranspose3d with kernel == stride, no bias, on channels-last bf16 (csrc/convt.hip)."""

    @staticmethod
    def forward(ctx, x, weight, w1, w2, stride):
        ctx.save_for_backward(x)
        ctx.w2 = w2
        ctx.stride = stride
        ctx.cin = x.shape[-1]
        return ops.convt_forward(x, w1, stride, weight.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.convt_dgrad(dy, ctx.w2, ctx.stride, ctx.cin) if ctx.needs_input_grad[0] else None
        dw = ops.convt_wgrad(x, dy, ctx.stride) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


def conv_transpose(owner, key, conv: torch.nn.ConvTranspose3d, x):
    stride = tuple(int(s) for s in conv.stride)
    if tuple(conv.kernel_size) != stride or conv.bias is not None:
        raise NotImplementedError("mivp_amd: ConvTranspose3d with kernel == stride and no bias only (MONAI UnetrUpBlock's form)")
    w1, w2 = owner._wcache.get(key, [conv.weight], lambda: ops.pack_convt_weight(conv.weight))
    return _ConvTransposeFn.apply(x, conv.weight, w1, w2, stride)


def unetr_up_block(owner, key, block, x, skip):
    """MONAI ``UnetrUpBlock.forward(inp, skip)``: transposed conv (kernel = stride) -> cat([out, skip]) -> UnetResBlock /
    UnetBasicBlock (networks/blocks/unetr_block.py), on channels-last bf16."""
    up = conv_transpose(owner, f"{key}.t", block.transp_conv.conv, x)
    cat = torch.cat([up, skip], dim=-1)
    return unetr_basic_block(owner, f"{key}.c", block, cat, layer=block.conv_block)


class _PointwiseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return ops.pointwise_conv(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = ops.pointwise_conv_backward(x, w, dy, ctx.needs_input_grad[0])
        return dx, dw.reshape(w.shape) if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None


class _GlobalAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return ops.global_avg_pool(x)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, D, Cc = ctx.shape
        g = (dy / float(H * W * D)).to(BF16)
        return g.view(B, 1, 1, 1, Cc).expand(B, H, W, D, Cc).contiguous()


def reconstruction_head(owner, head: torch.nn.Sequential, x):
    """[Conv3d 3^3 -> InstanceNorm3d -> LeakyReLU -> Upsample(trilinear, align_corners=True)] x (depth+1) -> Conv3d 1^3
    (swin_unetr.py:185-212) on the deepest encoder feature; returns f32 channels-last [B,H,W,D,input_channels]."""
    mods = list(head)
    i = 0
    stage = 0
    while i + 3 < len(mods):
        conv, norm, act, up = mods[i:i + 4]
        x = conv3d_plain(owner, f"rec{stage}", conv, x)
        x = _InstanceNormActFn.apply(x, float(norm.eps), True)
        x = upcat(x, None, tuple(int(s) for s in up.scale_factor), align_corners=True)
        i += 4
        stage += 1
    last = mods[i]
    return _PointwiseConvFn.apply(x, last.weight, last.bias)


def pooled_linear(linear: torch.nn.Linear, x):
    """AdaptiveAvgPool3d((1,1,1)) -> Linear (rotation / contrastive heads, swin_unetr.py:72-81); the [B, C] matmul is torch's."""
    return torch.nn.functional.linear(_GlobalAvgPoolFn.apply(x), linear.weight, linear.bias)
