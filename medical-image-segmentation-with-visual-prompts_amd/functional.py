"""Autograd glue between the nn.Module surface (swin_unetr.py) and the HIP kernels.

Each fused stage is one ``torch.autograd.Function`` whose forward/backward launch kernels through
``ops`` / ``swin_ops``.  Activations between stages are channels-last bf16 tensors
``[B, H, W, D, C]``; parameters stay fp32 masters in the modules and are re-laid-out for the
kernels by a small per-module cache that is refreshed when a parameter's version changes
(optimizer step, ``load_state_dict``, ``.to()``).

Gradients built so far: data gradients through every stage, prompt-token and prompt-bias
gradients of the Swin blocks, BatchNorm affine gradients, weight/bias gradients of the small
segmentation-head convolutions -- i.e. everything ``--training-mode downstream`` trains
(swin_unetr.py:33-40, segmentation.py:25-39).  A stage asked for a gradient it has no kernel for
raises NotImplementedError at forward time instead of silently training nothing.
"""
from __future__ import annotations

from typing import Optional

import torch

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


class WeightCache:
    """key -> kernel-ready tensors, rebuilt when any source parameter changed."""

    def __init__(self):
        self._store = {}

    @staticmethod
    def _stamp(params):
        return tuple((p._version, p.data_ptr(), str(p.device)) for p in params if p is not None)

    def get(self, key, params, builder):
        stamp = self._stamp(params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            val = builder()
        self._store[key] = (stamp, val)
        return val


def _no_grad_kernel(what: str, *params):
    """Fail loudly when a parameter that wants a gradient has no backward kernel yet."""
    if torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params):
        raise NotImplementedError(
            f"mivp_amd: weight gradients of {what} are not built yet (only the parameter set of "
            "--training-mode downstream trains on the HIP path so far: prompt tokens, prompt bias, "
            "segmentation head). Freeze these parameters or run under torch.no_grad().")


# ----------------------------------------------------------------------------------------------
# patch embedding + BatchNorm
# ----------------------------------------------------------------------------------------------
def patch_embed(owner, conv, bn, x):
    _no_grad_kernel("input_layer (patch embedding)", conv.weight, conv.bias, bn.weight, bn.bias)
    if x.requires_grad:
        raise NotImplementedError("mivp_amd: gradient w.r.t. the input volume is not built")
    training = bn.training
    with torch.no_grad():
        y = ops.patch_embed(x.detach(), conv.weight, conv.bias, bn.weight.detach().float().contiguous(),
                            bn.bias.detach().float().contiguous(), bn.eps, bn.running_mean, bn.running_var,
                            training=training, momentum=bn.momentum if bn.momentum is not None else 0.1)
        if training:
            bn.num_batches_tracked += 1
    return y


# ----------------------------------------------------------------------------------------------
# Swin block
# ----------------------------------------------------------------------------------------------
class _SwinBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, prompt, ts, w, window, shift):
        need = (ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        y, saved = swin_ops.swin_block_forward(x, prompt, w, ts, window, shift, save=need)
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
        return dx, dprompt, dts, None, None, None


def swin_block(block, x, prompt: Optional[torch.Tensor]):
    pe, attn = block.pe, block.attn
    body = [block.attn_norm.weight, block.attn_norm.bias, attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
            attn.proj.weight, attn.proj.bias, block.mlp_norm.weight, block.mlp_norm.bias, block.mlp.weight,
            block.mlp.bias]
    content = [pe.enc_content_h, pe.enc_content_w, pe.enc_content_d, pe.weights_content_h, pe.weights_content_w,
               pe.weights_content_d]
    _no_grad_kernel("a Swin block's LayerNorm/Linear/relative-bias weights", *body, *content)
    n_prompt = 0 if prompt is None else int(prompt.shape[0])
    need_bwd = torch.is_grad_enabled()

    def build():
        sd = {k: v for k, v in block.state_dict().items()}
        return swin_ops.weights_from_state(sd, "", block.num_heads, block.embed_dim, 0, x.device, need_bwd=True)

    w = block._wcache.get("w", body + content, build)
    ts = None
    if n_prompt:
        if not pe.use_token_params:
            raise RuntimeError("prompt tokens passed to a block built without token bias parameters")
        ts = pe.token_scores(n_prompt)            # tiny torch matmul: autograd carries d(ts) into the two params
    return _SwinBlockFn.apply(x, prompt, ts, w, block.window_size, block.shift_size)


# ----------------------------------------------------------------------------------------------
# patch merging
# ----------------------------------------------------------------------------------------------
class _PatchMergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w, w_t, merge_last):
        ctx.save_for_backward(x, ln_w, ln_b, w_t)
        ctx.merge_last = merge_last
        return ops.patch_merge(x, ln_w, ln_b, w, merge_last)

    @staticmethod
    def backward(ctx, dy):
        x, ln_w, ln_b, w_t = ctx.saved_tensors
        dx = ops.patch_merge_backward(dy.contiguous(), x, ln_w, ln_b, w_t, ctx.merge_last)
        return dx, None, None, None, None, None


def patch_merge(mod, x):
    params = [mod.norm.weight, mod.norm.bias, mod.reduction.weight]
    _no_grad_kernel("PatchMerging", *params)

    def build():
        w = mod.reduction.weight.detach().float()
        return (mod.norm.weight.detach().float().contiguous(), mod.norm.bias.detach().float().contiguous(),
                w.to(BF16).contiguous(), w.t().to(BF16).contiguous())

    ln_w, ln_b, w, w_t = mod._wcache.get("w", params, build)
    return _PatchMergeFn.apply(x, ln_w, ln_b, w, w_t, mod.merge_last_dim)


# ----------------------------------------------------------------------------------------------
# convolutions
# ----------------------------------------------------------------------------------------------
class _ConvFn(torch.autograd.Function):
    """Plain conv3d 3^3 (+bias) with an optional residual add: y = conv(x) + residual."""

    @staticmethod
    def forward(ctx, x, residual, wp, wd, bias, cout):
        ctx.wd = wd
        ctx.cin = x.shape[-1]
        ctx.has_res = residual is not None
        return ops.conv3d(x, wp, bias, cout, residual=residual)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            wd, cpad = ctx.wd
            if cpad != dy.shape[-1]:
                dy_p = torch.nn.functional.pad(dy, (0, cpad - dy.shape[-1]))
            else:
                dy_p = dy
            dx = ops.conv3d(dy_p, wd, None, ctx.cin)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[1]) else None
        return dx, dres, None, None, None, None


def _conv_weights(cache: WeightCache, key, conv):
    def build():
        wp = ops.pack_conv_weight(conv.weight)
        wd = ops.pack_conv_weight_dgrad(conv.weight)
        b = conv.bias.detach().float().contiguous() if conv.bias is not None else None
        return wp, wd, b
    return cache.get(key, [conv.weight, conv.bias], build)


def conv3d_plain(owner, key, conv, x, residual=None):
    _no_grad_kernel(f"conv '{key}'", conv.weight, conv.bias)
    wp, wd, b = _conv_weights(owner._wcache, key, conv)
    if x.shape[-1] != wp.shape[1] // 27 and x.shape[-1] * 27 > wp.shape[1]:
        raise RuntimeError("conv3d_plain: channel mismatch")
    return _ConvFn.apply(x, residual, wp, wd, b, conv.out_channels)


class _BnActConvFn(torch.autograd.Function):
    """y = conv3x3x3( act( BatchNorm(x) ) ) with the BatchNorm affine + activation fused into the conv's
    operand load.  Training mode: batch statistics (+ running-stat update); eval: running statistics."""

    @staticmethod
    def forward(ctx, x, bn_w, bn_b, conv_w, conv_b, bn, wp, wd, lrelu, out_f32):
        if bn.training:
            scale, shift, mean_rstd = ops.bn_batch_stats(
                x, bn_w.detach().float().contiguous(), bn_b.detach().float().contiguous(), bn.eps,
                bn.running_mean, bn.running_var, bn.momentum if bn.momentum is not None else 0.1)
            bn.num_batches_tracked += 1
        else:
            scale, shift, mean_rstd = ops.bn_eval_affine(bn_w.detach(), bn_b.detach(), bn.running_mean, bn.running_var, bn.eps)
        cout = conv_w.shape[0]
        # one elementwise pass applies BatchNorm affine + activation; the conv then streams plain bf16 operands
        # (applying them inside the conv's operand load costs ~100 VALU ops per k-step, 27x per voxel)
        if out_f32 and not lrelu and ops.head_conv_supported(x.shape[-1], cout):
            y = ops.head_conv(x, conv_w, conv_b, scale, shift)          # segmentation head: fused BN-affine + conv
        else:
            xa = ops.affine_act(x, scale, shift, lrelu)
            y = ops.conv3d(xa, wp, conv_b.detach().float().contiguous(), cout, None, None, False, None, out_f32)
        ctx.save_for_backward(x, scale, shift, mean_rstd)
        ctx.meta = (bn.training, lrelu, wd, cout, x.shape[-1])
        ctx.conv_w = conv_w
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean_rstd = ctx.saved_tensors
        training, lrelu, (wd, cpad), cout, cin = ctx.meta
        need_x, need_bw, need_bb, need_cw, need_cb = ctx.needs_input_grad[:5]
        dy_p = dy
        if dy.dtype != BF16 or dy.shape[-1] != cpad or not dy.is_contiguous():
            dy_p = torch.zeros(dy.shape[:-1] + (cpad,), dtype=BF16, device=dy.device)
            dy_p[..., :cout] = dy
        dx = dgamma = dbeta = dw = db = None
        rows_ok = (not lrelu) and cout <= 5 and cin < 64
        if need_x:
            dz = ops.conv3d(dy_p, wd, None, cin)            # gradient w.r.t. the conv operand act(BN(x))
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
                if cout > 8:
                    raise NotImplementedError("mivp_amd: conv weight gradient with Cout > 8 is not built yet")
                if (need_bw or need_bb) and not need_x:
                    dz = ops.conv3d(dy_p, wd, None, cin)
                    _, dgamma, dbeta = (ops.bn_backward if training else ops.bn_backward_eval)(x, dz, scale, shift, mean_rstd, lrelu)
                if need_cw or need_cb:
                    dw, db = ops.conv3d_wgrad_small(x, scale, shift, lrelu, dy_p, cout)
        return (dx if need_x else None, dgamma if need_bw else None, dbeta if need_bb else None,
                dw if need_cw else None, db if need_cb else None, None, None, None, None, None)


def bn_act_conv(owner, bn, conv, x, lrelu, out_f32=False, key=None):
    key = key or "bn_conv"
    if conv.out_channels > 8:
        _no_grad_kernel(f"conv '{key}'", conv.weight, conv.bias)

    def build():
        return ops.pack_conv_weight(conv.weight), ops.pack_conv_weight_dgrad(conv.weight)

    wp, wd = owner._wcache.get(key, [conv.weight], build)
    return _BnActConvFn.apply(x, bn.weight, bn.bias, conv.weight, conv.bias, bn, wp, wd, lrelu, out_f32)


# ----------------------------------------------------------------------------------------------
# upsample + crop + concat
# ----------------------------------------------------------------------------------------------
class _UpcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip, scale):
        ctx.meta = (tuple(x.shape[1:4]), tuple(scale), x.shape[-1], 0 if skip is None else skip.shape[-1])
        return ops.upcat(x, skip, scale)

    @staticmethod
    def backward(ctx, dy):
        idims, scale, cx, cs = ctx.meta
        dx, dskip = ops.upcat_backward(dy.contiguous(), idims, scale, cx, cs,
                                       need_skip=cs > 0 and ctx.needs_input_grad[1])
        return (dx if ctx.needs_input_grad[0] else None), dskip, None


def upcat(x, skip, scale):
    return _UpcatFn.apply(x, skip, tuple(int(s) for s in scale))
