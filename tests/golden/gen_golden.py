#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never has
it).  The reference's hot-path sub-packages need only torch + einops, so they
are imported through a synthetic parent package that skips
``modules/__init__.py`` (which pulls monai / cv2 / torchinfo, absent here):

    refmodules.swin_transformer.{swin_block,down}
    refmodules.multi_head_attention.{window_attention,relative_positional_encoding}

``swin_unetr/unet_blocks.py`` and ``swin_unetr/swin_unetr.py`` import three MONAI
factories.  For those two files only, a 3-symbol stand-in (``_monai_standin``,
own code) maps the factories to the stock torch modules MONAI resolves them to
(nn.LeakyReLU(0.01), nn.BatchNorm3d, nn.Conv3d(k3,p1,bias) under child name
``conv``).  Fixtures made through it (``upblock_*``, ``unetr_*``) are therefore
"parity unpinned at the MONAI boundary"; everything else is pinned by the
reference's own code with no shim.

Only DATA is written (inputs, weights, outputs, gradients) -- never reference
source.  Usage:  python tests/golden/gen_golden.py
"""
import json
import os
import sys
import types
import zlib
from argparse import Namespace

import numpy as np
import torch

REF = "/root/reference/src/modules"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    parent = types.ModuleType("refmodules")
    parent.__path__ = [REF]
    sys.modules["refmodules"] = parent
    sys.path.insert(0, OUT)
    import _monai_standin
    _monai_standin.install()
    from refmodules.swin_transformer import swin_block as sb
    from refmodules.swin_transformer import down
    from refmodules.multi_head_attention import window_attention as wa
    from refmodules.multi_head_attention import relative_positional_encoding as rpe
    from refmodules.swin_unetr import unet_blocks as ub
    from refmodules.swin_unetr import swin_unetr as su
    return sb, down, wa, rpe, ub, su


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, arrays, meta):
    arrays = {k: (v if isinstance(v, np.ndarray) else _np(v)) for k, v in arrays.items()}
    arrays["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def _randomize(module, gen, scale=1.0):
    """Re-draw every float parameter (LayerNorm/BN weights included) so no
    identity weights hide an error."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() == 1 and ("norm" in n and n.endswith("weight")):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(p + 0.0)  # keep the module's own init for matrices
        for n, b in module.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=gen))
            elif n.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=gen))


def gen_relpe(rpe):
    for tag, window, tokens in [("w332", (3, 3, 2), 8), ("w777", (7, 7, 7), 64), ("w884", (8, 8, 4), 64),
                                ("w332_notok", (3, 3, 2), 0)]:
        torch.manual_seed(11)
        heads = 4
        m = rpe.RelativePE(embed_dim=64, num_heads=heads, max_abs_pos=window, max_cap_dist=window,
                           max_prompts=1, tokens_per_prompt=max(tokens, 1), use_token_params=tokens > 0)
        out = m(window[0], window[1], window[2], tokens)[0]          # [heads, N+t, N+t]
        N = window[0] * window[1] * window[2]
        arrays = {f"sd/pe.{k}": v for k, v in m.state_dict().items()}
        rows = torch.arange(0, N, 17 if N > 64 else 1)
        cols = torch.arange(0, N + tokens, 13 if N > 64 else 1)
        arrays["rows"] = rows
        arrays["cols"] = cols
        arrays["out/sub"] = out[:, rows][:, :, cols]
        arrays["out/rowsum"] = out[:, :N].double().sum(-1)          # full-matrix checksum per row
        arrays["out/colsum"] = out[:, :N].double().sum(-2)
        if tokens:
            arrays["out/prompt_rows_absmax"] = out[:, N:].abs().max().reshape(1)
        _save(f"relpe_{tag}", arrays, {"window": window, "tokens": tokens, "heads": heads, "embed_dim": 64})


def gen_mask(sb):
    import math
    cases = [
        ("a", (6, 6, 4), (3, 3, 2), (1, 1, 1)),      # no padding, shift on every axis
        ("b", (5, 6, 4), (3, 3, 2), (1, 1, 1)),      # odd pad on axis 0, full-window pad on the others
        ("c", (3, 6, 4), (3, 3, 2), (0, 1, 1)),      # degenerate shift 0 on axis 0 (dim <= window)
        ("d", (7, 7, 5), (3, 3, 2), (1, 1, 1)),      # every axis padded, mixed parity
        ("e", (8, 8, 8), (4, 4, 2), (2, 2, 1)),
        ("f", (4, 4, 8), (4, 4, 2), (0, 0, 1)),      # two axes un-shifted
    ]
    for tag, dims, w, s in cases:
        paddings = (0, 0, 0, 0, 0, 0)
        if any(d % ww != 0 for d, ww in zip(dims, w)):
            paddings = []
            for d, ww in zip(dims, w):
                t = ww - d % ww
                paddings += [math.floor(t / 2), math.ceil(t / 2)]
        shape_p = tuple(d + paddings[2 * a] + paddings[2 * a + 1] for a, d in enumerate(dims))
        mask = sb.get_attn_mask(shape_x=shape_p, window_size=w, shift_size=s, paddings=paddings)
        _save(f"mask_{tag}", {"out/mask": _np(mask[0]).astype(np.uint8)},
              {"dims": dims, "window": w, "shift": s, "paddings": list(paddings), "padded": shape_p})


def gen_attention(wa):
    torch.manual_seed(5)
    B, P, Nq, Np, C, heads = 1, 2, 24, 8, 24, 2
    N = Nq + Np
    m = wa.WindowAttention(dim=C, num_heads=heads)
    for tag, use_bias, use_mask in [("plain", False, False), ("bias_mask", True, True)]:
        x = torch.randn(B, P, N, C, requires_grad=True)
        bias = 0.3 * torch.randn(1, 1, heads, N, N) if use_bias else None
        mask = (torch.rand(1, P, 1, N, N) > 0.4).float() if use_mask else None
        out = m(x, x, x, pos_bias=bias, mask=mask)
        g = torch.randn_like(out)
        m.zero_grad()
        out.backward(g)
        arrays = {f"sd/attn.{k}": v for k, v in m.state_dict().items()}
        arrays.update({"in/x": x, "in/gout": g, "out/y": out, "grad/x": x.grad})
        for k, p in m.named_parameters():
            arrays[f"grad/attn.{k}"] = p.grad
        if use_bias:
            arrays["in/bias"] = bias[0, 0]
        if use_mask:
            arrays["in/mask"] = mask[0, :, 0]
        _save(f"attn_{tag}", arrays, {"heads": heads, "n_query": N})


def gen_block(sb):
    # (tag, dims, window, shift, prompts, C, heads)
    cases = [
        ("nopad_noshift", (6, 6, 4), (3, 3, 2), (0, 0, 0), 0, 8, 2),
        ("nopad_shift", (6, 6, 4), (3, 3, 2), (1, 1, 1), 0, 8, 2),
        ("nopad_shift_prompt", (6, 6, 4), (3, 3, 2), (1, 1, 1), 8, 8, 2),
        ("oddpad_shift_prompt", (5, 6, 4), (3, 3, 2), (1, 1, 1), 8, 8, 2),   # odd pad + full-window pads
        ("evenpad_noshift_prompt", (4, 4, 4), (3, 3, 3), (0, 0, 0), 8, 8, 2),  # t=2 on every axis
        ("smalldim_shift", (3, 6, 4), (3, 3, 2), (1, 1, 1), 0, 8, 2),        # dim <= window on axis 0
        ("smalldim_pad_prompt", (2, 5, 4), (3, 3, 2), (1, 1, 1), 8, 8, 2),   # dim < window and padded
        ("w442_shift_prompt", (8, 8, 6), (4, 4, 2), (2, 2, 1), 16, 16, 4),
    ]
    for tag, dims, w, s, n_prompt, C, heads in cases:
        torch.manual_seed(zlib.crc32(tag.encode()) % 1000)
        gen = torch.Generator().manual_seed(3)
        blk = sb.SwinTransformerBlock(hidden_channels=C, window_size=w, pos_bias_embed_dim=64, num_heads=heads,
                                      max_prompts=1, tokens_per_prompt=max(n_prompt, 1),
                                      use_token_params=n_prompt > 0, shift_size=s)
        _randomize(blk, gen)
        B = 2
        x = torch.randn(B, C, *dims, requires_grad=True)
        prm = None
        if n_prompt:
            prm = torch.nn.Parameter(0.5 * torch.randn(n_prompt, C))
        p = prm.unsqueeze(0).repeat(B, 1, 1) if prm is not None else None
        out = blk(x, p)
        g = torch.randn_like(out)
        out.backward(g)
        arrays = {f"sd/{k}": v for k, v in blk.state_dict().items()}
        arrays.update({"in/x": x, "in/gout": g, "out/y": out, "grad/x": x.grad})
        if prm is not None:
            arrays["in/prompt"] = prm
            arrays["grad/prompt"] = prm.grad
        for k, q in blk.named_parameters():
            arrays[f"grad/{k}"] = q.grad
        _save(f"block_{tag}", arrays, {"dims": dims, "window": w, "shift": s, "n_prompt": n_prompt,
                                       "C": C, "heads": heads})


def gen_merge(down):
    for tag, dims, last in [("even_T", (6, 6, 6), True), ("odd_T", (5, 7, 6), True),
                            ("even_F", (6, 6, 5), False), ("odd_F", (5, 6, 7), False)]:
        torch.manual_seed(9)
        gen = torch.Generator().manual_seed(4)
        C = 8
        m = down.PatchMerging(in_channels=C, out_channels=2 * C, merge_last_dim=last)
        _randomize(m, gen)
        x = torch.randn(2, C, *dims, requires_grad=True)
        out = m(x)
        g = torch.randn_like(out)
        out.backward(g)
        arrays = {f"sd/{k}": v for k, v in m.state_dict().items()}
        arrays.update({"in/x": x, "in/gout": g, "out/y": out, "grad/x": x.grad})
        for k, q in m.named_parameters():
            arrays[f"grad/{k}"] = q.grad
        _save(f"merge_{tag}", arrays, {"dims": dims, "merge_last_dim": last, "C": C})


def gen_upblock(ub):
    for tag, in_dims, skip_dims, strides, training in [
        ("s221_train", (3, 3, 4), (6, 6, 4), (2, 2, 1), True),
        ("s222_eval", (3, 3, 2), (6, 6, 4), (2, 2, 2), False),
        ("s222_crop_train", (3, 4, 2), (5, 7, 4), (2, 2, 2), True),
    ]:
        torch.manual_seed(21)
        gen = torch.Generator().manual_seed(6)
        cin, cout = 16, 8
        m = ub.SwinUpBlock(in_channels=cin, out_channels=cout, strides=strides, kernel_size=(3, 3, 3),
                           pos_bias_embed_dim=64, num_heads=2, window_size=(3, 3, 2), max_prompts=1,
                           tokens_per_prompt=8, use_token_params=True)
        _randomize(m, gen)
        m.train(training)
        sd_before = {k: v.clone() for k, v in m.state_dict().items()}
        B = 2
        x = torch.randn(B, cin, *in_dims, requires_grad=True)
        skip = torch.randn(B, cin // 2, *skip_dims, requires_grad=True)
        prm = [torch.nn.Parameter(0.5 * torch.randn(8, cout)) for _ in range(2)]
        out = m(x, skip, [q.unsqueeze(0).repeat(B, 1, 1) for q in prm])
        g = torch.randn_like(out)
        out.backward(g)
        arrays = {f"sd/{k}": v for k, v in sd_before.items()}
        arrays.update({"in/x": x, "in/skip": skip, "in/gout": g, "out/y": out,
                       "grad/x": x.grad, "grad/skip": skip.grad,
                       "in/prompt0": prm[0], "in/prompt1": prm[1],
                       "grad/prompt0": prm[0].grad, "grad/prompt1": prm[1].grad})
        for k, v in m.state_dict().items():
            if "running_" in k:
                arrays[f"after/{k}"] = v
        for k, q in m.named_parameters():
            arrays[f"grad/{k}"] = q.grad
        _save(f"upblock_{tag}", arrays, {"strides": strides, "training": training, "window": (3, 3, 2),
                                         "heads": 2, "cin": cin, "cout": cout,
                                         "note": "MONAI stand-in used: parity unpinned at the MONAI boundary"})


def tiny_conf(mode, ep, dp, res="none"):
    return Namespace(
        training_mode=mode, input_channels=1, depth_unet=3, hidden_channels=[8, 16, 32, 64],
        input_patch_size=[2, 2, 2], unetr_res_block=res, unetr_up_block="swin", basic_block_res=True,
        num_heads_encoder=2, num_heads_decoder=2, attn_window_size=[4, 4, 2], pos_bias_embed_dim=64,
        use_checkpoint=False, attn_drop=0.0, proj_drop=0.0, max_prompts=1,
        tokens_per_prompt_encoder=8, tokens_per_prompt_decoder=8,
        use_encoder_prompting=ep, use_decoder_prompting=dp,
        use_reconstruction=False, use_mutual_learning=False, use_rotation_prediction=False,
        use_contrastive_learning=False, contrastive_coding_dim=32,
        output_channels_downstream=2, output_channels_pretrain=5)


def gen_unetr(su):
    combos = [("downstream", e, d, "none") for e in (False, True) for d in (False, True)]
    combos += [("self_supervised_learning_all", True, False, "none"),
               ("self_supervised_learning_decoder", True, True, "none"),
               ("supervised_learning_all", False, False, "none"),
               ("downstream", True, True, "simple")]
    for mode, ep, dp, res in combos:
        torch.manual_seed(33)
        gen = torch.Generator().manual_seed(8)
        conf = tiny_conf(mode, ep, dp, res)
        m = su.SwinUnetR(conf)
        _randomize(m, gen)
        m.train()
        sd_before = {k: v.clone() for k, v in m.state_dict().items()}
        x = torch.rand(2, 1, 16, 16, 16)
        out = m(x)
        key = "downstream" if mode == "downstream" else "latent_outputs"
        y = out[key]
        g = torch.randn_like(y) / y.numel() ** 0.5
        loss_terms = (y * g).sum()
        if "seg_pred" in out:
            g2 = torch.randn_like(out["seg_pred"]) / out["seg_pred"].numel() ** 0.5
            loss_terms = loss_terms + (out["seg_pred"] * g2).sum()
        loss_terms.backward()
        arrays = {f"sd/{k}": v for k, v in sd_before.items()}
        arrays.update({"in/x": x, "in/gout": g, f"out/{key}": y})
        if "seg_pred" in out:
            arrays["in/gout_seg"] = g2
            arrays["out/seg_pred"] = out["seg_pred"]
        for k, v in m.state_dict().items():
            if "running_" in k or "num_batches" in k:
                arrays[f"after/{k}"] = v
        trainable = []
        for k, q in m.named_parameters():
            if q.requires_grad:
                trainable.append(k)
                arrays[f"grad/{k}"] = q.grad if q.grad is not None else torch.zeros_like(q)
        groups = {}
        for gname in ("named_parameters_downstream", "named_parameters_encoder", "named_parameters_decoder",
                      "named_parameters_prompt_tokens_encoder", "named_parameters_prompt_tokens_decoder"):
            try:
                plist = getattr(m, gname)()
            except Exception as exc:  # e.g. KeyError when the flag is off
                groups[gname] = f"raises {type(exc).__name__}"
                continue
            ids = {id(q): k for k, q in m.named_parameters()}
            groups[gname] = [ids[id(q)] for _, q in plist]
        tag = f"{mode}_e{int(ep)}d{int(dp)}" + ("" if res == "none" else f"_{res}")
        _save(f"unetr_{tag}", arrays, {
            "conf": vars(conf), "trainable": trainable, "groups": groups,
            "state_keys": [[k, list(v.shape), str(v.dtype)] for k, v in sd_before.items()],
            "param_order": [k for k, _ in m.named_parameters()],
            "note": "MONAI stand-in used: parity unpinned at the MONAI boundary"})


if __name__ == "__main__":
    torch.set_num_threads(8)
    sb, down, wa, rpe, ub, su = _import_reference()
    gen_relpe(rpe)
    gen_mask(sb)
    gen_attention(wa)
    gen_block(sb)
    gen_merge(down)
    gen_upblock(ub)
    gen_unetr(su)
