#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never has
it).  The reference's hot-path sub-packages need only torch + einops, so they
are imported through a synthetic parent package that skips
``modules/__init__.py`` (which pulls monai / cv2 / torchinfo, absent here):

    refmodules.swin_transformer.{swin_block,down}
    refmodules.multi_head_attention.{window_attention,relative_positional_encoding}

    refmodules.losses.clustered_prototype_loss, refmodules.momentum_model            (G8: students/teacher step)
    refmodules.utils  (G9: MeanIoU, DiceCoefficient, WarmupCosineSchedule, map_label_indices; the file imports cv2 at
                       the top for its PNG viewers only, so an EMPTY placeholder module named cv2 is registered first --
                       no cv2 function is ever called by the four symbols captured here)

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


def coord_grid(shape):
    """datasets/transforms.py:336-344 ``get_coord_grid`` for a [C, H, W, D] image shape (restated: that file imports MONAI)."""
    g = torch.stack(torch.meshgrid(torch.arange(shape[1]), torch.arange(shape[2]), torch.arange(shape[3]), indexing="ij"), 0).float()
    return g - torch.tensor([(shape[1] - 1) / 2., (shape[2] - 1) / 2., (shape[3] - 1) / 2.]).reshape(3, 1, 1, 1)


def gen_prototype_loss():
    """G8: ClusteredPrototypeLoss (losses/clustered_prototype_loss.py:13-206) forward + input gradients.  The student crops
    are sub-volumes of the teacher volume (their coordinate grids are crops of the teacher's), as the data pipeline makes
    them; the spatial jitter comes from torch's global RNG, so it is re-drawn here with the same seed and stored."""
    import math
    from refmodules.losses import ClusteredPrototypeLoss
    cases = {
        # name: (teacher dims, [student crop slices], channels, reduction_factor, fwhm, k-means iterations, temp_s, temp_t)
        "proto_a": ((16, 16, 8), [((2, 14), (1, 13), (0, 8)), ((0, 16), (0, 16), (0, 8))], 8, 4.0, 128.0, 3, 0.066, 0.033),
        "proto_b": ((12, 20, 12), [((0, 8), (4, 20), (2, 12)), ((3, 12), (0, 16), (0, 12))], 6, 2.0, 16.0, 2, 0.1, 0.05),
        "proto_c": ((24, 24, 16), [((0, 24), (0, 24), (0, 16)), ((4, 20), (4, 20), (2, 14))], 8, 4.0, 128.0, 3, 0.066, 0.033),
    }
    for name, (tdims, crops, C, red, fwhm, iters, ts, tt) in cases.items():
        gen = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        B = 2
        emb_t = torch.randn(B, C, *tdims, generator=gen).requires_grad_(True)
        coord_t = coord_grid((1,) + tdims)[None].repeat(B, 1, 1, 1, 1)
        emb_s, coord_s = [], []
        for cr in crops:
            sl = tuple(slice(a, b) for a, b in cr)
            dims = tuple(b - a for a, b in cr)
            emb_s.append(torch.randn(B, C, *dims, generator=gen).requires_grad_(True))
            coord_s.append(coord_t[(slice(None), slice(None)) + sl].clone())
        loss_fn = ClusteredPrototypeLoss(reduction_factor=red, k_means_iterations=iters, fwhm=fwhm)
        seed = 1000 + len(name)
        torch.manual_seed(seed)
        jit = [torch.randint(low=0, high=int(math.ceil(red)), size=(6,)) for _ in crops]      # the draws the loss makes
        torch.manual_seed(seed)
        loss = loss_fn(emb_s, emb_t, coord_s, coord_t, temp_s=ts, temp_t=tt)
        loss.backward()
        arrays = {"in/emb_t": emb_t, "in/coord_t": coord_t, "out/loss": loss.reshape(1), "grad/emb_t": emb_t.grad}
        for i in range(len(crops)):
            arrays[f"in/emb_s{i}"] = emb_s[i]
            arrays[f"in/coord_s{i}"] = coord_s[i]
            arrays[f"in/jitter{i}"] = jit[i]
            arrays[f"grad/emb_s{i}"] = emb_s[i].grad
        _save(name, arrays, {"reduction_factor": red, "fwhm": fwhm, "k_means_iterations": iters, "temp_s": ts, "temp_t": tt,
                             "n_students": len(crops)})


def gen_momentum():
    """G8: MomentumModel (momentum_model/momentum_model.py:4-36) with a toy architecture: construction from
    ``architecture(conf=conf)``, ``copy_state_dict`` and two EMA steps of ``update_teacher``."""
    from refmodules.momentum_model import MomentumModel

    class Toy(torch.nn.Module):
        def __init__(self, conf):
            super().__init__()
            self.a = torch.nn.Linear(5, 7)
            self.b = torch.nn.Conv3d(2, 3, 3)
            self.n = torch.nn.BatchNorm3d(3)

        def forward(self, x):
            return {"latent_outputs": x}

    torch.manual_seed(77)
    mm = MomentumModel(Namespace(tau=0.99), Toy)
    arrays = {}
    for k, v in mm.net_student.named_parameters():
        arrays[f"student0/{k}"] = v.detach().clone()
    for k, v in mm.net_teacher.named_parameters():
        arrays[f"teacher0/{k}"] = v.detach().clone()
    mm.update_teacher()
    for k, v in mm.net_teacher.named_parameters():
        arrays[f"teacher1/{k}"] = v.detach().clone()
    with torch.no_grad():
        for q in mm.net_student.parameters():
            q.add_(0.05 * torch.randn(q.shape))
    for k, v in mm.net_student.named_parameters():
        arrays[f"student1/{k}"] = v.detach().clone()
    mm.update_teacher()
    for k, v in mm.net_teacher.named_parameters():
        arrays[f"teacher2/{k}"] = v.detach().clone()
    mm.copy_state_dict()
    copied = all(torch.equal(a, b) for a, b in zip(mm.net_student.parameters(), mm.net_teacher.parameters()))
    frozen = all(not q.requires_grad for q in mm.net_teacher.parameters())
    outs, outt = mm([torch.ones(1), torch.zeros(1)], torch.full((1,), 2.0))
    _save("momentum_model", arrays, {"tau": 0.99, "copy_state_dict_copies": copied, "copy_state_dict_freezes_teacher": frozen,
                                     "forward_returns": [len(outs), sorted(outt.keys())],
                                     "param_order": [k for k, _ in mm.net_student.named_parameters()]})


def gen_utils():
    """G9: metrics / schedule / label mapping of modules/utils.py:14-89,372-388 (the reference's own code, see the module
    docstring for the empty cv2 placeholder)."""
    if "cv2" not in sys.modules:
        try:
            import cv2  # noqa: F401
        except ImportError:
            sys.modules["cv2"] = types.ModuleType("cv2")
    from refmodules import utils as U
    gen = torch.Generator().manual_seed(21)
    arrays, meta = {}, {}
    # metrics: two update() calls, then compute()
    for ncls in (2, 5):
        miou, dice = U.MeanIoU(ncls), U.DiceCoefficient(ncls)
        for step in range(2):
            preds = torch.randn(2, ncls, 6, 5, 4, generator=gen)
            target = torch.randint(0, ncls, (2, 1, 6, 5, 4), generator=gen).float()
            miou.update(preds, target)
            dice.update(preds, target)
            arrays[f"metrics{ncls}/preds{step}"] = preds
            arrays[f"metrics{ncls}/target{step}"] = target
        arrays[f"metrics{ncls}/miou"] = miou.compute().reshape(1)
        arrays[f"metrics{ncls}/dice"] = dice.compute().reshape(1)
    # schedule: the lr of both groups over 40 scheduler steps
    p1, p2 = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))
    opt = torch.optim.AdamW([{"params": [p1], "lr": 5e-4}, {"params": [p2], "lr": 1e-3}], lr=5e-4)
    sched = U.WarmupCosineSchedule(opt, warmup_steps=10, t_total=30)
    lrs = []
    for _ in range(40):
        lrs.append([g["lr"] for g in opt.param_groups])
        opt.step()
        sched.step()
    arrays["sched/lrs"] = torch.tensor(lrs, dtype=torch.float64)
    meta["sched"] = {"warmup_steps": 10, "t_total": 30, "base_lrs": [5e-4, 1e-3]}
    # label mapping (the yml's active_labels_pretrain / _downstream)
    masks = torch.randint(0, 7, (2, 1, 5, 4, 3), generator=gen).float()
    arrays["labels/in"] = masks.clone()
    arrays["labels/pretrain"] = U.map_label_indices(masks.clone(), [0, 1, 2, 3, 5])
    arrays["labels/downstream"] = U.map_label_indices(masks.clone(), [5, 0])
    _save("utils_metrics_schedule", arrays, meta)


if __name__ == "__main__":
    torch.set_num_threads(8)
    sb, down, wa, rpe, ub, su = _import_reference()
    if "--only-g8g9" in sys.argv:
        gen_prototype_loss()
        gen_momentum()
        gen_utils()
        sys.exit(0)
    gen_relpe(rpe)
    gen_mask(sb)
    gen_attention(wa)
    gen_block(sb)
    gen_merge(down)
    gen_upblock(ub)
    gen_unetr(su)
    gen_prototype_loss()
    gen_momentum()
    gen_utils()
