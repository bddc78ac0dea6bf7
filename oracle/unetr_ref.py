"""Oracle (test infrastructure): functional fp32 Swin-UNETR over a flat state dict.

Follows swin_unetr/swin_unetr.py:46-144 (forward paths) and :146-431 (module
tree, used here only for key names / shapes when building a random state).
The state dict uses the reference's key names.
"""
from __future__ import annotations

import math
from argparse import Namespace
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import swin_ref as S

Tensor = torch.Tensor

TRAINING_MODES = (
    "self_supervised_learning_encoder",
    "self_supervised_learning_decoder",
    "self_supervised_learning_all",
    "supervised_learning_decoder",
    "supervised_learning_all",
    "downstream",
)


def default_conf(**over) -> Namespace:
    """The hot-path fields of configurations/example_configs.yml:1-23,95-104."""
    conf = Namespace(
        training_mode="downstream",
        input_channels=1, depth_unet=3, hidden_channels=[48, 96, 192, 384],
        input_patch_size=[2, 2, 2], unetr_res_block="none", unetr_up_block="swin", basic_block_res=True,
        num_heads_encoder=4, num_heads_decoder=4, attn_window_size=[8, 8, 4], pos_bias_embed_dim=64,
        use_checkpoint=False, attn_drop=0.0, proj_drop=0.0,
        max_prompts=1, tokens_per_prompt_encoder=64, tokens_per_prompt_decoder=64,
        use_encoder_prompting=False, use_decoder_prompting=False,
        use_reconstruction=False, use_mutual_learning=False, use_rotation_prediction=False,
        use_contrastive_learning=False, contrastive_coding_dim=512,
        output_channels_downstream=2, output_channels_pretrain=5,
    )
    for k, v in over.items():
        setattr(conf, k, v)
    return conf


def _xavier(shape, gen):
    t = torch.empty(shape)
    fan_out, fan_in = shape[0], shape[1]
    bound = math.sqrt(6.0 / (fan_in + fan_out))
    return t.uniform_(-bound, bound, generator=gen)


def _uniform(shape, bound, gen):
    return torch.empty(shape).uniform_(-bound, bound, generator=gen)


def _block_state(sd, prefix, C, heads, window, E, tokens, use_token, gen):
    """Key names/shapes of one SwinTransformerBlock (swin_block.py:98-143,
    relative_positional_encoding.py:22-97, window_attention.py:26-33)."""
    for a, name in enumerate("hwd"):
        sd[f"{prefix}pe.enc_content_{name}"] = _xavier((2 * window[a] - 1, E), gen)
    for name in "hwd":
        sd[f"{prefix}pe.weights_content_{name}"] = _xavier((heads, E), gen)
    if use_token:
        sd[f"{prefix}pe.weights_token"] = _xavier((heads, E), gen)
    for a, name in enumerate("hwd"):
        i = torch.arange(window[a])
        sd[f"{prefix}pe.relative_dist_{name}"] = (i.view(1, -1) - i.view(-1, 1) + window[a] - 1).clamp(0, 2 * window[a] - 2)
    if use_token:
        sd[f"{prefix}pe.enc_token.0"] = _xavier((tokens, E), gen)
    k = 1.0 / math.sqrt(C)
    sd[f"{prefix}attn_norm.weight"] = torch.ones(C)
    sd[f"{prefix}attn_norm.bias"] = torch.zeros(C)
    for n in ("to_q", "to_k", "to_v"):
        sd[f"{prefix}attn.{n}.weight"] = _uniform((C, C), k, gen)
    sd[f"{prefix}attn.proj.weight"] = _uniform((C, C), k, gen)
    sd[f"{prefix}attn.proj.bias"] = _uniform((C,), k, gen)
    sd[f"{prefix}mlp_norm.weight"] = torch.ones(C)
    sd[f"{prefix}mlp_norm.bias"] = torch.zeros(C)
    sd[f"{prefix}mlp.weight"] = _uniform((C, C), k, gen)
    sd[f"{prefix}mlp.bias"] = _uniform((C,), k, gen)


def _bn_state(sd, prefix, C):
    sd[f"{prefix}weight"] = torch.ones(C)
    sd[f"{prefix}bias"] = torch.zeros(C)
    sd[f"{prefix}running_mean"] = torch.zeros(C)
    sd[f"{prefix}running_var"] = torch.ones(C)
    sd[f"{prefix}num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _conv_state(sd, prefix, cin, cout, k, gen):
    fan_in = cin * k * k * k
    b = 1.0 / math.sqrt(fan_in)
    sd[f"{prefix}weight"] = _uniform((cout, cin, k, k, k), b, gen)
    sd[f"{prefix}bias"] = _uniform((cout,), b, gen)


def _basic_block_state(sd, prefix, cin, cout, res_block, gen):
    """MONAI ``UnetrBasicBlock`` (``.layer`` = ``UnetResBlock`` | ``UnetBasicBlock``, dynunet_block.py): bias-free convs
    wrapped in ``Convolution`` (child ``conv``); InstanceNorm3d without affine has no state.  MONAI is absent: the
    structure is restated from its documented source, parity unpinned."""
    for name, ci, k in (("conv1", cin, 3), ("conv2", cout, 3)):
        sd[f"{prefix}layer.{name}.conv.weight"] = _uniform((cout, ci, k, k, k), 1 / math.sqrt(ci * k ** 3), gen)
    if res_block and cin != cout:
        sd[f"{prefix}layer.conv3.conv.weight"] = _uniform((cout, cin, 1, 1, 1), 1 / math.sqrt(cin), gen)


def basic_block_forward(x: Tensor, sd, prefix: str, res_block: bool, inner: str = "layer.") -> Tensor:
    """MONAI ``UnetResBlock.forward`` / ``UnetBasicBlock.forward`` with norm 'instance' and LeakyReLU(0.01), stride 1."""
    out = F.conv3d(x, sd[f"{prefix}{inner}conv1.conv.weight"], None, padding=1)
    out = F.leaky_relu(F.instance_norm(out, eps=1e-5), 0.01)
    out = F.instance_norm(F.conv3d(out, sd[f"{prefix}{inner}conv2.conv.weight"], None, padding=1), eps=1e-5)
    if res_block:
        residual = x
        if f"{prefix}{inner}conv3.conv.weight" in sd:
            residual = F.instance_norm(F.conv3d(x, sd[f"{prefix}{inner}conv3.conv.weight"], None), eps=1e-5)
        out = out + residual
    return F.leaky_relu(out, 0.01)


def unetr_up_block_forward(x: Tensor, skip: Tensor, sd, prefix: str, stride, res_block: bool) -> Tensor:
    """MONAI ``UnetrUpBlock.forward`` (networks/blocks/unetr_block.py): ConvTranspose3d(kernel = stride, no bias) -> cat with
    the skip -> UnetResBlock | UnetBasicBlock.  Restated from MONAI's documented source: parity unpinned."""
    up = F.conv_transpose3d(x, sd[f"{prefix}transp_conv.conv.weight"], None, stride=tuple(stride))
    return basic_block_forward(torch.cat([up, skip], dim=1), sd, prefix, res_block, inner="conv_block.")


def _up_block_state(sd, prefix, cin, cout, stride, res_block, gen):
    k = stride[0] * stride[1] * stride[2]
    sd[f"{prefix}transp_conv.conv.weight"] = _uniform((cin, cout) + tuple(stride), 1 / math.sqrt(cout * k), gen)
    for name, ci in (("conv1", 2 * cout), ("conv2", cout)):
        sd[f"{prefix}conv_block.{name}.conv.weight"] = _uniform((cout, ci, 3, 3, 3), 1 / math.sqrt(ci * 27), gen)
    if res_block:
        sd[f"{prefix}conv_block.conv3.conv.weight"] = _uniform((cout, 2 * cout, 1, 1, 1), 1 / math.sqrt(2 * cout), gen)


def random_state(conf: Namespace, seed: int = 0) -> "OrderedDict[str, Tensor]":
    """A randomly initialised state dict with the reference's key names and
    shapes for the default block options
    (``unetr_up_block == 'swin'``, ``unetr_res_block in ('none', 'simple', 'full')``).
    Init distributions follow torch defaults; exact values never matter because
    parity runs always load the same dict on both sides."""
    gen = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    hc, depth = list(conf.hidden_channels), conf.depth_unet
    win, E = list(conf.attn_window_size), conf.pos_bias_embed_dim
    mode = conf.training_mode
    has_decoder = mode != "self_supervised_learning_encoder"
    if conf.use_encoder_prompting:
        for i in range(2 * depth):
            sd[f"prompt_tokens.enc.{i}"] = _xavier((conf.tokens_per_prompt_encoder, hc[i // 2]), gen)
    if has_decoder and conf.use_decoder_prompting:
        for i in range(2 * depth):
            sd[f"prompt_tokens.dec.{i}"] = _xavier((conf.tokens_per_prompt_decoder, hc[-(i + 1) // 2 - 1]), gen)
        if conf.unetr_res_block != "none" and conf.unetr_up_block == "swin":
            for i in range(2):
                sd[f"prompt_tokens.out.{i}"] = _xavier((conf.tokens_per_prompt_decoder, hc[0]), gen)
    if mode == "downstream":
        _bn_state(sd, "extra_heads.downstream.0.", hc[0])
        _conv_state(sd, "extra_heads.downstream.1.", hc[0], conf.output_channels_downstream, 3, gen)
    if mode in ("supervised_learning_decoder", "supervised_learning_all"):
        _bn_state(sd, "extra_heads.segmentation.0.", hc[0])
        _conv_state(sd, "extra_heads.segmentation.1.", hc[0], conf.output_channels_pretrain, 3, gen)
    ps = conf.input_patch_size[0]
    fan_in = conf.input_channels * ps ** 3
    sd["input_layer.0.weight"] = _uniform((hc[0], conf.input_channels, ps, ps, ps), 1 / math.sqrt(fan_in), gen)
    sd["input_layer.0.bias"] = _uniform((hc[0],), 1 / math.sqrt(fan_in), gen)
    _bn_state(sd, "input_layer.1.", hc[0])
    for i in range(depth):
        heads = conf.num_heads_encoder * (2 ** i)
        for b in range(2):
            _block_state(sd, f"encoder_blocks.{i}.swin_blocks.{b}.", hc[i], heads, win, E,
                         conf.tokens_per_prompt_encoder, conf.use_encoder_prompting, gen)
        k = 8 if i < 1 else 4
        sd[f"encoder_blocks.{i}.merge.norm.weight"] = torch.ones(k * hc[i])
        sd[f"encoder_blocks.{i}.merge.norm.bias"] = torch.zeros(k * hc[i])
        sd[f"encoder_blocks.{i}.merge.reduction.weight"] = _uniform((hc[i + 1], k * hc[i]), 1 / math.sqrt(k * hc[i]), gen)
    if has_decoder:
        in_chs = [hc[i] for i in range(depth)][::-1]
        out_chs = [hc[i + 1] for i in range(depth)][::-1]
        if conf.unetr_res_block == "full":
            _basic_block_state(sd, "bottleneck.", hc[depth], hc[depth], conf.basic_block_res, gen)
            for i in range(depth):
                _basic_block_state(sd, f"residual_blocks.{i}.", in_chs[i], in_chs[i], conf.basic_block_res, gen)
            _basic_block_state(sd, f"residual_blocks.{depth}.", conf.input_channels, in_chs[-1], conf.basic_block_res, gen)
        else:
            _conv_state(sd, "bottleneck.", hc[depth], hc[depth], 3, gen)
        if conf.unetr_res_block == "simple":
            for i in range(depth):
                _conv_state(sd, f"residual_blocks.{i}.", in_chs[i], in_chs[i], 3, gen)
            _conv_state(sd, f"residual_blocks.{depth}.", conf.input_channels, in_chs[-1], 3, gen)
        for j in range(depth):
            cin, cout = out_chs[j], in_chs[j]
            if conf.unetr_up_block != "swin":
                _up_block_state(sd, f"decoder_blocks.{j}.", cin, cout, (2, 2, 1 if j < depth - 1 else 2), conf.res_block, gen)
                continue
            hid = cin + cin // 2
            _bn_state(sd, f"decoder_blocks.{j}.norm_concat.", hid)
            _conv_state(sd, f"decoder_blocks.{j}.conv_concat.conv.", hid, cout, 3, gen)
            for b in range(2):
                _block_state(sd, f"decoder_blocks.{j}.swin_layer.swin_blocks.{b}.", cout, conf.num_heads_decoder,
                             win, E, conf.tokens_per_prompt_decoder, conf.use_decoder_prompting, gen)
        if conf.unetr_res_block != "none" and conf.unetr_up_block != "swin":
            _up_block_state(sd, "output_layer.", in_chs[-1], in_chs[-1], (2, 2, 2), conf.res_block, gen)
        elif conf.unetr_res_block != "none":
            c = in_chs[-1]
            _bn_state(sd, "output_layer.norm_concat.", 2 * c)
            _conv_state(sd, "output_layer.conv_concat.conv.", 2 * c, c, 3, gen)
            for b in range(2):
                # the reference builds this block with use_token_params defaulting to True (swin_unetr.py:357-371)
                _block_state(sd, f"output_layer.swin_layer.swin_blocks.{b}.", c, conf.num_heads_decoder,
                             win, E, conf.tokens_per_prompt_decoder, True, gen)
    return sd


class OracleSwinUnetR:
    """Functional forward of the reference model over ``self.sd``.

    ``forward(x, training=True)`` returns ``(outputs, new_buffers)``:
    ``outputs`` has the reference's keys (swin_unetr.py:65-144); ``new_buffers``
    holds the BatchNorm running-stat updates a ``.train()`` forward would make
    (the reference's frozen BatchNorms still run in train mode,
    segmentation.py:95 + swin_unetr.py:33-40).
    """

    def __init__(self, conf: Namespace, sd: Optional[Dict[str, Tensor]] = None, seed: int = 0,
                 emulate_bf16: bool = False):
        """``emulate_bf16``: round to bf16 wherever the HIP path stores a bf16 activation (swin_ref.r16; default path
        only: ``unetr_res_block in ('none', 'simple')``).  The arithmetic stays the reference's fp32 arithmetic."""
        self.emulate_bf16 = emulate_bf16
        if conf.training_mode not in TRAINING_MODES:
            raise ValueError(f"Training mode {conf.training_mode} not available!")
        self.conf = conf
        self.sd = sd if sd is not None else random_state(conf, seed)

    # -- encoder (swin_unetr.py:46-63) --------------------------------------
    def encoder(self, x: Tensor, training: bool, nb: Dict[str, Tensor]):
        conf, sd = self.conf, self.sd
        feats = [x]
        ps = tuple(conf.input_patch_size)
        enc = F.conv3d(x, sd["input_layer.0.weight"], sd["input_layer.0.bias"], stride=ps)
        enc = S.batch_norm_train(enc, sd, "input_layer.1.", 1e-6, training, nb)
        if self.emulate_bf16:
            enc = S.r16(enc)
        feats.insert(0, enc)
        for j in range(conf.depth_unet):
            if conf.use_encoder_prompting:
                pr = (sd[f"prompt_tokens.enc.{2 * j}"], sd[f"prompt_tokens.enc.{2 * j + 1}"])
            else:
                pr = (None, None)
            enc = S.swin_pair(enc, pr, sd, f"encoder_blocks.{j}.", conf.attn_window_size,
                              conf.num_heads_encoder * (2 ** j), conf.pos_bias_embed_dim,
                              down=True, merge_last_dim=(j < 1), emulate_bf16=self.emulate_bf16)
            feats.insert(0, enc)
        return feats

    # -- decoder (swin_unetr.py:86-112) --------------------------------------
    def decoder(self, feats, training: bool, nb: Dict[str, Tensor]) -> Tensor:
        conf, sd = self.conf, self.sd
        depth = conf.depth_unet
        c0 = feats[0]
        if conf.unetr_res_block == "full":
            dec = basic_block_forward(c0, sd, "bottleneck.", conf.basic_block_res) + c0
        else:
            dec = F.conv3d(c0, sd["bottleneck.weight"], sd["bottleneck.bias"], padding=1) + c0
            if self.emulate_bf16:
                dec = S.r16(dec)
        for j in range(depth):
            if conf.use_decoder_prompting:
                pr = (sd[f"prompt_tokens.dec.{2 * j}"], sd[f"prompt_tokens.dec.{2 * j + 1}"])
            else:
                pr = (None, None)
            skip = feats[j + 1]
            if conf.unetr_res_block == "simple":
                skip = F.conv3d(skip, sd[f"residual_blocks.{j}.weight"], sd[f"residual_blocks.{j}.bias"], padding=1)
                if self.emulate_bf16:
                    skip = S.r16(skip)
            elif conf.unetr_res_block == "full":
                skip = basic_block_forward(skip, sd, f"residual_blocks.{j}.", conf.basic_block_res)
            strides = (2, 2, 1 if j < depth - 1 else 2)
            if conf.unetr_up_block != "swin":
                dec = unetr_up_block_forward(dec, skip, sd, f"decoder_blocks.{j}.", strides, conf.res_block)
                continue
            dec = S.up_block(dec, skip, pr, sd, f"decoder_blocks.{j}.", strides, conf.attn_window_size,
                             conf.num_heads_decoder, conf.pos_bias_embed_dim, training, nb, self.emulate_bf16)
        if conf.unetr_res_block == "none":
            return F.interpolate(dec, scale_factor=(2.0, 2.0, 2.0), mode="trilinear", align_corners=False)
        if conf.use_decoder_prompting:
            pr = (sd["prompt_tokens.out.0"], sd["prompt_tokens.out.1"])
        else:
            pr = (None, None)
        if conf.unetr_res_block == "full":
            skip = basic_block_forward(feats[-1], sd, f"residual_blocks.{depth}.", conf.basic_block_res)
        else:
            skip = F.conv3d(feats[-1], sd[f"residual_blocks.{depth}.weight"], sd[f"residual_blocks.{depth}.bias"], padding=1)
        if conf.unetr_up_block != "swin":
            return unetr_up_block_forward(dec, skip, sd, "output_layer.", (2, 2, 2), conf.res_block)
        return S.up_block(dec, skip, pr, sd, "output_layer.", (2, 2, 2), conf.attn_window_size,
                          conf.num_heads_decoder, conf.pos_bias_embed_dim, training, nb, self.emulate_bf16)

    def _head(self, name: str, latent: Tensor, training: bool, nb) -> Tensor:
        sd = self.sd
        y = S.batch_norm_train(latent, sd, f"extra_heads.{name}.0.", 1e-5, training, nb)
        return F.conv3d(y, sd[f"extra_heads.{name}.1.weight"], sd[f"extra_heads.{name}.1.bias"], padding=1)

    def forward(self, x: Tensor, training: bool = True) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
        conf = self.conf
        nb: Dict[str, Tensor] = {}
        feats = self.encoder(x, training, nb)
        mode = conf.training_mode
        if mode == "self_supervised_learning_encoder":
            return {"out_vit": feats}, nb
        latent = self.decoder(feats, training, nb)
        if mode == "downstream":
            return {"downstream": self._head("downstream", latent, training, nb)}, nb
        out = {"latent_outputs": latent}
        if mode in ("supervised_learning_decoder", "supervised_learning_all"):
            out["seg_pred"] = self._head("segmentation", latent, training, nb)
        return out, nb

    __call__ = forward

    # -- which tensors train in which mode (swin_unetr.py:21-44,433-527) -----
    def trainable_keys(self):
        conf, keys = self.conf, []
        mode = conf.training_mode
        for k, v in self.sd.items():
            if not v.is_floating_point() or "running_" in k:
                continue
            is_enc_prompt = k.startswith("prompt_tokens.enc") or (
                k.startswith("encoder_blocks") and ("enc_token" in k or "weights_token" in k))
            is_dec_prompt = k.startswith("prompt_tokens.dec") or k.startswith("prompt_tokens.out") or (
                (k.startswith("decoder_blocks") or k.startswith("output_layer"))
                and ("enc_token" in k or "weights_token" in k))
            is_encoder = k.startswith("input_layer") or k.startswith("encoder_blocks")
            if mode == "downstream":
                if is_enc_prompt or is_dec_prompt or k.startswith("extra_heads.downstream"):
                    keys.append(k)
            elif mode in ("self_supervised_learning_decoder", "supervised_learning_decoder"):
                if not is_encoder and not is_enc_prompt:
                    keys.append(k)
            else:
                keys.append(k)
        return keys
