"""Drop-in ``SwinUnetR(conf)`` whose arithmetic runs in the HIP kernels of libmivp_hip.so.

Surface kept from the reference (swin_unetr/swin_unetr.py:8-527): constructor taking the config
Namespace (positionally or as ``conf=``), attributes ``input_layer / encoder_blocks / bottleneck /
residual_blocks / decoder_blocks / output_layer / prompt_tokens / extra_heads / conf``, ``forward(x)
-> dict`` with the reference's keys, the five ``named_parameters_*`` helpers, and a ``state_dict``
with the reference's names, shapes, dtypes and ordering (checked against the
reference's own key list in tests/test_module_surface.py).

How it is built: the sub-modules are stock ``torch.nn`` layers used purely as *parameter holders*
(same registration order as the reference => same keys and same default initialisation); none of
their ``forward`` methods is ever called.  ``forward`` below walks channels-last bf16 activations
through autograd Functions (functional.py) that launch the kernels.  There is no CPU path: a CPU
tensor raises.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import functional as Fn

TRAINING_MODES = (
    "self_supervised_learning_encoder",
    "self_supervised_learning_decoder",
    "self_supervised_learning_all",
    "supervised_learning_decoder",
    "supervised_learning_all",
    "downstream",
)


def _xavier(*shape):
    return nn.Parameter(nn.init.xavier_uniform_(torch.empty(shape), gain=nn.init.calculate_gain("linear")))


class RelativePE(nn.Module):
    """Parameter holder of the separable relative-position bias
    (multi_head_attention/relative_positional_encoding.py:7-154)."""

    def __init__(self, embed_dim, num_heads, max_abs_pos, max_cap_dist, max_prompts, tokens_per_prompt,
                 use_token_params=True):
        super().__init__()
        self.scale = embed_dim ** -0.5
        self.num_heads = num_heads
        self.enc_content_h = _xavier(2 * max_cap_dist[0] - 1, embed_dim)
        self.enc_content_w = _xavier(2 * max_cap_dist[1] - 1, embed_dim)
        self.enc_content_d = _xavier(2 * max_cap_dist[2] - 1, embed_dim)
        for axis, name in enumerate("hwd"):
            i = torch.arange(max_abs_pos[axis], dtype=torch.long)
            dist = (i.view(1, -1) - i.view(-1, 1) + max_cap_dist[axis] - 1).clamp(0, 2 * (max_cap_dist[axis] - 1))
            self.register_buffer(f"relative_dist_{name}", dist)
        self.weights_content_h = _xavier(num_heads, embed_dim)
        self.weights_content_w = _xavier(num_heads, embed_dim)
        self.weights_content_d = _xavier(num_heads, embed_dim)
        self.use_token_params = use_token_params
        if use_token_params:
            self.enc_token = nn.ParameterList([_xavier(tokens_per_prompt, embed_dim) for _ in range(max_prompts)])
            self.weights_token = _xavier(num_heads, embed_dim)

    def content_tables(self):
        """Three per-axis tables ``[heads, 2w-1]`` (already times scale/3): entry j-i+w-1 is the bias
        contribution of a key at slot coordinate j to a query at slot coordinate i."""
        s = self.scale / 3.0
        return tuple(((getattr(self, f"weights_content_{a}") @ getattr(self, f"enc_content_{a}").t()) * s).float().contiguous()
                     for a in "hwd")

    def token_scores(self, n_prompt):
        """``[heads, n_prompt]`` bias of the prompt-token key columns (times scale)."""
        emb = torch.cat(list(self.enc_token))[:n_prompt]
        return ((self.weights_token @ emb.t()) * self.scale).float().contiguous()

    def named_parameters_bias_content(self):
        return [(n, p) for n, p in self.named_parameters() if "enc_content" in n or "weights_content" in n]

    def named_parameters_bias_prompt_tokens(self):
        return [(n, p) for n, p in self.named_parameters() if "enc_token" in n or "weights_token" in n]


class WindowAttention(nn.Module):
    """Parameter holder (multi_head_attention/window_attention.py:11-33)."""

    def __init__(self, dim, num_heads, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        if dim % num_heads != 0:
            raise ValueError("WindowAttention: The dimension is not compatible with the number of heads!")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(dim, dim, bias=False)
        self.to_v = nn.Linear(dim, dim, bias=False)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class SwinTransformerBlock(nn.Module):
    """One (shifted-)window attention block (swin_transformer/swin_block.py:98-289)."""

    def __init__(self, hidden_channels, window_size, pos_bias_embed_dim, num_heads, max_prompts, tokens_per_prompt,
                 use_token_params=True, shift_size=None, attn_drop=0.0, proj_drop=0.0, use_checkpoint=False):
        super().__init__()
        self.num_heads = num_heads
        self.window_size = tuple(int(v) for v in window_size)
        self.shift_size = tuple(int(v) for v in (shift_size or (0, 0, 0)))
        self.use_checkpoint = use_checkpoint      # no-op: the fused kernels save O(tokens) state only
        self.embed_dim = pos_bias_embed_dim
        self.pe = RelativePE(pos_bias_embed_dim, num_heads, window_size, window_size, max_prompts, tokens_per_prompt,
                             use_token_params)
        self.attn_norm = nn.LayerNorm(hidden_channels, eps=1e-6)
        self.attn = WindowAttention(hidden_channels, num_heads, attn_drop, proj_drop)
        self.mlp_norm = nn.LayerNorm(hidden_channels, eps=1e-6)
        self.mlp = nn.Linear(hidden_channels, hidden_channels)
        self._wcache = Fn.WeightCache()

    def forward(self, x, p=None):
        """x: bf16 channels-last [B,H,W,D,C]; p: [Np, C] prompt parameter (or the reference's
        batch-broadcast [B,Np,C], of which row 0 is used -- all rows are identical)."""
        if p is not None and p.dim() == 3:
            p = p[0]
        return Fn.swin_block(self, x, p)

    def named_parameters_body(self):
        out = []
        for m in (self.attn_norm, self.attn, self.mlp_norm, self.mlp):
            out.extend(m.named_parameters())
        return out

    def named_parameters_bias_content(self):
        return list(self.pe.named_parameters_bias_content())

    def named_parameters_bias_prompt_tokens(self):
        return list(self.pe.named_parameters_bias_prompt_tokens())


class PatchMerging(nn.Module):
    """2x2x2 / 2x2x1 merge (swin_transformer/down.py:6-60)."""

    def __init__(self, in_channels, out_channels, merge_last_dim=True):
        super().__init__()
        k = 8 if merge_last_dim else 4
        self.norm = nn.LayerNorm(k * in_channels, eps=1e-6)
        self.reduction = nn.Linear(k * in_channels, out_channels, bias=False)
        self.merge_last_dim = merge_last_dim
        self._wcache = Fn.WeightCache()

    def forward(self, x):
        return Fn.patch_merge(self, x)

    def named_parameters_body(self):
        return [*self.reduction.named_parameters(), *self.norm.named_parameters()]


class ConsecutiveSwinBlocks(nn.Module):
    """W-MSA block, SW-MSA block, optional merge (swin_transformer/swin_block.py:16-95)."""

    def __init__(self, hidden_channels, num_heads, pos_bias_embed_dim, max_prompts, tokens_per_prompt, window_size,
                 use_token_params=True, shift_size=None, down=True, merge_last_dim=True, use_checkpoint=False,
                 out_channels=None, proj_drop=0.0, attn_drop=0.0):
        super().__init__()
        self.window_size = tuple(int(v) for v in window_size)
        self.shift_size = tuple(shift_size) if shift_size is not None else tuple(v // 2 for v in self.window_size)
        self.down = down
        self.swin_blocks = nn.ModuleList([
            SwinTransformerBlock(hidden_channels, self.window_size, pos_bias_embed_dim, num_heads, max_prompts,
                                 tokens_per_prompt, use_token_params, (0, 0, 0) if i == 0 else self.shift_size,
                                 attn_drop, proj_drop, use_checkpoint)
            for i in range(2)])
        if down:
            self.merge = PatchMerging(hidden_channels, out_channels or 2 * hidden_channels, merge_last_dim)

    def forward(self, x, p=(None, None)):
        for blk, prm in zip(self.swin_blocks, p):
            x = blk(x, prm)
        return self.merge(x) if self.down else x

    def named_parameters_body(self):
        out = []
        for blk in self.swin_blocks:
            out.extend(blk.named_parameters_body())
        if self.down:
            out.extend(self.merge.named_parameters())
        return out

    def named_parameters_bias_content(self):
        return [q for blk in self.swin_blocks for q in blk.named_parameters_bias_content()]

    def named_parameters_bias_prompt_tokens(self):
        return [q for blk in self.swin_blocks for q in blk.named_parameters_bias_prompt_tokens()]


class _ConvHolder(nn.Sequential):
    """State-dict twin of MONAI ``Convolution(..., conv_only=True)``: one child named ``conv``."""

    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("conv", nn.Conv3d(cin, cout, 3, 1, padding=1, bias=True))


class _BareConv(nn.Sequential):
    """State-dict twin of MONAI ``get_conv_layer(..., conv_only=False, act=None, norm=None)``: child ``conv``, no bias."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.add_module("conv", nn.Conv3d(cin, cout, k, 1, padding=k // 2, bias=False))


class _UnetBlockLayer(nn.Module):
    """MONAI ``UnetResBlock`` / ``UnetBasicBlock`` (networks/blocks/dynunet_block.py) at stride 1 with norm 'instance':
    the attribute names are MONAI's, the arithmetic runs in ``functional.unetr_basic_block``."""

    def __init__(self, cin, cout, res_block):
        super().__init__()
        self.conv1 = _BareConv(cin, cout, 3)
        self.conv2 = _BareConv(cout, cout, 3)
        self.lrelu = nn.LeakyReLU(0.01, inplace=True)
        self.norm1 = nn.InstanceNorm3d(cout)
        self.norm2 = nn.InstanceNorm3d(cout)
        if res_block:
            self.downsample = cin != cout
            if self.downsample:
                self.conv3 = _BareConv(cin, cout, 1)
                self.norm3 = nn.InstanceNorm3d(cout)


class UnetrBasicBlock(nn.Module):
    """Stand-in for MONAI ``UnetrBasicBlock(spatial_dims=3, kernel_size=3, stride=1, norm_name='instance', res_block=...)``
    (swin_unetr.py:248-289 with ``unetr_res_block: 'full'``).  MONAI is not importable here: structure and
    state-dict names are restated from its documented source, parity unpinned at that boundary."""

    def __init__(self, in_channels, out_channels, res_block):
        super().__init__()
        self.res_block = bool(res_block)
        self.layer = _UnetBlockLayer(in_channels, out_channels, self.res_block)


class _TransposedConv(nn.Sequential):
    """State-dict twin of MONAI ``get_conv_layer(..., conv_only=True, is_transposed=True)``: child ``conv``, no bias."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.add_module("conv", nn.ConvTranspose3d(cin, cout, kernel_size=stride, stride=stride, bias=False))


class UnetrUpBlock(nn.Module):
    """Stand-in for MONAI ``UnetrUpBlock(spatial_dims=3, in, out, kernel_size=3, upsample_kernel_size, norm_name='instance',
    res_block)`` (swin_unetr.py:338-348,372-380 with ``unetr_up_block != 'swin'``): ``transp_conv`` (kernel = stride, no
    bias) -> cat with the skip -> ``conv_block`` (UnetResBlock | UnetBasicBlock on 2 x out channels).  MONAI is not
    importable here: structure and state-dict names restated from its documented source, parity unpinned.  The reference
    cannot run this option (it calls the block with three arguments and up-samples D where the encoder never merged it,
    SURVEY 8 a16); here the up-sampling stride follows the encoder's merges so that the shapes close, and prompts -- which a
    convolutional block cannot take -- are ignored."""

    def __init__(self, in_channels, out_channels, upsample_stride, res_block):
        super().__init__()
        self.res_block = bool(res_block)
        self.transp_conv = _TransposedConv(in_channels, out_channels, tuple(int(s) for s in upsample_stride))
        self.conv_block = _UnetBlockLayer(2 * out_channels, out_channels, self.res_block)
        self._wcache = Fn.WeightCache()

    def forward(self, x, skip, p=None):
        return Fn.unetr_up_block(self, "up", self, x, skip)

    def named_parameters_body(self):
        return list(self.named_parameters())

    def named_parameters_bias_content(self):
        return []

    def named_parameters_bias_prompt_tokens(self):
        return []


class SwinUpBlock(nn.Module):
    """Decoder stage (swin_unetr/unet_blocks.py:11-92): upsample, crop+concat, BN, LeakyReLU, conv 3^3,
    two Swin blocks."""

    def __init__(self, in_channels, out_channels, strides, kernel_size, pos_bias_embed_dim, num_heads, window_size,
                 max_prompts, tokens_per_prompt, use_token_params=True, act="leakyrelu", norm="batch", attn_drop=0.0,
                 proj_drop=0.0, use_checkpoint=False, hidden_channels=None):
        super().__init__()
        self.strides = tuple(int(s) for s in strides)
        self.up = nn.Upsample(scale_factor=tuple(float(s) for s in strides), mode="trilinear", align_corners=False)
        self.act = nn.LeakyReLU()
        hidden = hidden_channels if hidden_channels is not None else in_channels + in_channels // 2
        self.norm_concat = nn.BatchNorm3d(hidden)
        self.conv_concat = _ConvHolder(hidden, out_channels)
        self.swin_layer = ConsecutiveSwinBlocks(out_channels, num_heads, pos_bias_embed_dim, max_prompts,
                                                tokens_per_prompt, window_size, use_token_params, down=False,
                                                attn_drop=attn_drop, proj_drop=proj_drop, use_checkpoint=use_checkpoint)
        self._wcache = Fn.WeightCache()

    def forward(self, x, c, p=(None, None)):
        y = Fn.upcat_bn_act_conv(self, self.norm_concat, self.conv_concat.conv, x, c, self.strides, lrelu=True)
        return self.swin_layer(y, p)

    def named_parameters_body(self):
        return [*self.norm_concat.named_parameters(), *self.conv_concat.named_parameters(),
                *self.swin_layer.named_parameters_body()]

    def named_parameters_bias_content(self):
        return self.swin_layer.named_parameters_bias_content()

    def named_parameters_bias_prompt_tokens(self):
        return self.swin_layer.named_parameters_bias_prompt_tokens()


class SwinUnetR(nn.Module):
    def __init__(self, conf):
        super().__init__()
        self.input_layer = None
        self.encoder_blocks = None
        self.bottleneck = None
        self.residual_blocks = None
        self.decoder_blocks = None
        self.output_layer = None
        self.prompt_tokens = nn.ModuleDict()
        self.extra_heads = nn.ModuleDict()
        self.conf = conf
        self._wcache = Fn.WeightCache()
        mode = conf.training_mode
        if mode not in TRAINING_MODES:
            raise ValueError(f"Training mode {mode} not available!")
        if mode == "self_supervised_learning_encoder":
            self._build_encoder()
            self._build_encoder_heads()
            if conf.use_encoder_prompting:
                self._build_prompts_enc()
        else:
            self._build_decoder()
            if mode == "downstream":
                self.extra_heads["downstream"] = nn.Sequential(
                    nn.BatchNorm3d(conf.hidden_channels[0]),
                    nn.Conv3d(conf.hidden_channels[0], conf.output_channels_downstream, 3, 1, padding=1))
        # what is frozen in which mode (swin_unetr.py:21-44)
        if mode in ("self_supervised_learning_decoder", "supervised_learning_decoder"):
            for _, q in self.named_parameters_encoder(include_prompt_tokens=conf.use_encoder_prompting):
                q.requires_grad = False
        elif mode == "downstream":
            for _, q in self.named_parameters_encoder(include_prompt_tokens=False):
                q.requires_grad = False
            for _, q in self.named_parameters_decoder(include_prompt_tokens=False):
                q.requires_grad = False

    # ------------------------------------------------------------------ construction
    def _build_encoder(self):
        c = self.conf
        hc = list(c.hidden_channels)
        self.input_layer = nn.Sequential(
            nn.Conv3d(c.input_channels, hc[0], kernel_size=tuple(c.input_patch_size), stride=tuple(c.input_patch_size)),
            nn.BatchNorm3d(hc[0], eps=1e-6))
        self.encoder_blocks = nn.ModuleList([
            ConsecutiveSwinBlocks(hc[i], c.num_heads_encoder * (2 ** i), c.pos_bias_embed_dim, c.max_prompts,
                                  c.tokens_per_prompt_encoder, c.attn_window_size,
                                  use_token_params=c.use_encoder_prompting, down=True, merge_last_dim=(i < 1),
                                  use_checkpoint=c.use_checkpoint, proj_drop=c.proj_drop, attn_drop=c.attn_drop)
            for i in range(c.depth_unet)])

    def _build_encoder_heads(self):
        c = self.conf
        hc = list(c.hidden_channels)
        depth = c.depth_unet
        if c.use_reconstruction or c.use_mutual_learning:
            chs = [hc[-1] // (2 ** i) for i in range(depth + 1)] + [hc[-1] // (2 ** depth)]
            layers: List[nn.Module] = []
            for i in range(depth + 1):
                layers += [nn.Conv3d(chs[i], chs[i + 1], 3, 1, 1), nn.InstanceNorm3d(chs[i + 1]), nn.LeakyReLU(),
                           nn.Upsample(scale_factor=(2, 2, 1 if i < depth - 1 else 2), mode="trilinear", align_corners=True)]
            layers.append(nn.Conv3d(chs[-1], c.input_channels, 1, 1))
            self.extra_heads["reconstruction"] = nn.Sequential(*layers)
        if c.use_rotation_prediction:
            self.extra_heads["rotation_prediction"] = nn.Linear(hc[-1], 4)
        if c.use_contrastive_learning:
            self.extra_heads["contrastive_coding"] = nn.Linear(hc[-1], c.contrastive_coding_dim)

    def _build_decoder(self):
        c = self.conf
        hc = list(c.hidden_channels)
        depth = c.depth_unet
        swin_up = c.unetr_up_block == "swin"
        self._build_encoder()
        dec_in = [hc[i + 1] for i in range(depth)][::-1]       # channels entering each decoder stage
        dec_out = [hc[i] for i in range(depth)][::-1]
        if c.unetr_res_block == "full":
            # conf.basic_block_res is read unguarded, as in the reference (swin_unetr.py:256,277,288): it is not a yml key
            self.bottleneck = UnetrBasicBlock(dec_in[0], dec_in[0], c.basic_block_res)
            self.residual_blocks = nn.ModuleList(
                [UnetrBasicBlock(dec_out[i], dec_out[i], c.basic_block_res) for i in range(depth)]
                + [UnetrBasicBlock(c.input_channels, dec_out[-1], c.basic_block_res)])
        else:
            self.bottleneck = nn.Conv3d(dec_in[0], dec_in[0], 3, 1, padding=1)
        if c.unetr_res_block == "full":
            pass
        elif c.unetr_res_block == "simple":
            self.residual_blocks = nn.ModuleList(
                [nn.Conv3d(dec_out[i], dec_out[i], 3, 1, padding=1) for i in range(depth)]
                + [nn.Conv3d(c.input_channels, dec_out[-1], 3, 1, padding=1)])
        else:
            self.residual_blocks = nn.ModuleList([nn.Identity() for _ in range(depth + 1)])
        if swin_up:
            self.decoder_blocks = nn.ModuleList([
                SwinUpBlock(dec_in[i], dec_out[i], (2, 2, 1 if i < depth - 1 else 2), (3, 3, 3), c.pos_bias_embed_dim,
                            c.num_heads_decoder, c.attn_window_size, c.max_prompts, c.tokens_per_prompt_decoder,
                            use_token_params=c.use_decoder_prompting, attn_drop=c.attn_drop, proj_drop=c.proj_drop,
                            use_checkpoint=c.use_checkpoint)
                for i in range(depth)])
        else:
            # CNN decoder (swin_unetr.py:338-348): MONAI UnetrUpBlock -- parity unpinned (restated from MONAI's documented
            # source).  ``conf.res_block`` is not a yml key (the reference reads it unguarded and fails with
            # an AttributeError): default False with a clear message instead.  The reference passes the decoder prompt as a
            # third argument the MONAI block does not take: prompts cannot act on this decoder.
            if not hasattr(c, "res_block"):
                import warnings
                warnings.warn("conf.res_block is not set (the reference's yml has no such key): UnetrUpBlock uses res_block=False")
            if getattr(c, "use_decoder_prompting", False):
                raise ValueError("use_decoder_prompting needs unetr_up_block='swin': the CNN decoder (UnetrUpBlock) takes no "
                                 "prompt tokens (the reference's call fails there too)")
            res_block = bool(getattr(c, "res_block", False))
            self.decoder_blocks = nn.ModuleList([
                UnetrUpBlock(dec_in[i], dec_out[i], (2, 2, 1 if i < depth - 1 else 2), res_block) for i in range(depth)])
        if c.unetr_res_block == "none":
            self.output_layer = nn.Upsample(scale_factor=(2, 2, 2), mode="trilinear", align_corners=False)
        elif not swin_up:
            self.output_layer = UnetrUpBlock(dec_out[-1], dec_out[-1], (2, 2, 2), bool(getattr(c, "res_block", False)))
        else:
            self.output_layer = SwinUpBlock(dec_out[-1], dec_out[-1], (2, 2, 2), (3, 3, 3), c.pos_bias_embed_dim,
                                            c.num_heads_decoder, c.attn_window_size, c.max_prompts,
                                            c.tokens_per_prompt_decoder, attn_drop=c.attn_drop, proj_drop=c.proj_drop,
                                            use_checkpoint=c.use_checkpoint, hidden_channels=2 * dec_out[-1])
        if c.training_mode in ("supervised_learning_decoder", "supervised_learning_all"):
            self.extra_heads["segmentation"] = nn.Sequential(
                nn.BatchNorm3d(hc[0]), nn.Conv3d(hc[0], c.output_channels_pretrain, 3, 1, padding=1))
        if c.use_encoder_prompting:
            self._build_prompts_enc()
        if c.use_decoder_prompting:
            self._build_prompts_dec()

    def _build_prompts_enc(self):
        c = self.conf
        self.prompt_tokens["enc"] = nn.ParameterList(
            [_xavier(c.tokens_per_prompt_encoder, c.hidden_channels[i // 2]) for i in range(2 * c.depth_unet)])

    def _build_prompts_dec(self):
        c = self.conf
        hc = c.hidden_channels
        self.prompt_tokens["dec"] = nn.ParameterList(
            [_xavier(c.tokens_per_prompt_decoder, hc[-(i + 1) // 2 - 1]) for i in range(2 * c.depth_unet)])
        if c.unetr_res_block != "none" and c.unetr_up_block == "swin":
            self.prompt_tokens["out"] = nn.ParameterList(
                [_xavier(c.tokens_per_prompt_decoder, hc[0]) for _ in range(2)])

    # ------------------------------------------------------------------ forward
    def _prompts(self, side, j):
        if side == "enc" and not self.conf.use_encoder_prompting:
            return (None, None)
        if side in ("dec", "out") and not self.conf.use_decoder_prompting:
            return (None, None)
        if side == "out" and side not in self.prompt_tokens:      # CNN up blocks take no prompts (no 'out' tokens are built)
            return (None, None)
        lst = self.prompt_tokens[side]
        return (lst[2 * j], lst[2 * j + 1])

    def forward_swin_transformer(self, x):
        """Returns channels-last bf16 features, deepest first, input volume last (as the reference's
        ``out_vit`` list, swin_unetr.py:46-63)."""
        feats = [x]
        if self.conf.use_encoder_prompting:          # prompt-side operands of every encoder block in three launches
            Fn.prepare_prompted_blocks([(blk, prm) for j in range(self.conf.depth_unet)
                                        for blk, prm in zip(self.encoder_blocks[j].swin_blocks, self._prompts("enc", j))])
        enc = Fn.patch_embed(self, self.input_layer[0], self.input_layer[1], x)
        feats.insert(0, enc)
        for j in range(self.conf.depth_unet):
            enc = self.encoder_blocks[j](enc, self._prompts("enc", j))
            feats.insert(0, enc)
        return feats

    def forward_decoder(self, feats, upsample_output=True):
        """``upsample_output=False`` (only with ``unetr_res_block == 'none'``) returns the last decoder stage's
        low-resolution output, i.e. skips ``output_layer`` (the x2 trilinear upsample)."""
        c = self.conf
        depth = c.depth_unet
        if c.use_decoder_prompting:                  # ... and of every decoder block
            pairs = []
            for j in range(depth):
                layer = getattr(self.decoder_blocks[j], "swin_layer", None)
                if layer is not None:
                    pairs += list(zip(layer.swin_blocks, self._prompts("dec", j)))
            Fn.prepare_prompted_blocks(pairs)
        if c.unetr_res_block == "full":
            dec = Fn.unetr_basic_block(self, "bottleneck", self.bottleneck, feats[0]) + feats[0]
        else:
            dec = Fn.conv3d_plain(self, "bottleneck", self.bottleneck, feats[0], residual=feats[0])
        for j in range(depth):
            skip = feats[j + 1]
            if c.unetr_res_block == "simple":
                skip = Fn.conv3d_plain(self, f"res{j}", self.residual_blocks[j], skip)
            elif c.unetr_res_block == "full":
                skip = Fn.unetr_basic_block(self, f"res{j}", self.residual_blocks[j], skip)
            dec = self.decoder_blocks[j](dec, skip, self._prompts("dec", j))
        if c.unetr_res_block == "none":
            return Fn.upcat(dec, None, (2, 2, 2)) if upsample_output else dec
        if c.unetr_res_block == "full":
            skip = Fn.unetr_basic_block(self, f"res{depth}", self.residual_blocks[depth], Fn.to_channels_last(feats[-1]))
        else:
            skip = Fn.conv3d_plain(self, f"res{depth}", self.residual_blocks[depth], Fn.to_channels_last(feats[-1]))
        return self.output_layer(dec, skip, self._prompts("out", 0))

    def forward(self, x):
        """x: float [B, Cin, H, W, D] on the GPU -> dict as the reference (swin_unetr.py:129-144);
        volumes in the dict are channels-first *views* of channels-last storage."""
        try:
            return self._forward(x)
        finally:
            Fn.flush_counters()                      # the BatchNorm step counters of this forward, one launch

    def _forward(self, x):
        Fn.require_device(x)
        mode = self.conf.training_mode
        feats = self.forward_swin_transformer(x)
        if mode == "self_supervised_learning_encoder":
            return self._encoder_outputs(feats)
        if mode == "downstream" and self.conf.unetr_res_block == "none":
            head = self.extra_heads["downstream"]
            dec = self.forward_decoder(feats, upsample_output=False)
            if Fn.uphead_applicable(dec, head[0], head[1]):
                # upsample -> BatchNorm -> conv evaluated from the low-resolution tensor (csrc/uphead.hip)
                return {"downstream": Fn.to_channels_first(Fn.uphead(head[0], head[1], dec))}
            latent = Fn.upcat(dec, None, (2, 2, 2))
        else:
            latent = self.forward_decoder(feats)
        if mode == "downstream":
            seg = Fn.bn_act_conv(self, self.extra_heads["downstream"][0], self.extra_heads["downstream"][1], latent,
                                 lrelu=False, out_f32=True, key="head_downstream")
            return {"downstream": Fn.to_channels_first(seg)}
        out = {"latent_outputs": Fn.to_channels_first(latent)}
        if mode in ("supervised_learning_decoder", "supervised_learning_all"):
            seg = Fn.bn_act_conv(self, self.extra_heads["segmentation"][0], self.extra_heads["segmentation"][1], latent,
                                 lrelu=False, out_f32=True, key="head_segmentation")
            out["seg_pred"] = Fn.to_channels_first(seg)
        return out

    def _encoder_outputs(self, feats):
        """forward_ssl_encoder (swin_unetr.py:64-83): the feature list plus the configured proxy-task heads on the deepest
        feature."""
        out = {}
        deepest = feats[0]
        if "reconstruction" in self.extra_heads:
            rec = Fn.reconstruction_head(self, self.extra_heads["reconstruction"], deepest)
            out["reconstruction"] = Fn.to_channels_first(rec)
        if "rotation_prediction" in self.extra_heads:
            out["rotation_prediction"] = Fn.pooled_linear(self.extra_heads["rotation_prediction"], deepest)
        if "contrastive_coding" in self.extra_heads:
            out["contrastive_coding"] = Fn.pooled_linear(self.extra_heads["contrastive_coding"], deepest)
        out["out_vit"] = [Fn.to_channels_first(f) if f.dim() == 5 and f.dtype == torch.bfloat16 else f for f in feats]
        return out

    # ------------------------------------------------------------------ parameter groups (swin_unetr.py:433-527)
    def named_parameters_downstream(self):
        out = []
        if self.conf.use_encoder_prompting:
            out.extend(self.named_parameters_prompt_tokens_encoder())
        if self.conf.use_decoder_prompting:
            out.extend(self.named_parameters_prompt_tokens_decoder())
        out.extend(self.extra_heads["downstream"].named_parameters())
        return out

    def named_parameters_prompt_tokens_encoder(self):
        out = list(self.prompt_tokens["enc"].named_parameters())
        for blk in self.encoder_blocks:
            out.extend(blk.named_parameters_bias_prompt_tokens())
        return out

    def named_parameters_prompt_tokens_decoder(self):
        tokens = list(self.prompt_tokens["dec"].named_parameters())
        bias = []
        for blk in self.decoder_blocks:
            bias.extend(blk.named_parameters_bias_prompt_tokens())
        if self.conf.unetr_res_block != "none" and self.conf.unetr_up_block == "swin":
            tokens.extend(self.prompt_tokens["out"].named_parameters())
        if self.conf.unetr_res_block != "none":
            bias.extend(self.output_layer.named_parameters_bias_prompt_tokens())
        return tokens + bias

    def named_parameters_encoder(self, include_prompt_tokens=False):
        out = list(self.input_layer.named_parameters())
        for blk in self.encoder_blocks:
            out.extend(blk.named_parameters_body())
            out.extend(blk.named_parameters_bias_content())
        if include_prompt_tokens and self.conf.use_encoder_prompting:
            out.extend(self.named_parameters_prompt_tokens_encoder())
        if self.conf.training_mode == "self_supervised_learning_encoder":
            for head in self.extra_heads.values():
                out.extend(head.named_parameters())
        return out

    def named_parameters_decoder(self, include_prompt_tokens=False):
        out = list(self.bottleneck.named_parameters())
        for blk in self.residual_blocks:
            out.extend(blk.named_parameters())
        for blk in self.decoder_blocks:
            out.extend(blk.named_parameters_body())
            out.extend(blk.named_parameters_bias_content())
        if self.conf.unetr_res_block != "none":
            out.extend(self.output_layer.named_parameters_body())
            out.extend(self.output_layer.named_parameters_bias_content())
        if include_prompt_tokens and self.conf.use_decoder_prompting:
            out.extend(self.named_parameters_prompt_tokens_decoder())
        if self.conf.training_mode in ("supervised_learning_decoder", "supervised_learning_all"):
            out.extend(self.extra_heads["segmentation"].named_parameters())
        return out
