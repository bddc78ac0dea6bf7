"""Oracle (test infrastructure): Swin block family restated in gather-index form.

Plain PyTorch fp32 on CPU.  Tensors are channels-first ``[B, C, H, W, D]`` at
the API like the reference; internally everything is an explicit index map, not
the reference's einops chain.  Parameters are looked up in a flat ``sd`` dict
that uses the reference's ``state_dict`` key names, so a
golden fixture / a product module / a reference checkpoint all plug in directly.

Reference lines each function follows are cited in its docstring
(paths relative to ``/root/reference/src/modules``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

LOG2E = 1.4426950408889634


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def r16(t: Tensor) -> Tensor:
    """Round to bf16 and back, with a straight-through gradient.  Used only by the ``emulate_bf16`` options below:
    the fp32 arithmetic of the reference with a rounding at every point where the HIP path STORES a bf16 value
    (DESIGN.md section 2), so that a parity test measures kernel arithmetic instead of the storage format."""
    return _RoundBf16.apply(t)


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------
class BlockGeometry:
    """Padding / shift / strided-window index maps of one Swin block call.

    Follows swin_transformer/swin_block.py:145-178,247-253,265-270,292-309.
    * effective shift per axis is 0 when ``dim <= window`` (:265-270)
    * if any axis is not divisible, EVERY axis is padded by ``w - dim % w``
      (a full window on divisible axes) (:151-161)
    * ``F.pad(x, reversed(paddings))`` puts ceil(t/2) zeros in FRONT and
      floor(t/2) behind (:163) while the crop uses [floor : L - ceil] (:247-253)
    * "window" p holds coordinates {p, p+n, p+2n, ...} of the rolled frame
      (:292-299: outer factor = in-window index, inner factor = window id).
    """

    def __init__(self, dims: Sequence[int], window: Sequence[int], shift_cfg: Sequence[int]):
        self.dims = tuple(int(d) for d in dims)
        self.window = tuple(int(w) for w in window)
        self.shift = tuple(int(s) if d > w else 0 for s, d, w in zip(shift_cfg, self.dims, self.window))
        need = any(d % w != 0 for d, w in zip(self.dims, self.window))
        self.total = tuple((w - d % w) if need else 0 for d, w in zip(self.dims, self.window))
        self.lo = tuple(t // 2 for t in self.total)            # floor
        self.hi = tuple(t - t // 2 for t in self.total)        # ceil
        self.padded = tuple(d + t for d, t in zip(self.dims, self.total))
        self.nwin = tuple(L // w for L, w in zip(self.padded, self.window))
        self.P = self.nwin[0] * self.nwin[1] * self.nwin[2]
        self.N = self.window[0] * self.window[1] * self.window[2]
        self.has_shift = any(s > 0 for s in self.shift)
        self.has_pad = any(t > 0 for t in self.total)

    def rolled_coord(self, axis: int) -> Tensor:
        """[n_a, w_a] rolled-frame coordinate j = i*n + p of (window p, slot i)."""
        n, w = self.nwin[axis], self.window[axis]
        p = torch.arange(n).view(n, 1)
        i = torch.arange(w).view(1, w)
        return i * n + p

    def padded_coord(self, axis: int) -> Tensor:
        """[n_a, w_a] coordinate in the (un-rolled) padded frame."""
        return (self.rolled_coord(axis) + self.shift[axis]) % self.padded[axis]

    def flat_index(self, per_axis) -> Tensor:
        """Combine three [n_a, w_a] coordinate maps into a [P, N] linear index
        into the padded frame (row-major over the three padded axes)."""
        c0, c1, c2 = per_axis
        n0, n1, n2 = self.nwin
        w0, w1, w2 = self.window
        L1, L2 = self.padded[1], self.padded[2]
        lin = (c0.view(n0, 1, 1, w0, 1, 1) * L1 + c1.view(1, n1, 1, 1, w1, 1)) * L2 \
            + c2.view(1, 1, n2, 1, 1, w2)
        return lin.reshape(self.P, self.N)

    def token_index(self) -> Tensor:
        return self.flat_index([self.padded_coord(a) for a in range(3)])


# ----------------------------------------------------------------------------
# relative position bias  (a9)
# ----------------------------------------------------------------------------
def rel_pos_bias(sd: Dict[str, Tensor], prefix: str, window: Sequence[int], n_prompt: int,
                 embed_dim: int, emulate_bf16: bool = False) -> Tensor:
    """Separable learned relative-position bias -> ``[heads, N, N + n_prompt]``.

    Follows multi_head_attention/relative_positional_encoding.py:99-142.
    ``bias[h,n,m] = s * (R_h[i0n,i0m] + R_w[i1n,i1m] + R_d[i2n,i2m]) / 3`` with
    ``R_a[h,i,j] = sum_c W_a[h,c] * E_a[clamp(j-i+w_a-1), c]`` and ``s = embed_dim**-0.5``
    (:20, :101-126).  Prompt columns get ``s * W_tok[h] . E_tok[t]``, constant over
    the query row (:134-141); this function returns only content-query rows
    (the reference's prompt-query rows are all zero and are discarded by the
    block, swin_block.py:223-225).

    Restated as a Toeplitz gather from per-axis ``[heads, 2w-1]`` tables instead
    of the reference's embedding gather + einsum.

    ``emulate_bf16``: the HIP path carries the bias as extra bf16 columns of K' in log2 units (one column per query
    slot coordinate: ``Th + c | Tw | Td - c`` with ``c = Td`` at query coordinate ``w2-1``, whose own column is dropped);
    each of the three summands is rounded to bf16 after the multiplication by log2(e), as is the prompt-token score.
    """
    scale = embed_dim ** -0.5
    w = [int(v) for v in window]
    per_axis = []
    for a, name in enumerate("hwd"):
        emb = sd[f"{prefix}pe.enc_content_{name}"]          # [2w-1, E]
        wts = sd[f"{prefix}pe.weights_content_{name}"]       # [heads, E]
        table = wts @ emb.t()                                 # [heads, 2w-1]
        i = torch.arange(w[a])
        dist = (i.view(1, -1) - i.view(-1, 1) + w[a] - 1).clamp(0, 2 * w[a] - 2)  # [i(query), j(key)]
        per_axis.append(table[:, dist])                       # [heads, w, w]
    Rh, Rw, Rd = per_axis
    if emulate_bf16:
        s3 = scale / 3.0
        fold = Rd[:, w[2] - 1, :]                             # [heads, k2]: the dropped column (query i2 = w2-1)
        q16 = lambda t: r16(t * (s3 * LOG2E)) / (s3 * LOG2E)
        # Rh + fold depends on (i0, k0, k2); Rd - fold on (i2, k2) and is exactly zero in the dropped column
        Rhf = q16(Rh[:, :, None, None, :, None, None] + fold[:, None, None, None, None, None, :])
        Rdf = q16(Rd - fold[:, None, :])
        content = (Rhf + q16(Rw)[:, None, :, None, None, :, None] + Rdf[:, None, None, :, None, None, :]) / 3
        heads = content.shape[0]
        N = w[0] * w[1] * w[2]
        content = content.reshape(heads, N, N) * scale
        if n_prompt == 0:
            return content
        tok = r16((sd[f"{prefix}pe.weights_token"] @ sd[f"{prefix}pe.enc_token.0"].t()) * (scale * LOG2E)) / LOG2E
        return torch.cat([content, tok[:, None, :n_prompt].expand(heads, N, n_prompt)], dim=2)
    content = (Rh[:, :, None, None, :, None, None]
               + Rw[:, None, :, None, None, :, None]
               + Rd[:, None, None, :, None, None, :]) / 3
    heads = content.shape[0]
    N = w[0] * w[1] * w[2]
    content = content.reshape(heads, N, N) * scale
    if n_prompt == 0:
        return content
    tok_emb = sd[f"{prefix}pe.enc_token.0"]                  # [tokens, E]  (max_prompts == 1)
    tok_w = sd[f"{prefix}pe.weights_token"]                  # [heads, E]
    tok = (tok_w @ tok_emb.t()) * scale                       # [heads, tokens]
    return torch.cat([content, tok[:, None, :n_prompt].expand(heads, N, n_prompt)], dim=2)


# ----------------------------------------------------------------------------
# shift mask  (a8)
# ----------------------------------------------------------------------------
def region_ids(geo: BlockGeometry) -> Tensor:
    """Region id of every rolled-frame coordinate, ``[Lp0, Lp1, Lp2]`` (int64).

    Follows swin_block.py:312-350.  Per axis: id 0 for ``j < L-w``, 1 for
    ``L-w <= j < L-s``, 2 for ``j >= L-s``; when ``s == 0`` the third slice is
    ``slice(-0, None)`` = everything and overwrites, so id == 2 everywhere
    (:320-341).  Ids combine as 9*a + 3*b + c (loop order :334-341).  If the
    block was padded, the box [floor_pad, L - ceil_pad) on every axis is set to
    100 (:345-350) -- in rolled-frame coordinates, as the reference does.
    """
    per_axis = []
    for a in range(3):
        L, w, s = geo.padded[a], geo.window[a], geo.shift[a]
        j = torch.arange(L)
        if s == 0:
            ida = torch.full((L,), 2, dtype=torch.long)
        else:
            ida = (j >= L - w).long() + (j >= L - s).long()
        per_axis.append(ida)
    rid = 9 * per_axis[0].view(-1, 1, 1) + 3 * per_axis[1].view(1, -1, 1) + per_axis[2].view(1, 1, -1)
    rid = rid.clone()
    if geo.has_pad:
        sl = tuple(slice(geo.lo[a], geo.padded[a] - geo.hi[a]) for a in range(3))
        rid[sl] = 100
    return rid


def shift_mask(geo: BlockGeometry) -> Optional[Tensor]:
    """Multiplicative {0,1} mask ``[P, N, N]`` or None for un-shifted blocks.

    Follows swin_block.py:174-203,352-364: tokens of one window may interact
    only when their region ids are equal; the mask MULTIPLIES the logits before
    softmax (window_attention.py:54-56), it is not an additive -inf mask.
    """
    if not geo.has_shift:
        return None
    rid = region_ids(geo).reshape(-1)
    idx = geo.flat_index([geo.rolled_coord(a) for a in range(3)])   # rolled frame, no shift
    r = rid[idx]                                                     # [P, N]
    return (r[:, :, None] == r[:, None, :]).float()


# ----------------------------------------------------------------------------
# attention  (a10)
# ----------------------------------------------------------------------------
def window_attention(y: Tensor, sd: Dict[str, Tensor], prefix: str, heads: int,
                     bias: Optional[Tensor], mask: Optional[Tensor], n_query: int,
                     attn_keep: Optional[Tensor] = None, proj_keep: Optional[Tensor] = None,
                     emulate_bf16: bool = False, zero_ref: bool = False) -> Tensor:
    """Windowed MHSA on normalised tokens ``y [B, P, Nk, C]``; returns the
    projected output for the first ``n_query`` rows ``[B, P, n_query, C]``.

    Follows multi_head_attention/window_attention.py:35-61:
    ``softmax((q k^T * hd**-0.5 + bias) * mask) v`` then ``proj`` (+bias).
    ``bias`` is ``[heads, n_query, Nk]``, ``mask`` ``[P, n_query, Nk]`` (1 = keep,
    0 = logit forced to 0).  Prompt rows are keys/values only, so their query
    rows (discarded by the block, swin_block.py:223-225) are never formed here.

    Training-mode dropout (window_attention.py:33,57,60) is stated with EXPLICIT masks so a test can
    hand over the very mask another implementation drew: ``attn_keep [B,P,heads,n_query,Nk]`` and
    ``proj_keep [B,P,n_query,C]`` hold 0 or 1/(1-p), i.e. ``nn.Dropout``'s multiplier.
    """
    B, P, Nk, C = y.shape
    hd = C // heads
    if C % heads != 0:
        raise ValueError("WindowAttention: The dimension is not compatible with the number of heads!")
    q = F.linear(y[:, :, :n_query], sd[f"{prefix}attn.to_q.weight"])
    k = F.linear(y, sd[f"{prefix}attn.to_k.weight"])
    v = F.linear(y, sd[f"{prefix}attn.to_v.weight"])
    q = q.reshape(B, P, n_query, heads, hd).permute(0, 1, 3, 2, 4)
    k = k.reshape(B, P, Nk, heads, hd).permute(0, 1, 3, 2, 4)
    v = v.reshape(B, P, Nk, heads, hd).permute(0, 1, 3, 2, 4)
    if emulate_bf16:
        q, k, v = r16(q * (hd ** -0.5)), r16(k * LOG2E) / LOG2E, r16(v)
        logits = q @ k.transpose(-1, -2)
    else:
        logits = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    if bias is not None:
        logits = logits + bias[None, None]
    if mask is not None:
        logits = logits * mask[None, :, None]
    if emulate_bf16:
        pe = r16(torch.exp(logits - logits.amax(dim=-1, keepdim=True).detach()))
        if zero_ref:
            # Forward-only HIP calls (nothing saved for backward) form P = exp2(logit * log2 e) against a ZERO reference point
            # (csrc/swin_fwd.hip, ZREF) and round THAT to bf16; only rows whose sum leaves [2^-100, 2^100) are redone
            # against the running maximum.  The rounding pattern of P depends on the reference, so the emulation follows the
            # same rule (row-wise; the kernel redoes whole 16-query tiles: visible only in the sharp-softmax stress cases).
            pe0 = r16(torch.exp(logits.clamp(max=80.0)))
            den0 = pe0.sum(dim=-1, keepdim=True)
            pe = torch.where((den0 < 2.0 ** 100) & (den0 > 2.0 ** -100), pe0, pe)
        denom = pe.sum(dim=-1, keepdim=True)
        if attn_keep is not None:
            pe = pe * attn_keep
        o = r16((pe @ v) / denom).permute(0, 1, 3, 2, 4).reshape(B, P, n_query, C)
    else:
        prob = logits.softmax(dim=-1)
        if attn_keep is not None:
            prob = prob * attn_keep
        o = (prob @ v).permute(0, 1, 3, 2, 4).reshape(B, P, n_query, C)
    out = F.linear(o, sd[f"{prefix}attn.proj.weight"], sd[f"{prefix}attn.proj.bias"])
    if proj_keep is not None:
        out = out * proj_keep
    return out


# ----------------------------------------------------------------------------
# the block  (a5)
# ----------------------------------------------------------------------------
def swin_block(x: Tensor, prompt: Optional[Tensor], sd: Dict[str, Tensor], prefix: str,
               window: Sequence[int], shift_cfg: Sequence[int], heads: int,
               embed_dim: int = 64, attn_keep: Optional[Tensor] = None,
               proj_keep: Optional[Tensor] = None, emulate_bf16: bool = False, zero_ref: bool = False) -> Tensor:
    """One SwinTransformerBlock, ``x [B,C,H,W,D]`` -> same shape (dropout only through the explicit
    multiplier masks of ``window_attention``).

    Follows swin_block.py:145-255.  ``prompt``
    is ``[Np, C]`` (the reference broadcasts the same tokens to every batch
    element and window, swin_unetr.py:55-60, swin_block.py:206-212).

    ``emulate_bf16`` rounds at the HIP path's storage points: both LayerNorm outputs, q/k/v, P, o (see
    ``window_attention``), the post-attention residual t1 and the block output.
    """
    rr = r16 if emulate_bf16 else (lambda t: t)
    B, C = x.shape[:2]
    geo = BlockGeometry(x.shape[2:], window, shift_cfg)
    xl = x.permute(0, 2, 3, 4, 1)
    frame = x.new_zeros((B,) + geo.padded + (C,))
    frame[:, geo.hi[0]:geo.hi[0] + geo.dims[0],
          geo.hi[1]:geo.hi[1] + geo.dims[1],
          geo.hi[2]:geo.hi[2] + geo.dims[2]] = xl
    idx = geo.token_index()                                   # [P, N]
    tok = frame.reshape(B, -1, C)[:, idx]                     # [B, P, N, C]
    n_prompt = 0 if prompt is None else prompt.shape[0]
    if n_prompt:
        tok_all = torch.cat([tok, prompt[None, None].expand(B, geo.P, n_prompt, C)], dim=2)
    else:
        tok_all = tok
    y = rr(F.layer_norm(tok_all, (C,), sd[f"{prefix}attn_norm.weight"], sd[f"{prefix}attn_norm.bias"], 1e-6))
    bias = rel_pos_bias(sd, prefix, window, n_prompt, embed_dim, emulate_bf16)
    mask = shift_mask(geo)
    if mask is not None and n_prompt:
        mask = torch.cat([mask, mask.new_ones(geo.P, geo.N, n_prompt)], dim=2)   # :189-196
    t1 = rr(window_attention(y, sd, prefix, heads, bias, mask, geo.N, attn_keep, proj_keep, emulate_bf16, zero_ref) + tok)
    t2 = rr(t1 + F.linear(rr(F.layer_norm(t1, (C,), sd[f"{prefix}mlp_norm.weight"], sd[f"{prefix}mlp_norm.bias"], 1e-6)),
                          sd[f"{prefix}mlp.weight"], sd[f"{prefix}mlp.bias"]))
    out = x.new_zeros(B, geo.padded[0] * geo.padded[1] * geo.padded[2], C)
    out[:, idx.reshape(-1)] = t2.reshape(B, -1, C)
    out = out.reshape((B,) + geo.padded + (C,))
    out = out[:, geo.lo[0]:geo.padded[0] - geo.hi[0],
              geo.lo[1]:geo.padded[1] - geo.hi[1],
              geo.lo[2]:geo.padded[2] - geo.hi[2]]
    return out.permute(0, 4, 1, 2, 3).contiguous()


def swin_pair(x: Tensor, prompts, sd: Dict[str, Tensor], prefix: str, window: Sequence[int],
              heads: int, embed_dim: int = 64, down: bool = True, merge_last_dim: bool = True,
              emulate_bf16: bool = False) -> Tensor:
    """ConsecutiveSwinBlocks: W-MSA block, SW-MSA block (shift = w // 2), optional
    PatchMerging.  Follows swin_block.py:36-71."""
    shift = tuple(int(w) // 2 for w in window)
    x = swin_block(x, prompts[0], sd, f"{prefix}swin_blocks.0.", window, (0, 0, 0), heads, embed_dim,
                   emulate_bf16=emulate_bf16)
    x = swin_block(x, prompts[1], sd, f"{prefix}swin_blocks.1.", window, shift, heads, embed_dim,
                   emulate_bf16=emulate_bf16)
    if down:
        x = patch_merge(x, sd, f"{prefix}merge.", merge_last_dim, emulate_bf16)
    return x


# ----------------------------------------------------------------------------
# patch merging  (a11)
# ----------------------------------------------------------------------------
def patch_merge(x: Tensor, sd: Dict[str, Tensor], prefix: str, merge_last_dim: bool,
                emulate_bf16: bool = False) -> Tensor:
    """2x2x2 (or 2x2x1) space-to-depth + LayerNorm(eps 1e-6) + bias-free Linear.

    Follows swin_transformer/down.py:21-53.  Odd axes are zero-padded by one AT
    THE FRONT (reversed-tuple effect, :25-28) -- including D when it is not
    merged.  Channel order of the concat: offsets (h,w,d) =
    000,100,010,001,110,101,011,111 (:31-39) or (h,w) = 00,10,01,11 (:42-46).
    """
    B, C, H, W, D = x.shape
    ph, pw, pd = H % 2, W % 2, D % 2
    frame = x.new_zeros(B, C, H + ph, W + pw, D + pd)
    frame[:, :, ph:, pw:, pd:] = x
    if merge_last_dim:
        order = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
        parts = [frame[:, :, a::2, b::2, c::2] for a, b, c in order]
    else:
        order = [(0, 0), (1, 0), (0, 1), (1, 1)]
        parts = [frame[:, :, a::2, b::2, :] for a, b in order]
    cat = torch.cat(parts, dim=1).permute(0, 2, 3, 4, 1)      # [B, h, w, d, kC]
    kc = cat.shape[-1]
    y = F.layer_norm(cat, (kc,), sd[f"{prefix}norm.weight"], sd[f"{prefix}norm.bias"], 1e-6)
    if emulate_bf16:                                          # LayerNorm output and the merged tokens are stored as bf16
        y = r16(y)
    y = F.linear(y, sd[f"{prefix}reduction.weight"])
    if emulate_bf16:
        y = r16(y)
    return y.permute(0, 4, 1, 2, 3).contiguous()


# ----------------------------------------------------------------------------
# decoder up block  (a13)
# ----------------------------------------------------------------------------
def batch_norm_train(x: Tensor, sd: Dict[str, Tensor], prefix: str, eps: float, training: bool,
                     new_buffers: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """BatchNorm3d; in training mode uses batch statistics and records the
    running-stat update (momentum 0.1, unbiased variance) in ``new_buffers``."""
    rm = sd[f"{prefix}running_mean"].clone()
    rv = sd[f"{prefix}running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[f"{prefix}weight"], sd[f"{prefix}bias"], training, 0.1, eps)
    if training and new_buffers is not None:
        new_buffers[f"{prefix}running_mean"] = rm
        new_buffers[f"{prefix}running_var"] = rv
        new_buffers[f"{prefix}num_batches_tracked"] = sd[f"{prefix}num_batches_tracked"] + 1
    return y


def up_block(x: Tensor, skip: Tensor, prompts, sd: Dict[str, Tensor], prefix: str,
             strides: Sequence[int], window: Sequence[int], heads: int, embed_dim: int = 64,
             training: bool = True, new_buffers: Optional[Dict[str, Tensor]] = None,
             emulate_bf16: bool = False) -> Tensor:
    """SwinUpBlock: trilinear upsample (align_corners=False) -> crop to skip ->
    concat [up, skip] -> BatchNorm3d(eps 1e-5) -> LeakyReLU(0.01) -> Conv3d 3^3 p1
    (+bias) -> two Swin blocks without merge.

    Follows swin_unetr/unet_blocks.py:31-76.  The three MONAI factories resolve
    to nn.LeakyReLU(0.01), nn.BatchNorm3d(C) and a bias-carrying nn.Conv3d under
    key ``conv_concat.conv`` (MONAI is absent from the image, so this
    mapping is "parity unpinned at the MONAI boundary").
    """
    up = F.interpolate(x, scale_factor=tuple(float(s) for s in strides), mode="trilinear", align_corners=False)
    up = up[..., :skip.shape[2], :skip.shape[3], :skip.shape[4]]
    rr = r16 if emulate_bf16 else (lambda t: t)              # stored: the concatenated tensor, act(BN(.)), the conv output
    cat = rr(torch.cat([up, skip], dim=1))
    y = batch_norm_train(cat, sd, f"{prefix}norm_concat.", 1e-5, training, new_buffers)
    y = rr(F.leaky_relu(y, 0.01))
    y = rr(F.conv3d(y, sd[f"{prefix}conv_concat.conv.weight"], sd[f"{prefix}conv_concat.conv.bias"], padding=1))
    return swin_pair(y, prompts, sd, f"{prefix}swin_layer.", window, heads, embed_dim, down=False,
                     emulate_bf16=emulate_bf16)
