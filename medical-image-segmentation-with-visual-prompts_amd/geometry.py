"""Host-side index tables for one Swin block call (immutable per shape).

Semantics follow the reference exactly (SURVEY Appendix A.1;
swin_transformer/swin_block.py:145-178 pad/shift, :247-253 crop, :265-270
effective shift, :292-309 strided windows, :312-364 region ids), including its
quirks: every axis padded by ``w - dim % w`` as soon as one axis needs padding,
ceil(t/2) zeros in front but crop from floor(t/2), region box in rolled-frame
coordinates.  Built with numpy, cached per (dims, window, shift_cfg, device).
"""
from dataclasses import dataclass
from functools import lru_cache
from typing import Optional, Tuple

import numpy as np
import torch


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass(frozen=True)
class BlockTables:
    dims: Tuple[int, int, int]
    window: Tuple[int, int, int]
    shift: Tuple[int, int, int]          # effective shift
    padded: Tuple[int, int, int]
    nwin: Tuple[int, int, int]
    P: int
    Nq: int
    Nqp: int
    has_mask: bool
    tok_src: torch.Tensor                 # int32 [P*Nqp] on device
    tok_dst: torch.Tensor
    tok_rid: torch.Tensor
    mask_words: Optional[torch.Tensor] = None     # shifted blocks: int64 [P][Nqp/16][Nqp/16][4] lane masks (mivp.h), forward layout
    cut_flags: Optional[torch.Tensor] = None      # uint8 [P]


def _axis_tables(dim, w, s, t):
    lo, hi = t // 2, t - t // 2
    L = dim + t
    n = L // w
    p = np.arange(n).reshape(n, 1)
    i = np.arange(w).reshape(1, w)
    j = i * n + p                                    # rolled-frame coordinate [n, w]
    c = (j + s) % L                                  # padded-frame coordinate
    src = np.where((c - hi >= 0) & (c - hi < dim), c - hi, -1)
    dst = np.where((c - lo >= 0) & (c - lo < dim), c - lo, -1)
    if s == 0:
        rid = np.full_like(j, 2)
    else:
        rid = (j >= L - w).astype(np.int64) + (j >= L - s).astype(np.int64)
    box = (j >= lo) & (j < L - hi)
    return n, L, src, dst, rid, box


def build_tables_numpy(dims, window, shift_cfg):
    dims = tuple(int(d) for d in dims)
    window = tuple(int(w) for w in window)
    shift = tuple(int(s) if d > w else 0 for s, d, w in zip(shift_cfg, dims, window))
    need = any(d % w != 0 for d, w in zip(dims, window))
    total = tuple((w - d % w) if need else 0 for d, w in zip(dims, window))
    ax = [_axis_tables(dims[a], window[a], shift[a], total[a]) for a in range(3)]
    n = tuple(a[0] for a in ax)
    L = tuple(a[1] for a in ax)
    w0, w1, w2 = window

    def comb(k, f):
        a0 = ax[0][k].reshape(n[0], 1, 1, w0, 1, 1)
        a1 = ax[1][k].reshape(1, n[1], 1, 1, w1, 1)
        a2 = ax[2][k].reshape(1, 1, n[2], 1, 1, w2)
        return f(a0, a1, a2)

    H, W, D = dims
    valid_src = comb(2, lambda a, b, c: (a >= 0) & (b >= 0) & (c >= 0))
    src = comb(2, lambda a, b, c: (a * W + b) * D + c)
    src = np.where(valid_src, src, -1)
    valid_dst = comb(3, lambda a, b, c: (a >= 0) & (b >= 0) & (c >= 0))
    dst = comb(3, lambda a, b, c: (a * W + b) * D + c)
    dst = np.where(valid_dst, dst, -1)
    rid = comb(4, lambda a, b, c: 9 * a + 3 * b + c)
    if any(t > 0 for t in total):
        inbox = comb(5, lambda a, b, c: a & b & c)
        rid = np.where(inbox, 100, rid)
    P = n[0] * n[1] * n[2]
    Nq = w0 * w1 * w2
    Nqp = round_up(Nq, 16)
    out = []
    for arr, fill in ((src, -2), (dst, -1), (rid, 0)):
        full = np.full((P, Nqp), fill, dtype=np.int32)
        full[:, :Nq] = arr.reshape(P, Nq)
        out.append(full.reshape(-1))
    meta = dict(dims=dims, window=window, shift=shift, padded=L, nwin=n, P=P, Nq=Nq, Nqp=Nqp,
                has_mask=any(s > 0 for s in shift))
    return meta, out


def mask_words_numpy(rid: np.ndarray, P: int, Nq: int, Nqp: int):
    """The shift mask as lane masks (mivp.h ``mask_words``): ``rid`` int32 [P * Nqp] region ids (swin_block.py:312-364 through
    build_tables_numpy).  Returns (forward words, backward words, cut flags).  Logit (query n, key m) of a window survives the
    multiplicative mask (swin_block.py:187-200) when both slots carry the same region id; key rows >= Nq (content padding)
    always survive, query rows >= Nq compare as class 0 -- exactly the classes the byte-compare kernels use.
      forward  word j of (qt, kt): bit 16 g + r <-> (query 16 qt + r, key 16 kt + 4 g + j)        (query on the lane)
      backward word j of (qt, kt): bit 16 g + r <-> (query 16 qt + 4 g + j, key 16 kt + r)        (key on the lane)"""
    rid = rid.reshape(P, Nqp)
    slot = np.arange(Nqp)
    qcls = np.where(slot[None, :] < Nq, rid, 0)
    kcls = np.where(slot[None, :] < Nq, rid, 254)
    nt = Nqp // 16
    fwd = np.empty((P, nt, nt, 4), np.uint64)
    bwd = np.empty((P, nt, nt, 4), np.uint64)
    cut = np.zeros(P, np.uint8)
    for p in range(P):                                            # (one window at a time: 352 x 352 booleans)
        live = (kcls[p][None, :] == 254) | (kcls[p][None, :] == qcls[p][:, None])          # [query, key]
        cut[p] = np.unique(rid[p, :Nq]).size > 1
        a = live.reshape(nt, 16, nt, 4, 4)                        # [qt, r, kt, g, j]
        bits = np.ascontiguousarray(a.transpose(0, 2, 4, 3, 1)).reshape(nt, nt, 4, 64)      # [qt, kt, j, (g, r)]
        fwd[p] = np.packbits(bits, axis=-1, bitorder="little").view(np.uint64).reshape(nt, nt, 4)
        b = live.reshape(nt, 4, 4, nt, 16)                        # [qt, g, j, kt, r]
        bits = np.ascontiguousarray(b.transpose(0, 3, 2, 1, 4)).reshape(nt, nt, 4, 64)      # [qt, kt, j, (g, r)]
        bwd[p] = np.packbits(bits, axis=-1, bitorder="little").view(np.uint64).reshape(nt, nt, 4)
    return fwd, bwd, cut


@lru_cache(maxsize=256)
def block_tables(dims, window, shift_cfg, device_str) -> BlockTables:
    meta, (src, dst, rid) = build_tables_numpy(dims, window, shift_cfg)
    dev = torch.device(device_str)
    extra = {}
    if meta["has_mask"]:
        fwd, bwd, cut = mask_words_numpy(rid, meta["P"], meta["Nq"], meta["Nqp"])
        # (the backward layout was built and measured too: in the fused backward the scalar loads share the LDS wait counter with
        #  the tile loop's dS exchange and the launch got 7 % SLOWER -- 419 vs 391 us at stage 0 -- so only the forward uses words)
        extra = dict(mask_words=torch.from_numpy(fwd.view(np.int64)).to(dev), cut_flags=torch.from_numpy(cut).to(dev))
    return BlockTables(tok_src=torch.from_numpy(src).to(dev), tok_dst=torch.from_numpy(dst).to(dev),
                       tok_rid=torch.from_numpy(rid).to(dev), **meta, **extra)
