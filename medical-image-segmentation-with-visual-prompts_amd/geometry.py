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
from typing import Tuple

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


@lru_cache(maxsize=256)
def block_tables(dims, window, shift_cfg, device_str) -> BlockTables:
    meta, (src, dst, rid) = build_tables_numpy(dims, window, shift_cfg)
    dev = torch.device(device_str)
    return BlockTables(tok_src=torch.from_numpy(src).to(dev), tok_dst=torch.from_numpy(dst).to(dev),
                       tok_rid=torch.from_numpy(rid).to(dev), **meta)
