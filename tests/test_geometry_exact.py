"""Bit-exact CPU tests of the product's host-side index tables (mivp_amd/geometry.py) -- the only place where the
reference's padding / shift / strided-window quirks live on the product side (swin_block.py:145-178 pad + roll,
:247-253 crop, :265-270 effective shift, :292-309 strided windows, :312-364 region ids).

Two independent pins, both integer-exact:
  * ``tok_rid``: the equality pattern of the product's region ids must equal the reference's own ``get_attn_mask``
    output captured in tests/golden/mask_{a..f}.npz;
  * ``tok_src`` / ``tok_dst``: must describe the same gather / scatter as the oracle's ``BlockGeometry`` (which is pinned
    against the reference's block goldens in test_oracle_golden.py) for every block golden's shape and for the stage
    shapes of the 96^3 and 128^3 configurations.
"""
import numpy as np
import pytest
import torch

import mivp_amd  # noqa: F401
from mivp_amd.geometry import build_tables_numpy
from oracle import swin_ref as S
from conftest import load_fixture


@pytest.mark.parametrize("tag", list("abcdef"))
def test_region_ids_reproduce_reference_masks_bit_exactly(tag):
    fx = load_fixture(f"mask_{tag}")
    m = fx.meta
    meta, (_, _, rid) = build_tables_numpy(m["dims"], m["window"], m["shift"])
    assert list(meta["padded"]) == list(m["padded"])
    P, Nq, Nqp = meta["P"], meta["Nq"], meta["Nqp"]
    rid = rid.reshape(P, Nqp)[:, :Nq]
    got = (rid[:, :, None] == rid[:, None, :])
    want = fx["out"]["mask"].numpy() != 0
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert meta["has_mask"] == any(s > 0 for s in meta["shift"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_mask_words_reproduce_reference_masks_bit_exactly(tag):
    """The lane-mask form of the shift mask (geometry.mask_words_numpy, mivp.h ``mask_words``: what the attention forward reads
    through scalar loads) against the reference's own ``get_attn_mask`` outputs: every bit of every word for the content
    slots, "always survives" for the padding key rows, the per-window cut flag."""
    from mivp_amd.geometry import mask_words_numpy
    fx = load_fixture(f"mask_{tag}")
    m = fx.meta
    meta, (_, _, rid) = build_tables_numpy(m["dims"], m["window"], m["shift"])
    if not meta["has_mask"]:
        return
    P, Nq, Nqp = meta["P"], meta["Nq"], meta["Nqp"]
    fwd, bwd, cut = mask_words_numpy(rid, P, Nq, Nqp)
    want = fx["out"]["mask"].numpy() != 0                          # [P, Nq, Nq]: 1 where the logit survives
    nt = Nqp // 16
    assert fwd.shape == bwd.shape == (P, nt, nt, 4) and fwd.dtype == np.uint64
    lane = np.arange(64, dtype=np.uint64)
    g, r = (lane >> np.uint64(4)).astype(np.int64), (lane & np.uint64(15)).astype(np.int64)
    live_f = np.zeros((P, Nqp, Nqp), bool)
    live_b = np.zeros((P, Nqp, Nqp), bool)
    for qt in range(nt):
        for kt in range(nt):
            for j in range(4):
                bits_f = ((fwd[:, qt, kt, j][:, None] >> lane[None, :]) & np.uint64(1)).astype(bool)      # [P, 64]
                live_f[:, 16 * qt + r, 16 * kt + 4 * g + j] = bits_f
                bits_b = ((bwd[:, qt, kt, j][:, None] >> lane[None, :]) & np.uint64(1)).astype(bool)
                live_b[:, 16 * qt + 4 * g + j, 16 * kt + r] = bits_b
    assert np.array_equal(live_f[:, :Nq, :Nq], want)
    assert np.array_equal(live_b, live_f)                          # the two layouts hold the same matrix
    assert live_f[:, :, Nq:].all()                                 # padding key rows always survive (their bias excludes them)
    assert np.array_equal(cut != 0, np.array([np.unique(x).size > 1 for x in rid.reshape(P, Nqp)[:, :Nq]]))


def _oracle_maps(dims, window, shift_cfg):
    """(src, dst) voxel index per (window, slot) from the oracle's geometry: the padded frame holds the volume at
    [hi, hi + dim) (pad: ceil in front) and the output is cropped from [lo, L - hi) (floor in front)."""
    geo = S.BlockGeometry(dims, window, shift_cfg)
    coords = [geo.padded_coord(a) for a in range(3)]          # [n_a, w_a] padded-frame coordinate

    def lin(offsets):
        per = []
        for a in range(3):
            c = coords[a] - offsets[a]
            per.append(torch.where((c >= 0) & (c < geo.dims[a]), c, torch.full_like(c, -1)))
        n, w = geo.nwin, geo.window
        c0 = per[0].view(n[0], 1, 1, w[0], 1, 1)
        c1 = per[1].view(1, n[1], 1, 1, w[1], 1)
        c2 = per[2].view(1, 1, n[2], 1, 1, w[2])
        ok = (c0 >= 0) & (c1 >= 0) & (c2 >= 0)
        idx = (c0 * geo.dims[1] + c1) * geo.dims[2] + c2
        return torch.where(ok, idx, torch.full_like(idx, -1)).reshape(geo.P, geo.N)

    return geo, lin(geo.hi), lin(geo.lo)


SHAPES = [
    # block goldens (tests/golden/gen_golden.py G4)
    ((6, 6, 4), (3, 3, 2), (0, 0, 0)), ((6, 6, 4), (3, 3, 2), (1, 1, 1)), ((7, 5, 3), (3, 3, 2), (1, 1, 1)),
    ((4, 6, 4), (3, 3, 2), (0, 0, 0)), ((2, 3, 6), (3, 3, 2), (1, 1, 1)), ((2, 2, 5), (3, 3, 2), (0, 0, 0)),
    ((8, 8, 4), (4, 4, 2), (2, 2, 1)),
    # 96^3 stages, window 7 (48 -> 49: the odd pad with the one-voxel shift) and the yml window
    ((48, 48, 48), (7, 7, 7), (0, 0, 0)), ((48, 48, 48), (7, 7, 7), (3, 3, 3)),
    ((24, 24, 24), (7, 7, 7), (3, 3, 3)), ((12, 12, 24), (7, 7, 7), (3, 3, 3)), ((6, 6, 24), (7, 7, 7), (3, 3, 3)),
    ((48, 48, 48), (8, 8, 4), (4, 4, 2)), ((12, 12, 24), (8, 8, 4), (4, 4, 2)),
    # 128^3 stages
    ((64, 64, 64), (7, 7, 7), (3, 3, 3)), ((32, 32, 32), (7, 7, 7), (3, 3, 3)), ((16, 16, 32), (7, 7, 7), (3, 3, 3)),
    ((8, 8, 32), (7, 7, 7), (3, 3, 3)), ((64, 64, 64), (8, 8, 4), (4, 4, 2)),
]


@pytest.mark.parametrize("dims,window,shift", SHAPES)
def test_gather_scatter_tables_equal_oracle_geometry(dims, window, shift):
    meta, (src, dst, rid) = build_tables_numpy(dims, window, shift)
    geo, osrc, odst = _oracle_maps(dims, window, shift)
    P, Nq, Nqp = meta["P"], meta["Nq"], meta["Nqp"]
    assert (P, Nq) == (geo.P, geo.N) and tuple(meta["shift"]) == tuple(geo.shift) and tuple(meta["padded"]) == tuple(geo.padded)
    src = src.reshape(P, Nqp)
    dst = dst.reshape(P, Nqp)
    assert np.array_equal(src[:, :Nq], osrc.numpy())          # -1 = zero-pad token (still attends, A.1 step 7)
    assert np.array_equal(dst[:, :Nq], odst.numpy())          # -1 = cropped away
    assert (src[:, Nq:] == -2).all() and (dst[:, Nq:] == -1).all()      # padding slots of the 16-aligned tile
    # every voxel is written exactly once by the scatter
    live = dst[:, :Nq][dst[:, :Nq] >= 0]
    assert np.array_equal(np.sort(live), np.arange(dims[0] * dims[1] * dims[2]))
    # region ids equal the oracle's mask pattern (itself bit-exact against the reference's goldens)
    mask = S.shift_mask(geo)
    r = rid.reshape(P, Nqp)[:, :Nq]
    if mask is None:
        assert not meta["has_mask"]
    elif P * Nq * Nq <= 64 * 343 * 343:                        # the big stages: sampled windows only
        assert np.array_equal(r[:, :, None] == r[:, None, :], mask.numpy() != 0)
    else:
        for p in (0, P // 2, P - 1, P - geo.nwin[2], geo.nwin[1] * geo.nwin[2] - 1):
            assert np.array_equal(r[p][:, None] == r[p][None, :], mask[p].numpy() != 0)
