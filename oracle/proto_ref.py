"""Oracle (test infrastructure): the students/teacher step's objective and bookkeeping, restated.

* ``clustered_prototype_loss``  -- losses/clustered_prototype_loss.py:13-206 (soft k-means prototypes on the teacher
  embedding, position-weighted; students are pulled to the teacher's prototype assignment at the nearest teacher point)
* ``ema_update``                -- momentum_model/momentum_model.py:27-36
* ``warmup_cosine_factor``      -- modules/utils.py:67-89
* ``coord_grid``                -- datasets/transforms.py:336-344

Restated in index form: the reference's ``affine_grid`` + ``grid_sample`` (identity transform, bilinear, align_corners =
False) is a separable linear interpolation at the cell centres of the reduced grid, written here as three per-axis
interpolation matrices; every sample point lies inside the volume, so the reflection padding never acts.  Pinned by
tests/golden/proto_{a,b,c}.npz, momentum_model.npz and utils_metrics_schedule.npz (tests/test_oracle_golden.py)."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def coord_grid(dims: Sequence[int]) -> Tensor:
    """[3, H, W, D] voxel coordinates centred on the volume (datasets/transforms.py:336-344)."""
    axes = [torch.arange(n, dtype=torch.float32) - (n - 1) / 2.0 for n in dims]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), 0)


def reduced_size(dims: Sequence[int], reduction_factor: float) -> List[int]:
    """clustered_prototype_loss.py:168-171."""
    return [max(int(n // reduction_factor), 1) for n in dims]


def axis_interp_matrix(n_in: int, n_out: int) -> Tensor:
    """[n_out, n_in]: row i interpolates at x = ((2 i + 1) n_in / n_out - 1) / 2, the position grid_sample(align_corners=False)
    assigns to the i-th cell centre of an n_out-point identity grid."""
    m = torch.zeros(n_out, n_in)
    for i in range(n_out):
        x = ((2 * i + 1) * n_in / n_out - 1.0) / 2.0
        x = min(max(x, 0.0), n_in - 1.0)
        lo = min(int(math.floor(x)), n_in - 1)
        hi = min(lo + 1, n_in - 1)
        w = x - lo
        m[i, lo] += 1.0 - w
        m[i, hi] += w
    return m


def sample_volume(vol: Tensor, out_dims: Sequence[int], jitter: Optional[Sequence[int]] = None) -> Tensor:
    """``vol [B, C, H, W, D]`` -> ``[B, C, H', W', D']`` (clustered_prototype_loss.py:162-204).  ``jitter`` = the six
    crop offsets (h0, h1, w0, w1, d0, d1) the reference draws; the reduced size was fixed BEFORE the crop (:168-171)."""
    if jitter is not None:
        j = [int(v) for v in jitter]
        vol = vol[:, :, j[0]: vol.shape[2] - j[1], j[2]: vol.shape[3] - j[3], j[4]: vol.shape[4] - j[5]]
    mh = axis_interp_matrix(vol.shape[2], out_dims[0]).to(vol)
    mw = axis_interp_matrix(vol.shape[3], out_dims[1]).to(vol)
    md = axis_interp_matrix(vol.shape[4], out_dims[2]).to(vol)
    return torch.einsum("bchwd,ih,jw,kd->bcijk", vol, mh, mw, md)


def _flat(t: Tensor) -> Tensor:
    return t.flatten(2).transpose(1, 2)                       # [B, C, ...] -> [B, N, C]


def _pair_dist(cx: Tensor, cy: Tensor) -> Tensor:
    """[B, Nx, Ny] Euclidean distance between coordinate sets [B, 3, ...] (:141-146)."""
    return (_flat(cx)[:, :, None, :] - _flat(cy)[:, None, :, :]).norm(dim=-1)


def clustered_prototype_loss(emb_s: List[Tensor], emb_t: Tensor, coord_s: List[Tensor], coord_t: Tensor,
                             jitters: List[Sequence[int]], reduction_factor: float = 8.0, k_means_iterations: int = 3,
                             fwhm: float = 128.0, temp_s: float = 0.066, temp_t: float = 0.033,
                             max_dist: float = 4.0) -> Tensor:
    """ClusteredPrototypeLoss.forward (clustered_prototype_loss.py:24-60).  ``jitters[i]`` replaces the reference's
    ``torch.randint(0, ceil(reduction_factor), (6,))`` draw for student i (:173-178)."""
    sigma2 = (fwhm / 2.355) ** 2
    rs_t = reduced_size(emb_t.shape[2:], reduction_factor)
    rs_p = reduced_size(emb_t.shape[2:], reduction_factor * 2)
    e_p, c_p = _flat(sample_volume(emb_t, rs_p)), sample_volume(coord_t, rs_p)              # initial prototypes (:34-35)
    e_t, c_t = _flat(sample_volume(emb_t, rs_t)), sample_volume(coord_t, rs_t)              # teacher points (:37-38)
    e_t_n = F.normalize(e_t, dim=-1)
    e_p_n = F.normalize(e_p, dim=-1)
    pshape = c_p.shape

    def soft_assign():
        sim = torch.softmax(e_t_n @ e_p_n.transpose(1, 2) / temp_t, dim=-1)                # [B, Nt, P]
        return sim * torch.exp(-_pair_dist(c_t, c_p) ** 2 / (2 * sigma2))

    for _ in range(k_means_iterations):                      # cluster_prototype (:87-138)
        w = soft_assign()
        den = w.sum(dim=1).unsqueeze(-1)                      # [B, P, 1]
        e_p = (w.transpose(1, 2) @ e_t) / den
        e_p_n = F.normalize(e_p, dim=-1)
        c_p = ((w.transpose(1, 2) @ _flat(c_t)) / den).transpose(1, 2).reshape(pshape)
    sim_t_p = soft_assign()

    total = emb_t.new_zeros(())
    for i in range(len(emb_s)):                               # assign_prototype (:63-84)
        rs_s = reduced_size(emb_s[i].shape[2:], reduction_factor)
        e_z = _flat(sample_volume(emb_s[i], rs_s, jitters[i]))
        c_z = sample_volume(coord_s[i], rs_s, jitters[i])
        dist = _pair_dist(c_z, c_t)
        dmin, idx = dist.min(dim=-1)
        keep = dmin <= max_dist
        sim = torch.softmax(F.normalize(e_z, dim=-1) @ e_p_n.transpose(1, 2) / temp_s, dim=-1)
        per_batch = []
        for b in range(e_z.shape[0]):
            target = sim_t_p[b][idx[b]][keep[b]]
            logp = torch.clamp(torch.log(sim[b][keep[b]] + 1e-16), min=-1e3, max=-0.0)
            per_batch.append(-(target * logp).sum(dim=1).mean(dim=0))
        total = total + torch.stack(per_batch).mean()
    return total


def ema_update(teacher: Tensor, student: Tensor, tau: float) -> Tensor:
    """momentum_model.py:33-36."""
    return tau * teacher + (1 - tau) * student


def warmup_cosine_factor(step: int, warmup_steps: int, t_total: int, cycles: float = 0.5) -> float:
    """utils.py:81-89: the multiplier LambdaLR applies to each group's base lr at scheduler step ``step``."""
    if step < warmup_steps:
        return float(step) / float(max(1.0, warmup_steps))
    progress = float(step - warmup_steps) / float(max(1, t_total - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(cycles) * 2.0 * progress)))


def map_label_indices(masks: Tensor, active_labels: Sequence[int]) -> Tensor:
    """utils.py:372-388 (out of place): labels outside ``active_labels`` -> 0, the sorted active labels -> 0..n-1, applied
    SEQUENTIALLY in sorted order like the reference's in-place loop (a label already mapped can be hit again)."""
    labels = sorted(active_labels)
    out = masks.clone()
    active = torch.zeros_like(out, dtype=torch.bool)
    for lbl in labels:
        active |= out == float(lbl)
    out[~active] = 0
    for new, lbl in enumerate(labels):
        out[out == lbl] = float(new)
    return out
