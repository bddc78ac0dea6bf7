"""Objectives of the reference's trainers on the device.

* ``ClusteredPrototypeLoss``  -- drop-in for losses/clustered_prototype_loss.py:13-206 (same constructor and ``forward``
  signature).  The point sampling of the full-resolution latents (the only part that touches big tensors: the model's
  channels-last bf16 ``latent_outputs``) is a pair of HIP kernels (csrc/proto.hip: forward gather, backward as a gather over
  voxels -- no atomics); the soft k-means / assignment algebra runs on the sampled points ([B, N / r^3, C] f32, a few
  hundred rows at the reference's sizes) with stock PyTorch GPU tensor ops.
* ``dice_focal_loss``         -- lives in train.py (fused HIP kernel, csrc/loss.hip).
* ``dice_loss``               -- MONAI ``DiceLoss(include_background, to_onehot_y=True, softmax=True)`` as documented
  (students_teacher.py:96-100); the Dice-focal kernels without the focal term; MONAI is absent from the image: parity
  unpinned, restated in oracle/loss_ref.py::dice_loss.
"""
from __future__ import annotations

import ctypes as C
import math
from functools import lru_cache
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L


def reduced_size(dims: Sequence[int], reduction_factor: float) -> List[int]:
    return [max(int(n // reduction_factor), 1) for n in dims]


@lru_cache(maxsize=2048)
def _axis_np(n_full: int, lo_crop: int, n_crop: int, n_out: int):
    """Interpolation taps of one axis: forward (lo, hi, w) per output index (absolute voxel coordinates) and backward
    (i1, w1, i2, w2) per voxel coordinate.  Sample i sits at x = ((2 i + 1) n_crop / n_out - 1) / 2 of the cropped axis
    (grid_sample with align_corners = False on an identity grid)."""
    lo = np.zeros(n_out, np.int32); hi = np.zeros(n_out, np.int32); w = np.zeros(n_out, np.float32)
    i1 = np.full(n_full, -1, np.int32); i2 = np.full(n_full, -1, np.int32)
    w1 = np.zeros(n_full, np.float32); w2 = np.zeros(n_full, np.float32)

    def add(c, i, wt):
        if wt == 0.0:
            return
        if i1[c] == i or i1[c] < 0:
            i1[c] = i; w1[c] += wt
        elif i2[c] == i or i2[c] < 0:
            i2[c] = i; w2[c] += wt
        else:
            raise RuntimeError("sample spacing below one voxel: more than two samples touch a voxel")

    for i in range(n_out):
        x = ((2 * i + 1) * n_crop / n_out - 1.0) / 2.0
        x = min(max(x, 0.0), n_crop - 1.0)
        l = min(int(math.floor(x)), n_crop - 1)
        h = min(l + 1, n_crop - 1)
        ww = np.float32(x - l)
        lo[i], hi[i], w[i] = lo_crop + l, lo_crop + h, ww
        add(lo_crop + l, i, float(np.float32(1.0) - ww))
        add(lo_crop + h, i, float(ww))
    return lo, hi, w, i1, w1, i2, w2


@lru_cache(maxsize=512)
def _axis_tables(n_full: int, lo_crop: int, n_crop: int, n_out: int, device_str: str):
    dev = torch.device(device_str)
    return tuple(torch.from_numpy(a).to(dev) for a in _axis_np(n_full, lo_crop, n_crop, n_out))


@lru_cache(maxsize=2048)
def _axis_packed_pinned(n_full: int, lo_crop: int, n_crop: int, n_out: int):
    """The seven tables of one axis as ONE pinned 4-byte-word buffer [3 n_out + 4 n_full] (lo | hi | w | i1 | w1 | i2 | w2):
    the source of a JitterSlot refresh (one asynchronous copy per axis; the buffer is never written again)."""
    words = np.concatenate([a.view(np.int32) for a in _axis_np(n_full, lo_crop, n_crop, n_out)])
    return torch.from_numpy(words).pin_memory()


class JitterSlot:
    """Persistent device tables of ONE student's jittered sampling grid.  ``_SamplePointsFn`` normally takes its tables from
    a cache keyed by the jitter, i.e. the table POINTERS change with the jitter; a recorded graph holds pointers, so in
    graph mode (``ClusteredPrototypeLoss(static_jitter=True)``) the kernels read these fixed buffers and a new jitter is
    a refresh of their CONTENT (``load``, three small host-to-device copies, stream-ordered before the replay)."""

    def __init__(self):
        self.geo = None
        self.bufs = None

    def load(self, dims, out_dims, jitter, device):
        geo = (tuple(int(v) for v in dims), tuple(int(v) for v in out_dims), str(device))
        capturing = torch.cuda.is_current_stream_capturing()
        if self.geo != geo:
            if capturing:
                raise RuntimeError("JitterSlot: the sampling geometry changed between the eager warm-up and the recording")
            self.bufs = [torch.zeros(3 * od + 4 * n, dtype=torch.int32, device=device) for n, od in zip(geo[0], geo[1])]
            self.geo = geo
        if capturing:
            return                                              # a recording must not freeze today's content into the graph
        j = [int(v) for v in jitter] if jitter is not None else [0] * 6
        for a, (n, od) in enumerate(zip(geo[0], geo[1])):
            lo, nc = j[2 * a], n - j[2 * a] - j[2 * a + 1]
            if nc < 1:
                raise ValueError("jitter crop leaves an empty volume")
            self.bufs[a].copy_(_axis_packed_pinned(n, lo, nc, od), non_blocking=True)

    def reload(self, jitter):
        if self.geo is None:
            raise RuntimeError("JitterSlot.reload before the first load")
        self.load(self.geo[0], self.geo[1], jitter, torch.device(self.geo[2]))

    def tables(self, dims, out_dims):
        if self.geo is None or self.geo[0] != tuple(int(v) for v in dims) or self.geo[1] != tuple(int(v) for v in out_dims):
            raise RuntimeError("JitterSlot: volume / grid size differs from the loaded tables")
        out = []
        for buf, n, od in zip(self.bufs, self.geo[0], self.geo[1]):
            f = buf.view(torch.float32)
            out.append((buf[:od], buf[od:2 * od], f[2 * od:3 * od], buf[3 * od:3 * od + n], f[3 * od + n:3 * od + 2 * n],
                        buf[3 * od + 2 * n:3 * od + 3 * n], f[3 * od + 3 * n:3 * od + 4 * n]))
        return out


def _ptr3(ts, ctype):
    return (C.POINTER(ctype) * 3)(*[C.cast(t.data_ptr(), C.POINTER(ctype)) for t in ts])


class _SamplePointsFn(torch.autograd.Function):
    """vol [B, C, H, W, D] (the model's channels-first VIEW of channels-last bf16 storage, or a contiguous f32 channels-first
    tensor) -> f32 [B, N, C] at the reduced grid's cell centres of the (jitter-cropped) volume."""

    @staticmethod
    def forward(ctx, vol, out_dims, jitter, slot=None):
        B, Cc, H, W, D = vol.shape
        if slot is not None:
            tabs = slot.tables((H, W, D), out_dims)             # content = the jitter last loaded into the slot
        else:
            j = [int(v) for v in jitter] if jitter is not None else [0] * 6
            crop = [(j[0], H - j[0] - j[1]), (j[2], W - j[2] - j[3]), (j[4], D - j[4] - j[5])]
            if min(c[1] for c in crop) < 1:
                raise ValueError("jitter crop leaves an empty volume")
            tabs = [_axis_tables(n, lo, nc, int(od), str(vol.device)) for n, (lo, nc), od in zip((H, W, D), crop, out_dims)]
        base = vol.permute(0, 2, 3, 4, 1)
        if base.is_contiguous():
            src, clast = base, 1
        elif vol.is_contiguous():
            src, clast = vol, 0
        else:
            src, clast = vol.contiguous(), 0
        if src.dtype not in (torch.bfloat16, torch.float32):
            src = src.float()
        od = (C.c_int32 * 3)(*[int(v) for v in out_dims])
        N = int(out_dims[0]) * int(out_dims[1]) * int(out_dims[2])
        out = torch.empty((B, N, Cc), dtype=torch.float32, device=vol.device)
        L.call("mivp_sample_points_fwd", L.ptr(src), C.c_int32(1 if src.dtype == torch.bfloat16 else 0), C.c_int32(clast),
               C.c_int32(B), C.c_int32(H), C.c_int32(W), C.c_int32(D), C.c_int32(Cc), od,
               _ptr3([t[0] for t in tabs], C.c_int32), _ptr3([t[1] for t in tabs], C.c_int32),
               _ptr3([t[2] for t in tabs], C.c_float), L.ptr(out), L.stream())
        ctx.tabs = tabs
        ctx.meta = (B, Cc, H, W, D, tuple(int(v) for v in out_dims), clast, src.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        B, Cc, H, W, D, out_dims, clast, dtype = ctx.meta
        tabs = ctx.tabs
        if Cc % 8 != 0:
            raise NotImplementedError("sample_points backward needs a channel count that is a multiple of 8")
        g = torch.empty((B, H, W, D, Cc), dtype=torch.bfloat16, device=gout.device)
        od = (C.c_int32 * 3)(*out_dims)
        L.call("mivp_sample_points_bwd", L.ptr(gout.contiguous().float()), C.c_int32(B), C.c_int32(H), C.c_int32(W), C.c_int32(D),
               C.c_int32(Cc), od, _ptr3([t[3] for t in tabs], C.c_int32), _ptr3([t[5] for t in tabs], C.c_int32),
               _ptr3([t[4] for t in tabs], C.c_float), _ptr3([t[6] for t in tabs], C.c_float), L.ptr(g), L.stream())
        g = g.permute(0, 4, 1, 2, 3)                            # channels-first view, like the forward's input
        return (g if dtype == torch.bfloat16 else g.float()), None, None, None


def sample_points(vol: torch.Tensor, out_dims: Sequence[int], jitter=None, slot: Optional["JitterSlot"] = None) -> torch.Tensor:
    if not vol.is_cuda:
        raise RuntimeError("mivp_amd.losses runs on the GPU only (the CPU oracle lives in oracle/proto_ref.py)")
    return _SamplePointsFn.apply(vol, tuple(int(v) for v in out_dims), None if jitter is None else tuple(int(v) for v in jitter), slot)


def _pair_dist(cx: torch.Tensor, cy: torch.Tensor) -> torch.Tensor:
    # [B, Nx, 3] x [B, Ny, 3] -> [B, Nx, Ny]; the exact difference form (the matmul form loses the digits that decide the
    # nearest-point argmin and the max_dist threshold)
    return torch.cdist(cx, cy, compute_mode="donot_use_mm_for_euclid_dist")


class ClusteredPrototypeLoss(torch.nn.Module):
    """losses/clustered_prototype_loss.py:13-60.  ``forward`` draws the spatial jitter of each student exactly like the
    reference (``torch.randint(0, ceil(reduction_factor), (6,))`` from the global CPU generator, :173-178); pass ``jitters``
    to fix it (parity tests).  ``detach_teacher`` (default True) treats the teacher-side quantities as constants: the
    reference builds an autograd graph through the teacher forward although no optimizer ever reads the teacher's
    gradients (students_teacher.py:150-207, momentum_model.py:24) -- student gradients and the loss value are identical."""

    def __init__(self, reduction_factor: float = 8.0, k_means_iterations: int = 3, fwhm: float = 128.0,
                 detach_teacher: bool = True, static_jitter: bool = False):
        super().__init__()
        self.reduction_factor = reduction_factor
        self.k_means_iterations = k_means_iterations
        self.fwhm = fwhm
        self.detach_teacher = detach_teacher
        # graph mode (train.GraphedStep): the students' sampling tables live in fixed buffers (JitterSlot)
        self.static_jitter = static_jitter
        self._slots: List[JitterSlot] = []

    def draw_jitters(self, n_students: int):
        """The reference's draw (clustered_prototype_loss.py:173-178): six integers per student from the global CPU generator."""
        return [torch.randint(low=0, high=int(math.ceil(self.reduction_factor)), size=(6,)).tolist() for _ in range(n_students)]

    def load_jitters(self, jitters):
        """Graph mode: refresh the recorded step's sampling tables (call before each replay)."""
        for slot, j in zip(self._slots, jitters):
            slot.reload(j)

    def forward(self, emb_s: List[torch.Tensor], emb_t: torch.Tensor, coord_s: List[torch.Tensor], coord_t: torch.Tensor,
                temp_s: float = 0.066, temp_t: float = 0.033, jitters: Optional[List[Sequence[int]]] = None,
                max_dist: float = 4.0) -> torch.Tensor:
        rf = self.reduction_factor
        sigma2 = (self.fwhm / 2.355) ** 2
        if jitters is None:
            jitters = self.draw_jitters(len(emb_s))
        if self.detach_teacher:
            emb_t = emb_t.detach()
        rs_t = reduced_size(emb_t.shape[2:], rf)
        rs_p = reduced_size(emb_t.shape[2:], rf * 2)
        coord_t = coord_t.float()
        e_p, c_p = sample_points(emb_t, rs_p), sample_points(coord_t, rs_p)          # [B, P, C], [B, P, 3]
        e_t, c_t = sample_points(emb_t, rs_t), sample_points(coord_t, rs_t)
        e_t_n = F.normalize(e_t, dim=-1)
        e_p_n = F.normalize(e_p, dim=-1)

        def soft_assign():
            sim = torch.softmax(e_t_n @ e_p_n.transpose(1, 2) / temp_t, dim=-1)
            return sim * torch.exp(-_pair_dist(c_t, c_p) ** 2 / (2 * sigma2))

        for _ in range(self.k_means_iterations):
            w = soft_assign()
            den = w.sum(dim=1).unsqueeze(-1)
            e_p = (w.transpose(1, 2) @ e_t) / den
            e_p_n = F.normalize(e_p, dim=-1)
            c_p = (w.transpose(1, 2) @ c_t) / den
        sim_t_p = soft_assign()

        total = torch.zeros((), dtype=torch.float32, device=emb_t.device)
        for i in range(len(emb_s)):
            rs_s = reduced_size(emb_s[i].shape[2:], rf)
            slot = None
            if self.static_jitter:
                while len(self._slots) <= i:
                    self._slots.append(JitterSlot())
                slot = self._slots[i]
                slot.load(emb_s[i].shape[2:], rs_s, jitters[i], emb_s[i].device)
            e_z = sample_points(emb_s[i], rs_s, jitters[i], slot)
            c_z = sample_points(coord_s[i].float(), rs_s, jitters[i], slot)
            dmin, idx = _pair_dist(c_z, c_t).min(dim=-1)
            keep = (dmin <= max_dist).float()
            sim = torch.softmax(F.normalize(e_z, dim=-1) @ e_p_n.transpose(1, 2) / temp_s, dim=-1)
            target = torch.gather(sim_t_p, 1, idx.unsqueeze(-1).expand(-1, -1, sim_t_p.shape[-1]))
            logp = torch.clamp(torch.log(sim + 1e-16), min=-1e3, max=-0.0)
            ce = -(target * logp).sum(dim=-1)                   # [B, Ns]
            # mean over the kept points of every batch element (an element without kept points is NaN, as in the reference)
            total = total + ((ce * keep).sum(dim=1) / keep.sum(dim=1)).mean()
        return total


def dice_loss(logits: torch.Tensor, target: torch.Tensor, include_background: bool = True) -> torch.Tensor:
    """MONAI ``DiceLoss(include_background, to_onehot_y=True, softmax=True)`` (students_teacher.py:96-100, used at :190-197):
    per (batch, class) ``1 - (2 sum(p t) + 1e-5) / (sum(p) + sum(t) + 1e-5)``, mean.  Parity unpinned (MONAI absent; restated
    in oracle/loss_ref.py::dice_loss).  On the device this is the Dice-focal statistics / gradient kernel pair without the
    focal term (csrc/loss.hip, ``gamma < 0``): two streaming passes over the logits, no softmax / one-hot tensors."""
    from .train import _DiceFocalFn
    if not logits.is_cuda:
        raise RuntimeError("mivp_amd.losses.dice_loss runs on the GPU (the CPU restatement is oracle/loss_ref.py)")
    if logits.shape[1] > 8:
        raise RuntimeError("dice_loss: at most 8 classes (csrc/loss.hip)")
    base = logits.permute(0, 2, 3, 4, 1)
    if base.dtype != torch.float32 or not base.is_contiguous():
        base = base.float().contiguous()                    # (the HIP model's logits already are channels-last f32 storage)
    tgt = target.to(torch.float32).contiguous()
    return _DiceFocalFn.apply(base, tgt, include_background, -1.0)
