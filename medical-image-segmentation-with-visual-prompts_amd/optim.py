"""Optimizer side of the reference's trainers on the device.

* ``FusedAdamW``            -- ``torch.optim.AdamW`` semantics (decoupled weight decay, bias correction, per-group lr / weight
  decay: students_teacher.py:27-68, segmentation.py:25-39) with ONE kernel launch per step for all parameter tensors
  (csrc/proto.hip ``k_adamw_multi``).  It is a ``torch.optim.Optimizer``: ``param_groups`` / ``state_dict`` have AdamW's
  layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter), so LR schedulers and the reference's checkpoint dicts
  (``optimizer_state_dict``) interoperate with ``torch.optim.AdamW``.
* ``WarmupCosineSchedule``  -- modules/utils.py:67-89 (a ``LambdaLR``).
* ``ema_update_``           -- the teacher's EMA (momentum_model/momentum_model.py:27-36) for all parameters in one launch.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List

import numpy as np
import torch
from torch.optim.lr_scheduler import LambdaLR

from . import _lib as L

_CHUNK = 1024


def _chunk_table(sizes: List[int], device, with_begin: bool = False):
    ids, offs, begin = [], [], [0]
    for t, n in enumerate(sizes):
        k = (n + _CHUNK - 1) // _CHUNK
        ids.append(np.full(k, t, np.int32))
        offs.append(np.arange(k, dtype=np.int32))
        begin.append(begin[-1] + k)
    tab = np.stack([np.concatenate(ids), np.concatenate(offs)], 1).astype(np.int32)
    dev_tab = torch.from_numpy(np.ascontiguousarray(tab)).to(device)
    return (dev_tab, np.asarray(begin, np.int32)) if with_begin else dev_tab


class FusedAdamW(torch.optim.Optimizer):
    """``capturable=True``: the launch reads lr / weight decay / bias corrections from a device buffer that ``step()``
    refreshes with one tiny stream-ordered launch (``mivp_store_floats``) -- the form ``train.GraphedStep`` records.  While
    the stream is being captured ``step()`` only records the update launch; every replay is preceded by ``advance()``,
    which does the host side of a step (step counts, this step's hyper-parameters into the device buffer)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) > 8:
            raise ValueError("FusedAdamW supports up to 8 parameter groups")
        self._plan = None
        self._steps = {}          # id(param) -> step count as a Python int (the state's ``step`` tensors are synced lazily)
        self.capturable = bool(capturable)
        self._hyper_dev = None
        self._graph_params = None  # the parameters that had gradients when a graph was recorded (they step on every replay)

    def state_dict(self):
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) in self._steps and p in self.state:
                    self.state[p]["step"] = torch.tensor(float(self._steps[id(p)]), dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._steps = {id(p): int(float(st["step"])) for p, st in self.state.items() if "step" in st}
        self._plan = None
        self._hyper_fast = None

    def _build(self, entries):
        dev = entries[0][1].device
        rows = np.zeros((len(entries), 5), np.int64)
        for i, (gi, p, st) in enumerate(entries):
            rows[i] = (p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), gi)
        if int(L.lib().mivp_sizeof_opt(0)) != 40:
            raise RuntimeError("AdamTensor layout mismatch")
        chunks, begin = _chunk_table([p.numel() for _, p, _ in entries], dev, with_begin=True)
        self._plan = {
            "key": tuple((id(p), p.data_ptr(), st["exp_avg"].data_ptr()) for _, p, st in entries),
            "tensors": torch.from_numpy(rows).to(dev), "chunks": chunks, "begin": begin,
        }

    def _hyper(self, stepping):
        """Advance the step counts of ``stepping`` (ids of the parameters that step now) and return this step's
        [groups][8] table: lr, beta1, beta2, eps, weight_decay, 1 - beta1^t, sqrt(1 - beta2^t), 0."""
        hyper = np.zeros((len(self.param_groups), 8), np.float32)
        # per group: the ids that step and their common count, kept between calls while the stepping set is the same (a
        # replayed step must not walk ~230 parameters in Python: that was 0.3 ms of host time per step)
        key = (id(stepping) if isinstance(stepping, frozenset) else None, len(stepping))
        fast = getattr(self, "_hyper_fast", None)
        if fast is None or fast[0] != key or key[0] is None:
            groups = []
            for gi, group in enumerate(self.param_groups):
                ids = [id(p) for p in group["params"] if id(p) in stepping]
                counts = {self._steps.get(i, 0) for i in ids}
                if len(counts) > 1:
                    # torch.optim.AdamW corrects the bias per PARAMETER; the kernel's table holds one pair per group
                    raise RuntimeError(
                        f"FusedAdamW: parameters of group {gi} that step together have different step counts ({sorted(counts)}: a "
                        "parameter that got its first gradient later, or a resumed state with mixed counts); put them in "
                        "separate groups")
                groups.append([ids, counts.pop() if counts else 0])
            fast = (key, groups)
            self._hyper_fast = fast if key[0] is not None else None
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            ids, n = fast[1][gi]
            step_no = 1.0
            if ids:
                n += 1
                fast[1][gi][1] = n
                self._steps.update(dict.fromkeys(ids, n))
                step_no = float(n)
            # all parameters of a group that receive gradients step together (as in the reference's trainers)
            hyper[gi] = (group["lr"], b1, b2, group["eps"], group["weight_decay"], 1.0 - b1 ** step_no,
                         math.sqrt(1.0 - b2 ** step_no), 0.0)
        return hyper

    def _upload(self, hyper, device):
        if self._hyper_dev is None or self._hyper_dev.device != device:
            self._hyper_dev = torch.zeros(64, dtype=torch.float32, device=device)
        flat = np.ascontiguousarray(hyper.reshape(-1))
        L.call("mivp_store_floats", L.ptr(self._hyper_dev), flat.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(flat.size), L.stream())

    @torch.no_grad()
    def advance(self):
        """Host side of one REPLAYED step (the recorded graph holds the update launch itself)."""
        if self._graph_params is None:
            raise RuntimeError("FusedAdamW.advance(): no step has been recorded in a graph")
        dev = self._hyper_dev.device
        self._upload(self._hyper(self._graph_params), dev)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("FusedAdamW: dense float32 device parameters only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                entries.append((gi, p, st))
        if not entries:
            return loss
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and not self.capturable:
            raise RuntimeError("FusedAdamW: build the optimizer with capturable=True to record its step in a graph")
        stepping = {id(p) for _, p, _ in entries}
        if capturing:
            # the recording is not a step: counts and the device table are advanced by advance() before each replay
            hyper = None
            self._graph_params = frozenset(stepping)
            if self._hyper_dev is None:
                raise RuntimeError("FusedAdamW: run at least one eager step before recording (state and tables are built there)")
        else:
            hyper = self._hyper(stepping)
        key = tuple((id(p), p.data_ptr(), st["exp_avg"].data_ptr()) for _, p, st in entries)
        if self._plan is None or self._plan["key"] != key:
            if capturing:
                raise RuntimeError("FusedAdamW: the parameter set changed between the eager warm-up and the recording")
            self._build(entries)
        plan = self._plan
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for _, p, _ in entries]
        for (_, p, _), g in zip(entries, grads):
            if not p.is_contiguous():
                raise RuntimeError("FusedAdamW: contiguous parameters only")
        # the gradient tensors are new every backward (zero_grad(set_to_none=True)): their pointers go to the kernel as
        # arguments (a host array here) -- an upload per step would be a synchronous pageable copy that stalls the launch queue
        gptr = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        if self.capturable:
            if hyper is not None:
                self._upload(hyper, entries[0][1].device)
            L.call("mivp_adamw_multi_dev", L.ptr(plan["tensors"]), gptr, C.c_int32(len(grads)),
                   plan["begin"].ctypes.data_as(C.POINTER(C.c_int32)), L.ptr(self._hyper_dev),
                   C.c_int32(len(self.param_groups)), L.ptr(plan["chunks"]), L.stream())
        else:
            L.call("mivp_adamw_multi", L.ptr(plan["tensors"]), gptr, C.c_int32(len(grads)),
                   plan["begin"].ctypes.data_as(C.POINTER(C.c_int32)), hyper.ctypes.data_as(C.POINTER(C.c_float)),
                   C.c_int32(len(self.param_groups)), L.ptr(plan["chunks"]), L.stream())
        return loss


class WarmupCosineSchedule(LambdaLR):
    """modules/utils.py:67-89: linear warm-up over ``warmup_steps`` scheduler steps, then a cosine to zero at ``t_total``."""

    def __init__(self, optimizer, warmup_steps: int, t_total: int, cycles: float = 0.5, last_epoch: int = -1):
        self.warmup_steps = warmup_steps
        self.t_total = t_total
        self.cycles = cycles
        super().__init__(optimizer, self.lr_lambda, last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1.0, self.warmup_steps))
        progress = float(step - self.warmup_steps) / float(max(1, self.t_total - self.warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(self.cycles) * 2.0 * progress)))


class EmaPlan:
    """Pointer tables of one (teacher, student) parameter pairing, rebuilt when a storage moves."""

    def __init__(self):
        self.key = None
        self.tensors = None
        self.chunks = None


@torch.no_grad()
def ema_update_(teacher_params, student_params, tau: float, plan: EmaPlan):
    """teacher = tau * teacher + (1 - tau) * student for every pair, in place, one launch (momentum_model.py:27-36)."""
    from . import functional as Fn
    pairs = [(t, s) for t, s in zip(teacher_params, student_params)]
    if not pairs:
        return
    key = tuple((t.data_ptr(), s.data_ptr(), t.numel()) for t, s in pairs)
    if plan.key != key:
        for t, s in pairs:
            if t.dtype != torch.float32 or s.dtype != torch.float32 or not t.is_cuda or t.shape != s.shape \
                    or not t.is_contiguous() or not s.is_contiguous():
                raise RuntimeError("ema_update_: contiguous float32 device parameter pairs of equal shape only")
        rows = np.asarray([(t.data_ptr(), s.data_ptr(), t.numel()) for t, s in pairs], np.int64)
        dev = pairs[0][0].device
        plan.tensors = torch.from_numpy(rows).to(dev)
        plan.chunks = _chunk_table([t.numel() for t, _ in pairs], dev)
        plan.key = key
    L.call("mivp_ema_multi", L.ptr(plan.tensors), L.ptr(plan.chunks), C.c_int32(plan.chunks.shape[0]), C.c_float(tau), L.stream())
    Fn.invalidate_weight_caches()                              # raw in-place writes: no version counter saw them
