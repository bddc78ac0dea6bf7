"""Checkpoint wire format of the reference's trainers: ``torch.save`` dicts

    {'current_epoch', 'model_state_dict', ['teacher_state_dict'], 'optimizer_state_dict', 'scheduler_state_dict'}

written to ``<dir>/<epoch:04d>.pt`` (segmentation.py:145-154, students_teacher.py:234-244), so that files written by either
code base load into the other (state-dict key names / shapes: SURVEY Appendix D, tests/test_module_surface.py).

Resume semantics follow the trainers (segmentation.py:69-82, students_teacher.py:119-136) with one deliberate fix: the
reference's *backbone* load writes into a COPY of ``state_dict()`` and never calls ``load_state_dict`` (weights silently not
loaded); ``load_backbone`` here really loads the matching entries.  Files are read with
``weights_only=True`` (nothing in a checkpoint is executed)."""
from __future__ import annotations

import os
from typing import Optional

import torch


def checkpoint_path(directory, epoch: int) -> str:
    return os.path.join(str(directory), f"{epoch:04d}.pt")


def save_checkpoint(directory, epoch: int, model, optimizer, scheduler, teacher=None) -> str:
    """``current_epoch`` is ``epoch + 1`` like the reference (the epoch to resume FROM)."""
    os.makedirs(str(directory), exist_ok=True)
    d = {"current_epoch": epoch + 1, "model_state_dict": model.state_dict()}
    if teacher is not None:
        d["teacher_state_dict"] = teacher.state_dict()
    d["optimizer_state_dict"] = optimizer.state_dict()
    d["scheduler_state_dict"] = scheduler.state_dict() if scheduler is not None else {}
    path = checkpoint_path(directory, epoch)
    torch.save(d, path)
    return path


def read_checkpoint(path, map_location="cpu") -> dict:
    return torch.load(path, map_location=map_location, weights_only=True)


def resume(path, model, optimizer=None, scheduler=None, teacher=None, map_location="cpu") -> int:
    """Full resume (segmentation.py:76-82, students_teacher.py:121-130): returns the epoch to continue from."""
    ck = read_checkpoint(path, map_location)
    model.load_state_dict(ck["model_state_dict"])
    if teacher is not None and "teacher_state_dict" in ck:
        teacher.load_state_dict(ck["teacher_state_dict"])
    if optimizer is not None and ck.get("optimizer_state_dict"):
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None and ck.get("scheduler_state_dict"):
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return int(ck["current_epoch"])


def load_backbone(path, model, map_location="cpu") -> int:
    """Load every entry of the checkpoint's ``model_state_dict`` whose name and shape exist in ``model`` (a pre-trained
    backbone into a downstream model: the heads and prompt tokens differ).  Returns the number of tensors loaded."""
    src = read_checkpoint(path, map_location)["model_state_dict"]
    own = model.state_dict()
    picked = {k: v for k, v in src.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
    model.load_state_dict(picked, strict=False)
    return len(picked)
