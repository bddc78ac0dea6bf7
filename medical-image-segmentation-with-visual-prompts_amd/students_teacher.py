"""The students/teacher side of the reference on the device: ``MomentumModel`` (momentum_model/momentum_model.py:4-36)
and the body of ``StudentsTeacherTrainer.train``'s step (students_teacher.py:150-207).

``MomentumModel`` keeps the reference's surface (``tau``, ``net_student``, ``net_teacher``, ``copy_state_dict``,
``forward(x_students, x_teacher)``, ``update_teacher``), so the reference's trainer can use either class; this one updates
the teacher IN PLACE with one fused launch (the reference re-allocates every teacher tensor per step) and runs the teacher
forward without an autograd graph unless ``teacher_graph=True`` (no optimizer reads the teacher's gradients)."""
from __future__ import annotations

from argparse import Namespace
from typing import List

import torch
import torch.nn as nn

from . import optim
from .losses import ClusteredPrototypeLoss, dice_loss


class MomentumModel(nn.Module):
    def __init__(self, conf, architecture, teacher_graph: bool = False):
        super().__init__()
        self.tau = conf.tau
        self.net_student = architecture(conf=conf)
        self.net_teacher = architecture(conf=conf)
        self.teacher_graph = teacher_graph
        self._ema_plan = optim.EmaPlan()

    def copy_state_dict(self):
        for (_, ps), (_, pt) in zip(self.net_student.named_parameters(), self.net_teacher.named_parameters()):
            pt.data.copy_(ps.data)
            pt.requires_grad = False

    def forward(self, x_students, x_teacher):
        out_sts = [self.net_student(x) for x in x_students]
        if self.teacher_graph:
            out_tch = self.net_teacher(x_teacher)
        else:
            with torch.no_grad():
                out_tch = self.net_teacher(x_teacher)
        return out_sts, out_tch

    def update_teacher(self):
        tp = [p for _, p in self.net_teacher.named_parameters()]
        sp = [p for _, p in self.net_student.named_parameters()]
        if tp and tp[0].is_cuda:
            optim.ema_update_(tp, sp, float(self.tau), self._ema_plan)
        else:                                                   # CPU construction / tests of the surface
            with torch.no_grad():
                for t, s in zip(tp, sp):
                    t.data = self.tau * t.data + (1 - self.tau) * s.data


def coord_grid(dims, device=None) -> torch.Tensor:
    """[3, H, W, D] voxel coordinates centred on the volume (datasets/transforms.py:336-344)."""
    axes = [torch.arange(n, dtype=torch.float32, device=device) - (n - 1) / 2.0 for n in dims]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), 0)


def synthetic_views(conf: Namespace, batch: int, size: int, device, rank: int = 0, student_sizes=None):
    """A teacher volume and the students' crops of it with their coordinate grids (the data pipeline's dict keys
    ``image`` / ``coord`` / ``image_st_i`` / ``coord_st_i``, students_teacher.py:151-157), plus integer masks for the
    supervised modes.  Student i is the centred crop of edge ``student_sizes[i]``."""
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    x_t = torch.rand(batch, conf.input_channels, size, size, size, generator=g)
    coord_t = coord_grid((size, size, size))[None].repeat(batch, 1, 1, 1, 1)
    sizes = list(student_sizes) if student_sizes is not None else [size, (size * 3 // 4) // 8 * 8]
    x_s, coord_s = [], []
    for s in sizes:
        o = (size - s) // 2
        sl = (slice(None), slice(None), slice(o, o + s), slice(o, o + s), slice(o, o + s))
        x_s.append((x_t[sl] + 0.05 * torch.rand(batch, conf.input_channels, s, s, s, generator=g)).clamp(0, 1).contiguous())
        coord_s.append(coord_t[sl].contiguous())
    n_cls = conf.output_channels_pretrain
    y0 = torch.randint(0, n_cls, (batch, 1, sizes[0], sizes[0], sizes[0]), generator=g).float()
    to = lambda t: t.to(device)
    return dict(image=to(x_t), coord=to(coord_t), image_st=[to(t) for t in x_s], coord_st=[to(t) for t in coord_s], mask_st_0=to(y0))


def students_teacher_forward_backward(model: MomentumModel, optimizer, loss_prt: ClusteredPrototypeLoss, conf: Namespace,
                                      batch: dict, jitters=None) -> torch.Tensor:
    """students_teacher.py:150-205 up to (not including) the optimizer step: EMA teacher update, students + teacher forward,
    prototype loss (+ Dice on student 0 in the supervised modes with real labels), backward."""
    model.update_teacher()
    out_sts, out_tch = model(batch["image_st"], batch["image"])
    total = torch.zeros((), dtype=torch.float32, device=batch["image"].device)
    if getattr(conf, "use_prototype_assignment", True):
        total = total + loss_prt([o["latent_outputs"] for o in out_sts], out_tch["latent_outputs"], batch["coord_st"],
                                 batch["coord"], jitters=jitters)
    if conf.training_mode in ("supervised_learning_decoder", "supervised_learning_all") and getattr(conf, "use_real_label", True):
        total = total + dice_loss(out_sts[0]["seg_pred"], batch["mask_st_0"], conf.include_background)
    optimizer.zero_grad(set_to_none=True)
    from .train import unit_grad
    total.backward(unit_grad(total))
    return total.detach()


def students_teacher_step(model: MomentumModel, optimizer, scheduler, loss_prt: ClusteredPrototypeLoss, conf: Namespace,
                          batch: dict, jitters=None) -> torch.Tensor:
    """One iteration of students_teacher.py:150-207: EMA teacher update, students + teacher forward, prototype loss
    (+ Dice on student 0 in the supervised modes with real labels), backward, optimizer and per-step scheduler."""
    total = students_teacher_forward_backward(model, optimizer, loss_prt, conf, batch, jitters)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return total


def graphed_students_teacher_step(model: MomentumModel, optimizer, scheduler, loss_prt: ClusteredPrototypeLoss, conf: Namespace,
                                  batch: dict, jitters=None, warmup: int = 2):
    """``students_teacher_step`` recorded in a HIP graph (train.GraphedStep): EMA update, three forwards, prototype loss,
    backward and the optimizer launch replay as one launch; per replay the host draws the students' jitter like the
    reference (``loss_prt.draw_jitters``, or ``jitters()`` if given: a callable returning one list per student), refreshes the
    sampling tables, advances the optimizer's step counts / hyper-parameters and steps the scheduler.  ``batch`` holds the
    fixed input tensors.  ``loss_prt`` must have been built with ``static_jitter=True``."""
    from . import train
    if not loss_prt.static_jitter:
        raise ValueError("graph mode needs ClusteredPrototypeLoss(static_jitter=True)")
    n_st = len(batch["image_st"])
    state = {"j": None}

    def refresh():
        state["j"] = jitters() if jitters is not None else loss_prt.draw_jitters(n_st)
        if loss_prt._slots:                                    # (the first eager warm-up step creates and loads the slots itself)
            loss_prt.load_jitters(state["j"])

    def forward_backward():
        return students_teacher_forward_backward(model, optimizer, loss_prt, conf, batch, jitters=state["j"])

    return train.GraphedStep(forward_backward, optimizer, scheduler, refresh, warmup)
