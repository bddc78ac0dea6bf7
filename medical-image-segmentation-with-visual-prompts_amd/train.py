"""Training-step harness for the HIP SwinUnetR: the step body of the reference's
``SegmentationTrainer.train`` (modules/segmentation.py:96-122) on synthetic data, single GPU or one
process per GPU under ``torch.distributed`` (RCCL).

Loss and optimizer: the fused Dice+focal kernel (csrc/loss.hip; DiceFocal restated from MONAI's
documented formulas -- parity unpinned, see oracle/loss_ref.py), ``optim.FusedAdamW`` (one multi-tensor launch per step),
and for the ``self_supervised_*`` modes the students/teacher step of students_teacher.py:150-207
(``mivp_amd.students_teacher``: EMA teacher, two students + teacher forward, ClusteredPrototypeLoss, per-step schedule).
"""
from __future__ import annotations

import os
from argparse import Namespace
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the headline metric is quoted on
    "cfg1": dict(training_mode="downstream", use_encoder_prompting=False, use_decoder_prompting=False,
                 input_channels=1, size=96, batch=4),
    # configs[2] (per-GPU shape): + encoder prompting
    "cfg2": dict(training_mode="downstream", use_encoder_prompting=True, use_decoder_prompting=False,
                 input_channels=1, size=96, batch=4),
    # configs[3] (per-GPU shape): both prompt sides, 4-channel 128^3
    "cfg3": dict(training_mode="downstream", use_encoder_prompting=True, use_decoder_prompting=True,
                 input_channels=4, size=128, batch=1),
    # configs[4] (per-GPU shape): prompt-token-only fine-tune of the frozen backbone at batch 8.  The config asks for fp8
    # window attention; the attention kernels are exp/VALU-bound, so they stay bf16 here
    "cfg4": dict(training_mode="downstream", use_encoder_prompting=True, use_decoder_prompting=False,
                 input_channels=1, size=96, batch=8),
    # configs[0]: the reference's CPU-runnable plumbing case (all parameters train)
    "cfg0": dict(training_mode="self_supervised_learning_all", use_encoder_prompting=False, use_decoder_prompting=False,
                 input_channels=1, size=32, batch=2),
    # every parameter trains on the segmentation objective (students_teacher.py:25-68,190-197), 96^3
    "sup_all": dict(training_mode="supervised_learning_all", use_encoder_prompting=False, use_decoder_prompting=False,
                    input_channels=1, size=96, batch=4),
    # smoke-sized
    "tiny": dict(training_mode="downstream", use_encoder_prompting=True, use_decoder_prompting=True,
                 input_channels=1, size=32, batch=2),
}


def make_conf(workload: str, window=(7, 7, 7), dropout: float = 0.0) -> Tuple[Namespace, int, int]:
    """yml defaults of configurations/example_configs.yml with the north star's 7x7x7 window; ``dropout`` sets
    attn_drop = proj_drop (0 for parity runs, the yml's 0.1 for throughput: SURVEY 8d); returns
    (conf, volume size, per-GPU batch)."""
    w = dict(WORKLOADS[workload])
    size, batch = w.pop("size"), w.pop("batch")
    conf = Namespace(
        depth_unet=3, hidden_channels=[48, 96, 192, 384], input_patch_size=[2, 2, 2], unetr_res_block="none",
        unetr_up_block="swin", basic_block_res=True, num_heads_encoder=4, num_heads_decoder=4,
        attn_window_size=list(window), pos_bias_embed_dim=64, use_checkpoint=False, attn_drop=float(dropout),
        proj_drop=float(dropout),
        max_prompts=1, tokens_per_prompt_encoder=64, tokens_per_prompt_decoder=64,
        use_reconstruction=False, use_mutual_learning=False, use_rotation_prediction=False,
        use_contrastive_learning=False, contrastive_coding_dim=512, output_channels_downstream=2,
        output_channels_pretrain=5, include_background=True, lr_downstream=1e-3, weight_decay_downstream=0.0,
        lr_students_teacher=5e-4, weight_decay_students_teacher=0.1, lr_prompt_tokens=5e-4,
        weight_decay_prompt_tokens=0.1,
        # students-teacher trainer (example_configs.yml:84-99)
        tau=0.99, reduction_factor=4, fwhm=128, k_means_iterations=3, use_prototype_assignment=True, use_real_label=True,
        warmup_steps_students_teacher=100, t_total_students_teacher=2400, **w)
    return conf, size, batch


def synthetic_batch(conf: Namespace, batch: int, size: int, device, rank: int = 0):
    """Images uniform in [0,1) like the reference's ScaleIntensityRanged output (datasets/transforms.py:41-44),
    integer masks in [0, out_ch) stored as float like its label volumes (modules/utils.py:372-388)."""
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.rand(batch, conf.input_channels, size, size, size, generator=g)
    n_cls = conf.output_channels_downstream if conf.training_mode == "downstream" else conf.output_channels_pretrain
    y = torch.randint(0, n_cls, (batch, 1, size, size, size), generator=g).float()
    return x.to(device), y.to(device)


class _DiceFocalFn(torch.autograd.Function):
    """Loss value from one streaming pass + a tiny reduction; d loss / d logits from a second streaming pass when backward
    asks for it, already multiplied by the incoming gradient (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, logits_cl, target, include_background, gamma):
        import ctypes as C
        from . import _lib as L
        B, H, W, D, Cc = logits_cl.shape
        vol = H * W * D
        ws = torch.empty(L.lib().mivp_dice_focal_ws(C.c_int32(B), C.c_int64(vol)), dtype=torch.float32, device=logits_cl.device)
        loss = torch.empty(1, dtype=torch.float32, device=logits_cl.device)
        L.call("mivp_dice_focal", L.ptr(logits_cl), L.ptr(target), C.c_int32(B), C.c_int64(vol), C.c_int32(Cc),
               C.c_int32(1 if include_background else 0), C.c_float(gamma), L.ptr(ws), L.ptr(loss), L.ptr(None), L.stream())
        ctx.save_for_backward(logits_cl, target, ws)
        ctx.meta = (B, vol, Cc, include_background, gamma)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from . import _lib as L
        logits_cl, target, ws = ctx.saved_tensors
        B, vol, Cc, include_background, gamma = ctx.meta
        g = g.detach().to(torch.float32).contiguous()
        dz = torch.empty_like(logits_cl)
        L.call("mivp_dice_focal_grad", L.ptr(logits_cl), L.ptr(target), C.c_int32(B), C.c_int64(vol), C.c_int32(Cc),
               C.c_int32(1 if include_background else 0), C.c_float(gamma), L.ptr(ws), L.ptr(g), L.ptr(dz), L.stream())
        return dz, None, None, None


def dice_focal_loss(logits: torch.Tensor, target: torch.Tensor, include_background: bool = True,
                    gamma: float = 4.0) -> torch.Tensor:
    """MONAI ``DiceFocalLoss(include_background, to_onehot_y=True, softmax=True, gamma=4)`` as documented
    (segmentation.py:44-50): Dice on softmax probabilities (smooth 1e-5) + sigmoid focal loss, both
    averaged.  logits [B,C,H,W,D] float, target [B,1,H,W,D] float class indices.

    When the logits are the channels-first VIEW the HIP model returns (fp32, channels-last storage) the loss
    and its gradient come from the fused HIP kernel; any other layout takes the plain PyTorch formula."""
    base = logits.permute(0, 2, 3, 4, 1)
    if logits.is_cuda and logits.dtype == torch.float32 and base.is_contiguous() and logits.shape[1] <= 8 \
            and target.is_contiguous() and target.dtype == torch.float32:
        return _DiceFocalFn.apply(base, target, include_background, gamma)
    C = logits.shape[1]
    onehot = F.one_hot(target[:, 0].long(), C).permute(0, 4, 1, 2, 3).to(logits.dtype)
    prob = logits.softmax(dim=1)
    x = logits
    if not include_background:
        onehot, prob, x = onehot[:, 1:], prob[:, 1:], logits[:, 1:]
    dims = (2, 3, 4)
    inter = (prob * onehot).sum(dims)
    denom = prob.sum(dims) + onehot.sum(dims)
    dice = (1.0 - (2.0 * inter + 1e-5) / (denom + 1e-5)).mean()
    bce = x - x * onehot - F.logsigmoid(x)
    inv = F.logsigmoid(-x * (onehot * 2 - 1))
    focal = ((inv * gamma).exp() * bce).mean()
    return dice + focal


def build_optimizer(model, conf: Namespace, capturable: bool = False):
    """AdamW over the reference's parameter partition for the mode: ``named_parameters_downstream()`` for
    ``downstream`` (segmentation.py:25-39); decoder(+encoder) and prompt-token groups with their own lr /
    weight decay for the ``*_all`` / ``*_decoder`` modes (students_teacher.py:25-68)."""
    from .optim import FusedAdamW
    core = model.module if hasattr(model, "module") else model
    core = core.net_student if hasattr(core, "net_student") else core      # MomentumModel: the student trains
    mode = conf.training_mode
    if mode == "downstream":
        params = [p for _, p in core.named_parameters_downstream()]
        return FusedAdamW(params, lr=float(conf.lr_downstream), weight_decay=float(conf.weight_decay_downstream), capturable=capturable)
    lr, wd = float(conf.lr_students_teacher), float(conf.weight_decay_students_teacher)
    groups = []
    if mode in ("self_supervised_learning_all", "supervised_learning_all"):
        groups.append({"params": [p for _, p in core.named_parameters_decoder()] + [p for _, p in core.named_parameters_encoder()],
                       "lr": lr, "weight_decay": wd})
        if conf.use_encoder_prompting:
            groups.append({"params": [p for _, p in core.named_parameters_prompt_tokens_encoder()],
                           "lr": float(conf.lr_prompt_tokens), "weight_decay": float(conf.weight_decay_prompt_tokens)})
    elif mode in ("self_supervised_learning_decoder", "supervised_learning_decoder"):
        groups.append({"params": [p for _, p in core.named_parameters_decoder()], "lr": lr, "weight_decay": wd})
    else:
        raise ValueError(f"no optimizer recipe for training mode {mode!r}")
    if conf.use_decoder_prompting:
        groups.append({"params": [p for _, p in core.named_parameters_prompt_tokens_decoder()],
                       "lr": float(conf.lr_prompt_tokens), "weight_decay": float(conf.weight_decay_prompt_tokens)})
    return FusedAdamW(groups, lr=lr, weight_decay=wd, capturable=capturable)


def build_scheduler(opt, conf: Namespace):
    """Per-step WarmupCosineSchedule of the students/teacher trainer (students_teacher.py:69-76,207); the segmentation
    trainer's per-EPOCH StepLR(100, 0.8) (segmentation.py:34-38) never fires inside a benchmark run."""
    from .optim import WarmupCosineSchedule
    if conf.training_mode == "downstream":
        return None
    return WarmupCosineSchedule(opt, warmup_steps=int(conf.warmup_steps_students_teacher), t_total=int(conf.t_total_students_teacher))


def step_loss(out: dict, conf: Namespace, y) -> torch.Tensor:
    """The objective of a SINGLE-NETWORK step.  ``downstream``: DiceFocal on out['downstream'] (segmentation.py:44-50,
    104-106).  ``supervised_*``: the trainer's segmentation term, ``DiceLoss`` on out['seg_pred'] (students_teacher.py:96-100,
    190-197) -- the ``sup_all`` throughput workload; the full supervised step with its prototype term is
    ``students_teacher.students_teacher_step``.  The ``self_supervised_*`` modes have no single-network objective: their
    step is the students/teacher step."""
    mode = conf.training_mode
    if mode == "downstream":
        return dice_focal_loss(out["downstream"], y, conf.include_background, 4.0)
    if mode in ("supervised_learning_all", "supervised_learning_decoder"):
        from .losses import dice_loss
        return dice_loss(out["seg_pred"], y, conf.include_background)
    raise ValueError(f"{mode}: use mivp_amd.students_teacher.students_teacher_step (students_teacher.py:150-207)")


_unit_grads = {}


def unit_grad(loss: torch.Tensor) -> torch.Tensor:
    """The constant 1 that starts ``backward`` (autograd otherwise fills a new ``ones_like(loss)`` every step: a launch)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    if key not in _unit_grads:
        _unit_grads[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    return _unit_grads[key]


def train_step(model, opt, conf: Namespace, x, y) -> torch.Tensor:
    out = model(x)
    loss = step_loss(out, conf, y)
    opt.zero_grad(set_to_none=True)
    loss.backward(unit_grad(loss))
    opt.step()
    return loss.detach()


class GraphedStep:
    """One whole training step -- forward, loss, backward, optimizer launch -- recorded ONCE in HIP graphs and replayed.

    An eager step is ~85 (cfg1) to ~1700 (cfg0: two students + teacher on 32^3 volumes) launches issued from Python at
    10-15 us each; where the kernels are shorter than that (cfg0) -- or where eight Python processes share a host
    (``bench.py --gpus 8``) -- the step is bound by the host.  A replay is one launch.

    ``forward_backward()`` runs on the current stream and returns the loss tensor: model forward, loss,
    ``optimizer.zero_grad(set_to_none=True)``, ``backward()`` -- NOT the optimizer step (this class calls it), no scheduler
    (host arithmetic on ``param_groups``: it runs after every replay here), no ``.item()``.  Its inputs are fixed tensors: feed
    a new batch by copying into them (``x.copy_(new)``) before the call.  What changes per step on the host side travels
    through device memory: the optimizer's lr / bias corrections (``FusedAdamW(capturable=True).advance()``), whatever
    ``refresh()`` loads (the prototype loss's jitter tables), and the dropout masks: the recording starts by incrementing
    the device's dropout epoch word, which the kernels fold into their seeds (functional.dropout_epoch, mivp.h
    ``seed_epoch``), so every replay draws fresh attention / projection masks.

    Data parallel (``torch.distributed`` initialised, world size > 1): the model is NOT wrapped in DistributedDataParallel.
    Forward + backward are one graph that ends by gathering the gradients into ONE flat fp32 bucket; the bucket is
    all-reduced (mean) eagerly on the same stream -- one collective per step: 0.57 MB in ``downstream`` mode, 37.5 MB in the
    ``*_all`` modes; RCCL over xGMI on the GPU box, gloo in the tests -- and a second graph holds the optimizer launch,
    which reads its gradients from the bucket.  Parameters are broadcast from rank 0 once at construction; BatchNorm
    statistics stay per replica like the single-device reference.

    After a replay the packed-weight caches are marked stale, so an eager forward / evaluation between replays sees the
    current parameters."""

    def __init__(self, forward_backward, optimizer, scheduler=None, refresh=None, warmup: int = 2):
        import torch.distributed as dist
        from . import functional as Fn
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedStep needs FusedAdamW(capturable=True) (train.build_optimizer(..., capturable=True))")
        self._fn = Fn
        self.optimizer, self.scheduler, self.refresh = optimizer, scheduler, refresh
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if os.environ.get("MIVP_GRAPH_FORCE_SINGLE"):             # (debug: the single-graph path under an initialised process group)
            self.world = 1
        params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        self.params = params
        dev = params[0].device
        self._epoch = Fn.dropout_epoch(dev)
        if self.world > 1:                                       # identical replicas to start from (DDP does this at construction)
            flat = torch.cat([p.detach().reshape(-1) for p in params])
            dist.broadcast(flat, src=0)
            off = 0
            with torch.no_grad():
                for p in params:
                    p.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()
            Fn.invalidate_weight_caches()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                      # eager: optimizer state, tables and caches come to life here
                if refresh is not None:
                    refresh()
                forward_backward()
                if self.world > 1:
                    self._eager_mean_grads()
                optimizer.step()
                if scheduler is not None:
                    scheduler.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer.zero_grad(set_to_none=True)                    # the recorded backward allocates the gradients in the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        if self.world == 1:
            with torch.cuda.graph(self.graph):
                self._epoch.add_(1)
                self.loss = forward_backward()
                optimizer.step()
        else:
            self.flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
            views, off = [], 0
            for p in params:
                views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            with torch.cuda.graph(self.graph):
                self._epoch.add_(1)
                self.loss = forward_backward()
                missing = [i for i, p in enumerate(params) if p.grad is None]
                if missing:
                    raise RuntimeError(f"GraphedStep (data parallel): {len(missing)} trainable parameters received no gradient")
                self.local_grads = [p.grad for p in params]      # this replica's own gradients of the last replay (tests)
                torch._foreach_copy_(views, self.local_grads)
            for p, v in zip(params, views):                      # the optimizer launch reads the reduced gradients in place
                p.grad = v
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool()):
                optimizer.step()
        self.launches_eager = None
        self.host_seconds = self.collective_seconds = 0.0
        self.replays = 0

    def _eager_mean_grads(self):
        import torch.distributed as dist
        grads = [p.grad for p in self.params if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self._allreduce_mean(flat)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def _allreduce_mean(self, flat):
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:                                                    # gloo has no AVG
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / self.world)

    def __call__(self) -> torch.Tensor:
        import time
        t0 = time.perf_counter()
        if self.refresh is not None:
            self.refresh()
        self.optimizer.advance()
        self.graph.replay()
        if self.graph_opt is not None:
            t1 = time.perf_counter()
            self._allreduce_mean(self.flat)                      # (gloo stages through the host and waits for the device)
            self.collective_seconds += time.perf_counter() - t1
            self.graph_opt.replay()
        self._fn.invalidate_weight_caches()
        if self.scheduler is not None:
            self.scheduler.step()
        self.host_seconds += time.perf_counter() - t0
        self.replays += 1
        return self.loss

    def host_ms_per_replay(self) -> float:
        """Host time of a replayed step outside the collective call (table refresh, two graph launches, scheduler)."""
        return 1e3 * (self.host_seconds - self.collective_seconds) / max(1, self.replays)


def graphed_train_step(model, opt, conf: Namespace, x, y, warmup: int = 2) -> GraphedStep:
    """``train_step`` as recorded graphs: call the result with no arguments; ``x`` / ``y`` are its fixed input tensors.  Under
    ``torch.distributed`` pass the BARE model (GraphedStep synchronises the gradients itself, see there)."""
    if hasattr(model, "module"):
        raise ValueError("GraphedStep takes the bare model: it all-reduces one flat gradient bucket between its two graphs "
                         "instead of DistributedDataParallel's hooks")

    def forward_backward():
        out = model(x)
        loss = step_loss(out, conf, y)
        opt.zero_grad(set_to_none=True)
        loss.backward(unit_grad(loss))
        return loss.detach()

    return GraphedStep(forward_backward, opt, None, None, warmup)


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local, world


def init_distributed(device=None, backend: Optional[str] = None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  ``nccl`` is RCCL on ROCm;
    ``gloo`` is used by the CPU tests of this plumbing."""
    import torch.distributed as dist
    rank, local, world = dist_env()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if device is not None and torch.device(device).type == "cuda" else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend, device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def barrier_sync(device=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize()


def max_over_ranks(seconds: float, device=None) -> float:
    """Step time of the slowest rank (what the whole job waits for)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def wrap_ddp(model, local_rank: Optional[int], gloo: bool = False):
    """One process per GPU, gradient all-reduce (mean) overlapped with backward through DDP buckets
    (RCCL over xGMI on the GPU box).  BatchNorm statistics stay per replica as in the single-device
    reference: no buffer broadcast, no SyncBN.  ``local_rank=None`` wraps a CPU module (gloo tests);
    ``gloo=True``: a device module over a gloo group (several ranks may share one GPU: device_ids must stay unset)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if local_rank is None or gloo:
        return DDP(model, broadcast_buffers=False, gradient_as_bucket_view=True)
    return DDP(model, device_ids=[local_rank], output_device=local_rank, broadcast_buffers=False,
               gradient_as_bucket_view=True)
