#!/usr/bin/env python3
"""Headline benchmark: 3D volumes/sec of one Swin-UNETR training step (forward + DiceFocal loss +
backward + AdamW) on synthetic 96^3 volumes, bf16 activations / MFMA operands, on N MI355X GPUs.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU, data parallel over RCCL (weak scaling: per-GPU batch fixed).  Rank 0 prints ONE
JSON line.  Workload = BASELINE.json configs[1] (downstream mode, 1-channel 96^3, batch 4 per GPU,
window 7x7x7) unless --workload says otherwise.  Inputs are resident in HBM before the timed region.

Extra objects on the line:
  roofline            the DOMINANT kernel family of the step: the window-attention forward of the stage-0 blocks (the kernel
                      BASELINE.json's north star sets its MFMA target on).  achieved = algorithmic FLOPs 4 Nq Nk hd per
                      (window, head) / mean launch duration, measured with HIP events on the launch stream inside the timed
                      region; peak = dense bf16 MFMA; beside `frac` the fractions of the kernel's real bounds: the
                      transcendental issue rate (one v_exp_f32 per (query, key) pair) and the whole vector-issue budget of a
                      32-key step (8 v_exp + 4 v_cvt_pk + 3 MFMA issue slots; costs measured by tools/ubench/valu_rates.hip and
                      mfma_overlap.hip); traffic = fabric-side bytes per launch from the committed PMC passes
                      (profiles/r03_traffic.json).
  roofline_conv       the largest MFMA-shaped kernel (3x3x3 halo-brick conv of the last decoder stage, 144->48 channels at
                      48^3) against the dense bf16 MFMA peak (this was `roofline` in rounds 1-2).
  cpu_baseline        the CPU oracle (oracle/, "port") timed on this host's cores on a bounded sample (2 warm-up + 5 timed
                      training steps as BASELINE.md section 3 asks, best and median reported; batch 4 for the prompt-free
                      workloads, batch 1 otherwise), rank 0 at N=1 only.
--backend gloo runs the same multi-rank control flow over gloo (ranks may then share one GPU: tests/test_hip_ddp.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg1", choices=["cfg0", "cfg1", "cfg2", "cfg3", "cfg4", "sup_all", "tiny"])
    ap.add_argument("--window", default="7,7,7")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0 = the workload's)")
    ap.add_argument("--dropout", type=float, default=0.0, help="attn_drop = proj_drop (the yml default is 0.1)")
    ap.add_argument("--settle", type=float, default=1.0, help="seconds of untimed steps before the warm-up (>= 10 steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="record the step in HIP graphs and time replays (train.GraphedStep: with --gpus N one flat gradient "
                         "all-reduce runs between the forward+backward graph and the optimizer graph; dropout through the device "
                         "epoch word); the roofline kernels are then timed in a few eager steps AFTER the timed region")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL; gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--device", type=int, default=-1, help="GPU index (default: LOCAL_RANK)")
    ap.add_argument("--fp8-attn", action="store_true",
                    help="BASELINE.json configs[4]'s named arithmetic: E4M3 MFMA window-attention forward in the head_dim < 16 blocks "
                         "(an experiment that measured slower and less accurate than bf16: off by default, DESIGN.md section 8)")
    return ap.parse_args()


def conv_flops(desc):
    vox = desc.B * desc.dims[0] * desc.dims[1] * desc.dims[2]
    return 2.0 * 27 * desc.Cin * desc.Cout * vox


def traffic_bytes(key):
    """Fabric-side bytes per launch of a roofline kernel from the committed PMC passes of THIS round's build
    (profiles/r03_traffic.json, tools/pmc_traffic.sh: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate
    rocprofv3 --pmc runs); None if the file or the entry is absent."""
    path = os.path.join(ROOT, "profiles", "r03_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)[key]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(conf, size):
    """The oracle's training step (forward + loss + backward + AdamW) on all usable host cores: two warm-up steps, then five
    timed steps, best and median reported (~11 s per step on the GPU box's 16 cores: ~80 s)."""
    from oracle.unetr_ref import OracleSwinUnetR, random_state
    from oracle.loss_ref import dice_focal_loss
    from oracle import proto_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    mode = conf.training_mode
    sd = random_state(conf, seed=0)
    model = OracleSwinUnetR(conf, sd)
    keys = model.trainable_keys()
    for k in keys:
        sd[k].requires_grad_(True)
    opt = torch.optim.AdamW([sd[k] for k in keys], lr=1e-3, weight_decay=0.0)
    g = torch.Generator().manual_seed(1234)
    ssl = mode.startswith("self_supervised")
    nvol = 2 if ssl else (4 if not (conf.use_encoder_prompting or conf.use_decoder_prompting) else 1)   # ~10-30 s of CPU work
    x = torch.rand(nvol, conf.input_channels, size, size, size, generator=g)
    n_cls = conf.output_channels_downstream if mode == "downstream" else conf.output_channels_pretrain
    y = torch.randint(0, n_cls, (nvol, 1, size, size, size), generator=g).float()
    if ssl:                                            # the students/teacher step (students_teacher.py:150-207)
        s1 = (size * 3 // 4) // 8 * 8
        o = (size - s1) // 2
        coord_t = proto_ref.coord_grid((size,) * 3)[None].repeat(nvol, 1, 1, 1, 1)
        crop = (slice(None), slice(None), slice(o, o + s1), slice(o, o + s1), slice(o, o + s1))
        teacher_sd = {k: v.detach().clone() for k, v in sd.items()}

    def step():
        if ssl:
            for k in keys:
                teacher_sd[k] = proto_ref.ema_update(teacher_sd[k], sd[k].detach(), float(conf.tau))
            outs = [model(x, training=True)[0]["latent_outputs"], model(x[crop].contiguous(), training=True)[0]["latent_outputs"]]
            with torch.no_grad():
                out_t = OracleSwinUnetR(conf, teacher_sd)(x, training=True)[0]["latent_outputs"]
            loss = proto_ref.clustered_prototype_loss(outs, out_t, [coord_t, coord_t[crop]], coord_t, [[0] * 6, [1, 0, 2, 1, 0, 3]],
                                                      float(conf.reduction_factor), int(conf.k_means_iterations), float(conf.fwhm))
        else:
            out, _ = model(x, training=True)
            loss = dice_focal_loss(out["downstream" if mode == "downstream" else "seg_pred"], y, conf.include_background)
        opt.zero_grad()
        loss.backward()
        opt.step()

    n_warm, n_timed = 2, 5
    for _ in range(n_warm):                            # warm-up (allocator, thread pool)
        step()
    times = []
    for _ in range(n_timed):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    dt = min(times)
    med = sorted(times)[len(times) // 2]
    what = "students/teacher step (2 students + teacher, prototype loss)" if ssl else "training step"
    return {"value": nvol / dt, "unit": "volumes/s", "cores": cores, "kind": "port", "median_value": nvol / med,
            "sample": f"{n_warm} warm-up + {n_timed} timed {what}s on a batch of {nvol} volume(s) of {size}^3 (fp32 PyTorch oracle): "
                      f"best {dt:.1f} s, median {med:.1f} s, worst {max(times):.1f} s"}


def attn_flops(desc):
    """Algorithmic FLOPs of one window-attention forward launch: QK^T and PV over the real queries and keys
    (4 Nq Nk hd per (window, head); prompt keys count, padding does not)."""
    hd = desc.C // desc.heads
    return 4.0 * desc.Nq * (desc.Nq + desc.Np) * hd * desc.heads * desc.P * desc.B


def attn_exp_bound_seconds(desc, n_cu=256, clock_hz=2.4e9):
    """Lower bound of a launch from the transcendental pipe alone: one v_exp_f32 per (query, key) pair incl. the padded
    tiles, 8 issue cycles per 64 pairs on each of the 4 SIMDs of a CU (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost')."""
    pairs = float(desc.Nqp) * desc.Nkp * desc.heads * desc.P * desc.B
    return pairs / 64.0 * 8.0 / (n_cu * 4) / clock_hz


def attn_issue_bound_seconds(desc, n_cu=256, clock_hz=2.4e9):
    """Lower bound from the vector issue port of a SIMD, which every instruction of the key loop shares: per 32 keys x 16
    queries (512 pairs) a wave issues 8 v_exp_f32 (8.4 cycles each at eight waves per SIMD), 4 v_cvt_pk_bf16_f32 (4.5) and 3
    MFMAs (each holds the port for 8 of its 16 cycles) = 109 cycles; measured on this part by tools/ubench (the MFMAs of other
    waves overlap with plain VALU work but not with the port time they hold themselves)."""
    pairs = float(desc.Nqp) * desc.Nkp * desc.heads * desc.P * desc.B
    return pairs / 512.0 * (8 * 8.4 + 4 * 4.5 + 3 * 8.0) / (n_cu * 4) / clock_hz


def main():
    args = parse()
    import mivp_amd
    from mivp_amd import _lib, train
    from mivp_amd.swin_unetr import SwinUnetR

    rank, local, world = train.dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = args.device if args.device >= 0 else local
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    train.init_distributed(dev, args.backend)

    from mivp_amd import swin_ops
    if args.fp8_attn:
        swin_ops.USE_FP8_ATTN_FWD = True
    window = tuple(int(v) for v in args.window.split(","))
    conf, size, batch = train.make_conf(args.workload, window, args.dropout)
    if args.batch:
        batch = args.batch
    torch.manual_seed(0)
    ssl = conf.training_mode.startswith("self_supervised")
    if ssl:
        # BASELINE.json configs[0]: the students/teacher step (students_teacher.py:150-207) through MomentumModel
        from mivp_amd import students_teacher as ST
        from mivp_amd.losses import ClusteredPrototypeLoss
        mm = ST.MomentumModel(conf, SwinUnetR).to(dev).train()
        mm.copy_state_dict()
        # recorded steps synchronise one flat gradient bucket themselves (train.GraphedStep): no DistributedDataParallel
        student = train.wrap_ddp(mm.net_student, dev_index, gloo=args.backend == "gloo") if (world > 1 and not args.graph) else mm.net_student
        if world > 1 and not args.graph:
            mm.net_student = student
        net = mm
        opt = train.build_optimizer(mm, conf, capturable=args.graph)
        sched = train.build_scheduler(opt, conf)
        loss_prt = ClusteredPrototypeLoss(float(conf.reduction_factor), int(conf.k_means_iterations), float(conf.fwhm),
                                          static_jitter=args.graph)
        views = ST.synthetic_views(conf, batch, size, dev, rank)

        def eager_step():
            return ST.students_teacher_step(mm, opt, sched, loss_prt, conf, views)

        one_step = ST.graphed_students_teacher_step(mm, opt, sched, loss_prt, conf, views) if args.graph else eager_step
    else:
        model = SwinUnetR(conf).to(dev).train()
        net = train.wrap_ddp(model, dev_index, gloo=args.backend == "gloo") if (world > 1 and not args.graph) else model
        opt = train.build_optimizer(net, conf, capturable=args.graph)
        x, y = train.synthetic_batch(conf, batch, size, dev, rank)

        def eager_step():
            return train.train_step(net, opt, conf, x, y)

        one_step = train.graphed_train_step(net, opt, conf, x, y) if args.graph else eager_step

    # the largest MFMA-shaped kernel: the last decoder stage's conv_concat (implicit GEMM, K = 27*144, N = 48)
    hc = conf.hidden_channels
    dom_cin, dom_cout = hc[0] + hc[1], hc[0]
    _lib.profile_select(("mivp_conv3d_halo_fwd", "mivp_conv3d_fwd"),
                        lambda a: a[0]._obj.Cin == dom_cin and a[0]._obj.Cout == dom_cout, key="roofline")
    # the stage-0 window-attention forward launches (encoder stage 0 and the last decoder stage: C = hidden_channels[0])
    _lib.profile_select("mivp_win_attn_fwd", lambda a: a[0]._obj.C == hc[0] and a[0]._obj.has_mask == 0, key="attn")
    _lib.profile_select("mivp_win_attn_fwd", lambda a: a[0]._obj.C == hc[0] and a[0]._obj.has_mask != 0, key="attn_shift")

    def sync():
        train.barrier_sync(dev)

    # settle phase (untimed, before the W warm-up steps): the first ~20 steps after start-up run 15-25 % slower on
    # this pool (clock ramp from the low-power state, allocator growth); measured with tools/cpu_bound_check.py
    t_settle = time.perf_counter()
    n_settle = 0
    # under DDP every rank must run the same number of steps (each one is a collective): fixed count there
    while (n_settle < 30) if world > 1 else (n_settle < 10 or (time.perf_counter() - t_settle < args.settle and n_settle < 200)):
        one_step()
        n_settle += 1
        if n_settle % 10 == 0:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_step()
    sync()
    _lib.profile_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    sync()
    dt = time.perf_counter() - t0
    if args.graph:
        # a replay issues no C-ABI calls from Python, so the per-kernel event pairs have nothing to wrap: the same kernels
        # are timed in eager steps of the same state, outside the timed region
        loss = loss.clone()
        for _ in range(5):
            eager_step()
        sync()
    _lib.profile_reset(False)
    dt = train.max_over_ranks(dt, dev if args.backend == "nccl" else None)
    if not torch.isfinite(loss):
        raise SystemExit("loss is not finite")

    if rank == 0:
        kern_ms, kern_n, kern_desc = _lib.profile_result("roofline")
        roof_conv = None
        if kern_n:
            fl = conv_flops(kern_desc)
            achieved = fl / (kern_ms * 1e-3) / 1e12
            kname = "k_conv3d_halo<3>" if _lib.profile_entry("roofline") == "mivp_conv3d_halo_fwd" else "k_conv3d_fwd<3,8,2>"
            roof_conv = {"kernel": f"{kname} (decoder stage 2 conv_concat: 3x3x3 conv as MFMA GEMM, {kern_desc.Cin}->{kern_desc.Cout} channels, "
                                   f"{kern_desc.dims[0]}x{kern_desc.dims[1]}x{kern_desc.dims[2]} voxels x batch {kern_desc.B})",
                         "bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic_bytes("conv3d_dec2"),
                         "launches_timed": kern_n, "avg_launch_ms": kern_ms, "flops_per_launch": fl}
        roof = None
        parts = []
        for key in ("attn", "attn_shift"):
            ms, n, dsc = _lib.profile_result(key)
            if n:
                parts.append((key, ms, n, dsc))
        if parts:
            fl = sum(attn_flops(dsc) * n for _, _, n, dsc in parts)
            secs = sum(ms * 1e-3 * n for _, ms, n, _ in parts)
            bound = sum(attn_exp_bound_seconds(dsc) * n for _, _, n, dsc in parts)
            issue = sum(attn_issue_bound_seconds(dsc) * n for _, _, n, dsc in parts)
            dsc = parts[0][3]
            achieved = fl / secs / 1e12
            roof = {"kernel": f"k_win_attn_fwd<1,1,8,...> (stage-0 window attention forward, the dominant kernel family of the step: "
                              f"{dsc.B * dsc.P} windows x {dsc.heads} heads, {dsc.Nq} queries x {dsc.Nq}(+{dsc.Np} prompt) keys, "
                              f"head_dim {dsc.C // dsc.heads}; un-shifted and shifted launches together)",
                    "bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                    "binding_resource": "vector issue port (v_exp_f32 + v_cvt_pk + MFMA issue slots), not the matrix pipe: "
                                        "head_dim 12 gives one v_exp per 96 MFMA FLOP",
                    "frac_of_exp_issue_bound": bound / secs, "frac_of_vector_issue_bound": issue / secs,
                    "traffic": traffic_bytes("attn_fwd_stage0"),
                    "avg_launch_ms": {k: ms for k, ms, _, _ in parts}, "launches_timed": {k: n for k, _, n, _ in parts},
                    "flops_per_launch": attn_flops(dsc)}
        units = world * batch * args.steps
        line = {
            "metric": "3D volumes/sec (96^3, bf16) training step", "value": units / dt,
            "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" + (" (fp8 e4m3 window-attention forward)" if swin_ops.USE_FP8_ATTN_FWD else ""),
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: swin_unetr {conf.training_mode}, {conf.input_channels}-ch {size}^3, "
                                   f"batch {batch}/GPU, window {window}, enc_prompt={conf.use_encoder_prompting}, "
                                   f"dec_prompt={conf.use_decoder_prompting}, dropout {conf.attn_drop}, random-init weights"
                                   + (", students/teacher step (2 students + EMA teacher, ClusteredPrototypeLoss)" if ssl else ""),
                       "global_batch": world * batch, "parallelism": f"dp{world}", "final_loss": float(loss),
                       "backend": args.backend if world > 1 else None,
                       "launch": ("hip-graph replay (step recorded once" + ("; flat gradient all-reduce between the two graphs)" if world > 1 else ")"))
                                 if args.graph else "eager (one C-ABI call per kernel)"},
            "roofline": roof, "roofline_conv": roof_conv,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(conf, size)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
