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
  roofline      the dominant kernel (3x3x3 implicit-GEMM conv of the last decoder stage, 144->48 channels
                at 48^3): algorithmic FLOPs per launch / its mean duration measured with HIP events on the
                launch stream inside the timed region, against the dense bf16 MFMA peak.
  cpu_baseline  the CPU oracle (oracle/, "port") timed on this host's cores on a bounded sample
                (one training step on ONE 96^3 volume), rank 0 at N=1 only.
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
    return ap.parse_args()


def conv_flops(desc):
    vox = desc.B * desc.dims[0] * desc.dims[1] * desc.dims[2]
    return 2.0 * 27 * desc.Cin * desc.Cout * vox


def traffic_bytes():
    """HBM-side bytes per launch of the roofline kernel from the committed PMC passes (profiles/r01_traffic.json:
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate rocprofv3 --pmc runs); None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["conv3d_dec2_6x6"]["hbm_bytes_per_launch"])   # the brick form the model runs
    except (OSError, KeyError, ValueError):
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
    """One oracle training step (forward + loss + backward + AdamW) on ONE volume, all usable host cores."""
    from oracle.unetr_ref import OracleSwinUnetR, random_state
    from oracle.loss_ref import dice_focal_loss
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = random_state(conf, seed=0)
    model = OracleSwinUnetR(conf, sd)
    keys = model.trainable_keys()
    for k in keys:
        sd[k].requires_grad_(True)
    opt = torch.optim.AdamW([sd[k] for k in keys], lr=1e-3, weight_decay=0.0)
    g = torch.Generator().manual_seed(1234)
    nvol = 4 if not (conf.use_encoder_prompting or conf.use_decoder_prompting) else 1     # ~10-30 s of CPU work
    x = torch.rand(nvol, conf.input_channels, size, size, size, generator=g)
    mode = conf.training_mode
    n_cls = conf.output_channels_downstream if mode == "downstream" else conf.output_channels_pretrain
    y = torch.randint(0, n_cls, (nvol, 1, size, size, size), generator=g).float()
    t0 = time.perf_counter()
    out, _ = model(x, training=True)
    if mode == "downstream":                       # same objectives as mivp_amd.train.step_loss
        loss = dice_focal_loss(out["downstream"], y, conf.include_background)
    elif mode.startswith("supervised"):
        loss = dice_focal_loss(out["seg_pred"], y, conf.include_background)
    else:
        loss = (out["latent_outputs"] ** 2).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
    dt = time.perf_counter() - t0
    return {"value": nvol / dt, "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"1 training step on a batch of {nvol} volume(s) of {size}^3 (fp32 PyTorch oracle, {dt:.1f} s)"}


def main():
    args = parse()
    import mivp_amd
    from mivp_amd import _lib, train
    from mivp_amd.swin_unetr import SwinUnetR

    rank, local, world = train.dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    train.init_distributed(dev, "nccl")

    window = tuple(int(v) for v in args.window.split(","))
    conf, size, batch = train.make_conf(args.workload, window, args.dropout)
    if args.batch:
        batch = args.batch
    torch.manual_seed(0)
    model = SwinUnetR(conf).to(dev).train()
    net = train.wrap_ddp(model, local) if world > 1 else model
    opt = train.build_optimizer(net, conf)
    x, y = train.synthetic_batch(conf, batch, size, dev, rank)

    # dominant kernel: the last decoder stage's conv_concat (implicit GEMM, K = 27*144, N = 48)
    hc = conf.hidden_channels
    dom_cin, dom_cout = hc[0] + hc[1], hc[0]
    _lib.profile_select(("mivp_conv3d_halo_fwd", "mivp_conv3d_fwd"),
                        lambda a: a[0]._obj.Cin == dom_cin and a[0]._obj.Cout == dom_cout)

    def sync():
        train.barrier_sync(dev)

    # settle phase (untimed, before the W warm-up steps): the first ~20 steps after start-up run 15-25 % slower on
    # this pool (clock ramp from the low-power state, allocator growth); measured with tools/cpu_bound_check.py
    t_settle = time.perf_counter()
    n_settle = 0
    # under DDP every rank must run the same number of steps (each one is a collective): fixed count there
    while (n_settle < 30) if world > 1 else (n_settle < 10 or (time.perf_counter() - t_settle < args.settle and n_settle < 200)):
        train.train_step(net, opt, conf, x, y)
        n_settle += 1
        if n_settle % 10 == 0:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        train.train_step(net, opt, conf, x, y)
    sync()
    _lib.profile_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train.train_step(net, opt, conf, x, y)
    sync()
    dt = time.perf_counter() - t0
    _lib.profile_reset(False)
    dt = train.max_over_ranks(dt, dev)
    if not torch.isfinite(loss):
        raise SystemExit("loss is not finite")

    if rank == 0:
        kern_ms, kern_n, kern_desc = _lib.profile_result()
        roof = None
        if kern_n:
            fl = conv_flops(kern_desc)
            achieved = fl / (kern_ms * 1e-3) / 1e12
            kname = "k_conv3d_halo<3>" if _lib.profile_entry() == "mivp_conv3d_halo_fwd" else "k_conv3d_fwd<3,8,2>"
            roof = {"kernel": f"{kname} (decoder stage 2 conv_concat: 3x3x3 conv as MFMA GEMM, {kern_desc.Cin}->{kern_desc.Cout} channels, "
                              f"{kern_desc.dims[0]}x{kern_desc.dims[1]}x{kern_desc.dims[2]} voxels x batch {kern_desc.B})",
                    "bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic_bytes(),
                    "launches_timed": kern_n, "avg_launch_ms": kern_ms, "flops_per_launch": fl}
        line = {
            "metric": "3D volumes/sec (96^3, bf16) training step", "value": world * batch * args.steps / dt,
            "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: swin_unetr {conf.training_mode}, {conf.input_channels}-ch {size}^3, "
                                   f"batch {batch}/GPU, window {window}, enc_prompt={conf.use_encoder_prompting}, "
                                   f"dec_prompt={conf.use_decoder_prompting}, dropout {conf.attn_drop}, random-init weights",
                       "global_batch": world * batch, "parallelism": f"dp{world}", "final_loss": float(loss)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(conf, size)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
