#!/usr/bin/env python3
"""Which Python lines issue the small torch kernels of a step?  tools/trace_small_ops.py [workload]
torch.profiler with stacks over three eager steps; prints every aten op that launches a device kernel, with the innermost
mivp_amd / bench frame, sorted by device time."""
import sys, collections
import torch
sys.path.insert(0, ".")
import mivp_amd  # noqa
from mivp_amd import train
from mivp_amd.swin_unetr import SwinUnetR

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
conf, size, batch = train.make_conf(wl)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = SwinUnetR(conf).to(dev).train()
opt = train.build_optimizer(model, conf)
x, y = train.synthetic_batch(conf, batch, size, dev)
for _ in range(5):
    train.train_step(model, opt, conf, x, y)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        train.train_step(model, opt, conf, x, y)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
        continue
    frame = next((s for s in ev.stack if "visual-prompts_amd" in s or "mivp_amd" in s or "bench" in s), ev.stack[0] if ev.stack else "?")
    k = (ev.name, frame.strip()[-110:])
    agg[k][0] += 1
    agg[k][1] += sum(kk.duration for kk in ev.kernels)
for (name, frame), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n/3:6.1f}/step {us/3:8.1f} us/step  {name:28s} {frame}")

print("--- host-side op counts per step (copy-like ops, whether or not a kernel was attributed) ---")
cnt = collections.Counter()
for ev in prof.events():
    n = ev.name
    if any(t in n for t in ("copy", "clone", "contiguous", "Memcpy", "memcpy", "AccumulateGrad", "aten::to", "fill", "zero")):
        par = ev.cpu_parent.name if getattr(ev, "cpu_parent", None) is not None else "-"
        gp = ev.cpu_parent.cpu_parent.name if par != "-" and ev.cpu_parent.cpu_parent is not None else "-"
        cnt[(n, par[:60], gp[:60])] += 1
for (n, par, gp), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c/3:6.1f}/step  {n:36s} <- {par} <- {gp}")
