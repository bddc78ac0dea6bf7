"""Micro-benchmark of the 3x3x3 conv kernel at a model shape (for rocprofv3 runs).
usage: python tools/bench_conv.py [dec2|dec1|dec0|bott|head] [iters]"""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mivp_amd
from mivp_amd import ops
SH = {"headf": (48, 2, (96, 96, 96), True), "dec2": (144, 48, (48, 48, 48), True), "dec1": (288, 96, (24, 24, 24), True), "dec0": (576, 192, (12, 12, 24), True),
      "bott": (384, 384, (6, 6, 24), False), "head": (48, 2, (96, 96, 96), True)}
name = sys.argv[1] if len(sys.argv) > 1 else "dec2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
halo = int(sys.argv[3]) if len(sys.argv) > 3 else None      # 8 / 4: force that brick width, 0: force the im2col kernel
cin, cout, dims, aff = SH[name]
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
x = torch.randn(4, *dims, cin, generator=g).to(dev, torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, 3, generator=g).to(dev) / (27 * cin) ** 0.5
b = torch.zeros(cout, device=dev)
wp = ops.pack_conv_weight(w)
scale = None
shift = None
for it in range(iters + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    if name == "headf":
        y = ops.head_conv(x, w, b, torch.ones(cin, device=dev), torch.zeros(cin, device=dev))
    else:
        if halo is None:
            y = ops.conv3d(x, wp, b, cout, scale, shift, aff and name != "head", None, name == "head")
        elif halo == 0:
            ops.halo_brick_saved = getattr(ops, "halo_brick_saved", ops.halo_brick)
            ops.halo_brick = lambda *a: 0
            y = ops.conv3d(x, wp, b, cout, scale, shift, aff and name != "head", None, name == "head")
        else:
            y = ops.conv3d(x, wp, b, cout, force_halo=halo)
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / iters
fl = 2.0 * 27 * cin * cout * 4 * dims[0] * dims[1] * dims[2]
print(name, "ms", ms, "TFLOP/s", fl / ms / 1e9)
