"""Halo-brick geometries on the deep decoder convolutions: result against the im2col kernel + time per call.
usage: python tools/ab_conv_bricks.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mivp_amd
from mivp_amd import ops
SH = {"dec2": (576, 192, (12, 12, 24)), "bott": (384, 384, (6, 6, 24)), "dec1": (288, 96, (24, 24, 24)), "odd": (32, 48, (7, 9, 13))}
dev = torch.device("cuda")
for name, (cin, cout, dims) in SH.items():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, *dims, cin, generator=g).to(dev, torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g).to(dev) / (27 * cin) ** 0.5
    b = torch.randn(cout, generator=g).to(dev)
    wp = ops.pack_conv_weight(w)
    saved = ops.halo_brick
    ops.halo_brick = lambda *a: 0
    ref = ops.conv3d(x, wp, b, cout).float()
    ops.halo_brick = saved
    print(name, "default brick", ops.halo_brick(4, dims, cout))
    for code in (4, 8, 6, 66, 36):
        y = ops.conv3d(x, wp, b, cout, force_halo=code)
        err = (y.float() - ref).abs().max().item()
        for it in range(12):
            if it == 2:
                torch.cuda.synchronize(); t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
            y = ops.conv3d(x, wp, b, cout, force_halo=code)
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / 10 * 1e3
        fl = 2.0 * 27 * cin * cout * 4 * dims[0] * dims[1] * dims[2]
        print("  brick %2d: max |diff to im2col| %.3g   %.1f us   %.0f TFLOP/s" % (code, err, us, fl / us / 1e6))
