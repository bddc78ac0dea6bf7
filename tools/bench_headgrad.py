"""Time the two head weight-gradient paths on the cfg1 head shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mivp_amd
from mivp_amd import ops

dev = "cuda"
x = torch.randn(4, 96, 96, 96, 48, device=dev).bfloat16()
dy = torch.zeros(4, 96, 96, 96, 8, device=dev).bfloat16()
dy[..., :2] = torch.randn(4, 96, 96, 96, 2, device=dev).bfloat16()

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

print("wgrad_rows      %.3f ms" % timeit(lambda: ops.conv3d_wgrad_rows(x, dy, 2)))
print("gemm_tn swapped %.3f ms" % timeit(lambda: ops.conv3d_wgrad(x, dy, 2)))
y = torch.empty(4, 96, 96, 96, 48, device=dev, dtype=torch.bfloat16)
lat = torch.randn(4, 48, 48, 48, 48, device=dev).bfloat16()
print("upcat final     %.3f ms" % timeit(lambda: ops.upcat(lat, None, (2, 2, 2))))
