"""Time the low-resolution head kernels on the cfg1 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mivp_amd
from mivp_amd import ops

dev = "cuda"
x = torch.randn(4, 48, 48, 48, 48, device=dev).bfloat16()
dy = torch.randn(4, 96, 96, 96, 2, device=dev)
w = torch.randn(2, 48, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(2, device=dev)
g1, b1 = torch.ones(48, device=dev), torch.zeros(48, device=dev)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n

sc, sh, mr = ops.uphead_batch_stats(x, g1, b1, 1e-5)
wf = ops.uphead_fold(w, sc, sh)
print("stats   %.3f ms" % timeit(lambda: ops.uphead_batch_stats(x, g1, b1, 1e-5)))
print("forward %.3f ms" % timeit(lambda: ops.uphead_forward(x, wf, b, 2)))
print("grads   %.3f ms" % timeit(lambda: ops.uphead_gs(x, dy, 2)))
