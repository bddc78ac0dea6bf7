"""Diagnostic (GPU box): per-stage error of a full-size forward, end to end and restarted from the HIP path's own stage
inputs (tests/stagewise.py).  usage: python tools/stage_err.py [workload] [size] [fp32]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import mivp_amd  # noqa: F401
from mivp_amd import train
from oracle.unetr_ref import random_state
import stagewise

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 96
emul = (sys.argv[3] != "fp32") if len(sys.argv) > 3 else True
conf, _, _ = train.make_conf(workload)
sd = stagewise.round_weights(random_state(conf, seed=3))
x = torch.rand(1, conf.input_channels, size, size, size, generator=torch.Generator().manual_seed(9))
print(f"{'stage':16s} {'end-to-end':>11s} {'restarted':>11s}")
stagewise.stagewise_errors(conf, sd, x, emul, report=lambda s: print(s, flush=True))
