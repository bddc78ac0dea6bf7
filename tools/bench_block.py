"""Micro-benchmark of one Swin block (forward + backward) at a model-stage shape, for rocprofv3 runs.
usage: python tools/bench_block.py [stage] [iters]   stage in {enc0, enc1, enc2, dec0, dec1, dec2}"""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mivp_amd
from mivp_amd import swin_ops
from mivp_amd.swin_unetr import SwinTransformerBlock

STAGES = {  # C, heads, dims (96^3 input), prompts
    "enc0": (48, 4, (48, 48, 48), 64), "enc1": (96, 8, (24, 24, 24), 64), "enc2": (192, 16, (12, 12, 24), 64),
    "dec0": (192, 4, (12, 12, 24), 0), "dec1": (96, 4, (24, 24, 24), 0), "dec2": (48, 4, (48, 48, 48), 0),
}
if os.environ.get("MIVP_TWO_PASS_ATTN_BWD"):          # A/B: the dq + dkv pair instead of the one-pass backward
    swin_ops.USE_FUSED_ATTN_BWD = False
stage = sys.argv[1] if len(sys.argv) > 1 else "enc0"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shift = (3, 3, 3) if (len(sys.argv) > 3 and "shift" in sys.argv[3:]) else (0, 0, 0)
fwd_only = "fwdonly" in sys.argv[3:]          # forward-only calls (nothing saved: the zero-reference attention walk), no prompts
no_prompt = fwd_only or "noprompt" in sys.argv[3:]
C, heads, dims, npr = STAGES[stage]
if no_prompt:
    npr = 0
B, window = 4, (7, 7, 7)
gen = torch.Generator().manual_seed(0)
torch.manual_seed(0)
sd = SwinTransformerBlock(C, window, 64, heads, 1, max(npr, 1), use_token_params=npr > 0).state_dict()   # random init
dev = torch.device("cuda")
w = swin_ops.weights_from_state(sd, "", heads, 64, npr, dev, need_bwd=True)
x = torch.randn(B, *dims, C, generator=gen).to(dev, torch.bfloat16)
prm = (0.5 * torch.randn(npr, C, generator=gen)).to(dev) if npr else None
dy = torch.randn(B, *dims, C, generator=gen).to(dev, torch.bfloat16)
for it in range(iters + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    y, saved = swin_ops.swin_block_forward(x, prm, w, None, window, shift, save=not fwd_only)
    if not fwd_only:
        dx, dp, dts = swin_ops.swin_block_backward(saved, w, prm, dy, True, npr > 0)
t1.record(); torch.cuda.synchronize()
print(stage, "fwd+bwd ms/iter", t0.elapsed_time(t1) / iters)
