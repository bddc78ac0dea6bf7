"""fp8 vs bf16 window attention at BASELINE.json configs[4]'s stage-0 shape (96^3, batch 8, encoder prompts): time of the
forward attention launch and error of the block output against the fp32 oracle.  Writes gpurun_out/fp8_attention.json.
usage: python tools/fp8_attn.py"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mivp_amd  # noqa: F401
from mivp_amd import swin_ops, _lib
from oracle import swin_ref as S
from oracle.unetr_ref import _block_state


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


res = {}
dev = torch.device("cuda")
window, C, heads, npr = (7, 7, 7), 48, 4, 64
gen = torch.Generator().manual_seed(0)
sd = {}
_block_state(sd, "", C, heads, list(window), 64, npr, True, gen)
sd = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() and v.dim() == 2 and ".pe." not in k and not k.startswith("pe.") else v)
      for k, v in sd.items()}
prm = 0.5 * torch.randn(npr, C, generator=gen)
# --- accuracy on a small volume (the CPU oracle) ---
for shift in ((0, 0, 0), (3, 3, 3)):
    x = torch.randn(2, C, 14, 14, 14, generator=gen).to(torch.bfloat16).float()
    want = S.swin_block(x, prm, sd, "", window, shift, heads)
    want16 = S.swin_block(x, prm, sd, "", window, shift, heads, emulate_bf16=True)
    w = swin_ops.weights_from_state(sd, "", heads, 64, npr, dev)
    xc = x.permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16)
    out = {}
    for name, flag in (("bf16", False), ("fp8", True)):
        swin_ops.USE_FP8_ATTN_FWD = flag
        y, _ = swin_ops.swin_block_forward(xc, prm.to(dev), w, None, window, shift)
        torch.cuda.synchronize()
        got = y.float().cpu().permute(0, 4, 1, 2, 3)
        out[name] = {"block_rel_l2_vs_fp32_oracle": rel(got, want), "block_rel_l2_vs_rounding_aware_oracle": rel(got, want16)}
    swin_ops.USE_FP8_ATTN_FWD = False
    res["accuracy_shift%d" % shift[0]] = out
# --- time at the configs[4] stage-0 shape: B = 8, 48^3 tokens, 4 heads, 343 + 64 keys ---
x = torch.randn(8, 48, 48, 48, C, generator=gen).to(dev, torch.bfloat16)
for shift in ((0, 0, 0), (3, 3, 3)):
    t = {}
    for name, flag in (("bf16", False), ("fp8", True)):
        swin_ops.USE_FP8_ATTN_FWD = flag
        entry = "mivp_win_attn_fwd_fp8" if flag else "mivp_win_attn_fwd"
        _lib.profile_select(entry, None, key="a")
        for it in range(12):
            if it == 2:
                torch.cuda.synchronize()
                _lib.profile_reset(True)
            swin_ops.swin_block_forward(x, prm.to(dev), w, None, window, shift)
        ms, n, _ = _lib.profile_result("a")
        _lib.profile_reset(False)
        t[name] = {"attention_launch_us": 1e3 * ms, "launches": n}
    swin_ops.USE_FP8_ATTN_FWD = False
    res["time_shift%d" % shift[0]] = t
res["shape"] = "stage-0 block of configs[4]: batch 8 x 343 windows x 4 heads, 343 queries x (343 + 64 prompt) keys, head_dim 12"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fp8_attention.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
