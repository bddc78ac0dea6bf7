"""Diagnostic (not a test): per-parameter gradient error of the HIP path vs the oracle, with two
conditioning yardsticks (how far the oracle's / the HIP path's own gradient moves under bf16-level
input noise).  usage: python tests/aux/grad_report.py TAG [SIZE]"""
import sys
from argparse import Namespace
import torch
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
from conftest import load_fixture, rel_l2
from test_hip_model import round_weights
import mivp_amd
from mivp_amd.swin_unetr import SwinUnetR
from oracle.unetr_ref import OracleSwinUnetR

tag = sys.argv[1] if len(sys.argv) > 1 else "downstream_e1d1"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fx = load_fixture(f"unetr_{tag}")
conf = Namespace(**fx.meta["conf"])
sd = round_weights(fx["sd"])
g0 = torch.Generator().manual_seed(0)
x = fx["in"]["x"] if size == 16 else torch.rand(2, 1, size, size, size, generator=g0)
gout = fx["in"]["gout"] if size == 16 else torch.randn(2, 2, size, size, size, generator=g0) / (2 * 2 * size ** 3) ** 0.5
noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
xp = x * (1 + 2.0 ** -9 * noise)


def oracle_grads(xx):
    osd = {k: v.clone() for k, v in sd.items()}
    for k in fx.meta["trainable"]:
        osd[k].requires_grad_(True)
    want, _ = OracleSwinUnetR(conf, osd)(xx, training=True)
    (want["downstream"] * gout).sum().backward()
    return want["downstream"].detach(), {k: osd[k].grad for k in fx.meta["trainable"]}


def hip_grads(xx):
    model = SwinUnetR(conf); model.load_state_dict(sd); model.cuda().train()
    out = model(xx.cuda())["downstream"]
    (out * gout.cuda()).sum().backward()
    torch.cuda.synchronize()
    return out.detach().cpu(), {k: q.grad.cpu() for k, q in model.named_parameters() if q.requires_grad}


wo, go = oracle_grads(x)
_, go2 = oracle_grads(xp)
wh, gh = hip_grads(x)
_, gh2 = hip_grads(xp)
print(tag, size, "fwd rel", rel_l2(wh, wo))
for k in fx.meta["trainable"]:
    cos = float(torch.nn.functional.cosine_similarity(gh[k].reshape(-1), go[k].reshape(-1), dim=0))
    print(f"  {k:66s} |g|={float(go[k].norm()):.2e} err={rel_l2(gh[k], go[k]):.3f} cos={cos:.4f} "
          f"yard_oracle={rel_l2(go2[k], go[k]):.3f} yard_hip={rel_l2(gh2[k], gh[k]):.3f}")
