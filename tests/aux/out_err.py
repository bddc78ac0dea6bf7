"""Forward error of the product model against the oracle on the toy fixtures (prints rel-L2 per output)."""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))           # tests/ (conftest, test_hip_model)
sys.path.insert(0, os.path.join(HERE, "..", ".."))     # repository root
from argparse import Namespace
import torch
from conftest import load_fixture, rel_l2
from test_hip_model import round_weights, OTHER
import mivp_amd
from mivp_amd.swin_unetr import SwinUnetR
from oracle.unetr_ref import OracleSwinUnetR

for tag in OTHER:
    fx = load_fixture(f"unetr_{tag}")
    conf = Namespace(**fx.meta["conf"])
    sd = round_weights(fx["sd"])
    x = fx["in"]["x"]
    with torch.no_grad():
        want, _ = OracleSwinUnetR(conf, sd)(x, training=True)
        xp = x * (1 + 2.0 ** -9 * torch.randn(x.shape, generator=torch.Generator().manual_seed(1)))
        wantp, _ = OracleSwinUnetR(conf, sd)(xp, training=True)
        model = SwinUnetR(conf); model.load_state_dict(sd, strict=True); model.to("cuda").train()
        out = model(x.to("cuda"))
        model2 = SwinUnetR(conf); model2.load_state_dict(sd, strict=True); model2.to("cuda").train()
        outp = model2(xp.to("cuda"))
    for k, v in want.items():
        print(tag, k, "hip-vs-oracle %.4f" % rel_l2(out[k].float().cpu(), v), "oracle self-noise %.4f" % rel_l2(wantp[k], v),
              "hip self-noise %.4f" % rel_l2(outp[k].float().cpu(), out[k].float().cpu()))
