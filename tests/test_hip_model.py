"""GPU end-to-end parity: the product SwinUnetR (HIP kernels through the C ABI) vs the oracle on the
tiny-model golden configurations (parity unpinned at the MONAI boundary, see gen_golden.py).

Both sides load the SAME state dict whose matrix / conv weights were rounded to bf16 (the product
stores GEMM operands in bf16), so the comparison isolates kernel arithmetic.  Tolerances: the
reference's own bf16-autocast path differs from its fp32 path by 1.1-1.4e-2 rel-L2 end to end
(BASELINE.md); we require the HIP forward within 1.5e-2 rel-L2 of the fp32 oracle and every trained
parameter's gradient within 5e-2 rel-L2 (tiny 16^3 volumes make BatchNorm statistics noisy; at
realistic sizes the gap is smaller, see test_real_channels_forward).  The golden (un-rounded fp32)
output must be within 3e-2."""
from argparse import Namespace

import pytest
import torch

from conftest import load_fixture, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def round_weights(sd):
    out = {}
    for k, v in sd.items():
        if v.is_floating_point() and v.dim() >= 2 and not k.startswith("prompt_tokens") and ".pe." not in k \
                and not k.startswith("input_layer.0"):
            out[k] = r16(v)
        else:
            out[k] = v.clone()
    return out


DOWN = ["downstream_e0d0", "downstream_e0d1", "downstream_e1d0", "downstream_e1d1", "downstream_e1d1_simple"]


@pytest.mark.parametrize("tag", DOWN)
def test_downstream_forward_backward(tag):
    import mivp_amd
    from mivp_amd.swin_unetr import SwinUnetR
    from oracle.unetr_ref import OracleSwinUnetR
    fx = load_fixture(f"unetr_{tag}")
    conf = Namespace(**fx.meta["conf"])
    sd = round_weights(fx["sd"])
    x = fx["in"]["x"]
    gout = fx["in"]["gout"]
    # oracle
    osd = {k: v.clone() for k, v in sd.items()}
    for k in fx.meta["trainable"]:
        osd[k].requires_grad_(True)
    oracle = OracleSwinUnetR(conf, osd)
    want, nb = oracle(x, training=True)
    (want["downstream"] * gout).sum().backward()
    # product
    model = SwinUnetR(conf)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    out = model(x.to(DEV))["downstream"]
    assert out.shape == want["downstream"].shape and out.dtype == torch.float32
    (out * gout.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    err = rel_l2(out.cpu(), want["downstream"])
    assert err < 1.5e-2, err
    assert rel_l2(out.cpu(), fx["out"]["downstream"]) < 3e-2
    params = dict(model.named_parameters())
    assert sorted(k for k, q in params.items() if q.requires_grad) == sorted(fx.meta["trainable"])
    worst = {}
    for k in fx.meta["trainable"]:
        g, w = params[k].grad, osd[k].grad
        assert g is not None, k
        if float(w.norm()) < 1e-7:
            assert float(g.norm()) < 1e-4, k
            continue
        worst[k] = rel_l2(g.cpu(), w)
    bad = {k: v for k, v in worst.items() if v > 5e-2}
    assert not bad, bad
    # frozen BatchNorms still ran in train mode: running statistics moved exactly as the reference's
    msd = model.state_dict()
    for k, v in nb.items():
        if v.is_floating_point():
            assert rel_l2(msd[k].cpu(), v) < 2e-2, k
        else:
            assert int(msd[k]) == int(v), k


@pytest.mark.parametrize("tag", ["self_supervised_learning_all_e1d0", "supervised_learning_all_e0d0"])
def test_other_modes_forward_only(tag):
    import mivp_amd
    from mivp_amd.swin_unetr import SwinUnetR
    from oracle.unetr_ref import OracleSwinUnetR
    fx = load_fixture(f"unetr_{tag}")
    conf = Namespace(**fx.meta["conf"])
    sd = round_weights(fx["sd"])
    want, _ = OracleSwinUnetR(conf, sd)(fx["in"]["x"], training=True)
    model = SwinUnetR(conf)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    with torch.no_grad():
        out = model(fx["in"]["x"].to(DEV))
    torch.cuda.synchronize()
    for k, v in want.items():
        assert rel_l2(out[k].float().cpu(), v) < 1.5e-2, k
    # training these modes needs weight-gradient kernels that are not built yet: must fail loudly
    with pytest.raises(NotImplementedError):
        model(fx["in"]["x"].to(DEV))


def test_real_channels_forward():
    """yml channel widths (48..384, heads 4/8/16), window 7, 32^3 volume, encoder+decoder prompts."""
    import mivp_amd
    from mivp_amd.swin_unetr import SwinUnetR
    from oracle.unetr_ref import OracleSwinUnetR, default_conf, random_state
    conf = default_conf(attn_window_size=[7, 7, 7], use_encoder_prompting=True, use_decoder_prompting=True)
    sd = round_weights(random_state(conf, seed=5))
    x = torch.rand(1, 1, 32, 32, 32, generator=torch.Generator().manual_seed(0))
    want, _ = OracleSwinUnetR(conf, sd)(x, training=True)
    model = SwinUnetR(conf)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    with torch.no_grad():
        out = model(x.to(DEV))["downstream"]
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), want["downstream"]) < 1.5e-2
    agree = float((out.cpu().argmax(1) == want["downstream"].argmax(1)).float().mean())
    assert agree > 0.98, agree
