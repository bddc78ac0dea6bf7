"""GPU parity: HIP Swin block (through the C ABI) vs the oracle.

Tolerance (written here as the north star asks): activations and GEMM weights are
bf16, accumulation fp32.  The oracle runs in fp32 on the SAME bf16-rounded input
and weights, so what remains is the kernel's internal bf16 rounding points
(LN output, q/k/v, P, o, t1) plus the final bf16 store: relative L2 error must
stay below 1.5e-2 of the residual-free signal; we assert rel-L2(y) <= 6e-3 on the
block output (dominated by the 2^-9 output rounding) and max-abs <= 6e-2.

PRIMARY bar (round 2): the rounding-aware oracle (``emulate_bf16=True``: the same fp32 arithmetic
with a bf16 rounding at exactly the points where the HIP path stores bf16 -- LayerNorm outputs, q, k
(in log2 units), v, the bias columns, P, o, t1 and the block output).  Against it only accumulation
order, the exp2 / rsqrt approximations and double-rounding flips remain: rel-L2(y) <= TIGHT_FWD.
The fp32-oracle assertion stays as the secondary, looser check.
"""
import pytest
import torch

from conftest import load_fixture, rel_l2

pytestmark = pytest.mark.gpu

TIGHT_FWD = 2e-3        # block output vs the rounding-aware oracle (measured margins: profiles/README.md, round 2)

BLOCKS = ["nopad_noshift", "nopad_shift", "nopad_shift_prompt", "oddpad_shift_prompt",
          "evenpad_noshift_prompt", "smalldim_shift", "smalldim_pad_prompt", "w442_shift_prompt"]


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _rounded_state(sd):
    out = {}
    for k, v in sd.items():
        if v.is_floating_point() and v.dim() == 2 and ("to_" in k or "proj.weight" in k or k.endswith("mlp.weight")):
            out[k] = _bf16_round(v)
        else:
            out[k] = v.clone()
    return out


def test_mfma_lane_map():
    import ctypes as C
    import mivp_amd
    from mivp_amd import _lib as L
    torch.manual_seed(0)
    a = torch.randint(-4, 5, (16, 32)).float()
    b = torch.randint(-4, 5, (16, 32)).float()
    ad, bd = a.to("cuda", torch.bfloat16), b.to("cuda", torch.bfloat16)
    c = torch.zeros(16, 16, device="cuda")
    L.call("mivp_selftest_mfma", L.ptr(ad), L.ptr(bd), L.ptr(c), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(c.cpu(), a @ b.t())


@pytest.mark.parametrize("tag", BLOCKS)
def test_block_forward_golden_shapes(tag):
    import mivp_amd
    from mivp_amd import swin_ops
    from oracle import swin_ref as S
    fx = load_fixture(f"block_{tag}")
    m = fx.meta
    sd = _rounded_state(fx["sd"])
    x = _bf16_round(fx["in"]["x"])
    prm = fx["in"].get("prompt")
    want = S.swin_block(x, prm, sd, "", m["window"], m["shift"], m["heads"])
    dev = torch.device("cuda")
    w = swin_ops.weights_from_state(sd, "", m["heads"], 64, m["n_prompt"], dev)
    xc = x.permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16)
    y, _ = swin_ops.swin_block_forward(xc, None if prm is None else prm.to(dev), w, None, m["window"], m["shift"])
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 4, 1, 2, 3)
    err = rel_l2(got, want)
    # (a forward-only call: the HIP attention keeps its softmax reference point at zero, oracle: zero_ref)
    want16 = S.swin_block(x, prm, sd, "", m["window"], m["shift"], m["heads"], emulate_bf16=True, zero_ref=True)
    err16 = rel_l2(got, want16)
    print(f"[tight] block_{tag}: vs rounding-aware oracle {err16:.3e}, vs fp32 oracle {err:.3e}")
    assert err16 < TIGHT_FWD, (tag, err16)
    assert err < 6e-3, (tag, err)
    assert float((got - want).abs().max()) < 6e-2


@pytest.mark.parametrize("window,dims,C,heads,n_prompt,shift", [
    ((7, 7, 7), (14, 14, 14), 48, 4, 64, (3, 3, 3)),     # stage-0 shape class: hd 12, 20 aug dims -> DK 32
    ((7, 7, 7), (12, 12, 24), 96, 8, 64, (0, 0, 0)),     # padded (12->14, 24->28), un-shifted
    ((7, 7, 7), (6, 6, 24), 192, 16, 64, (3, 3, 3)),     # dim < window on two axes, odd pad
    ((7, 7, 7), (12, 12, 24), 96, 4, 0, (3, 3, 3)),      # decoder: hd 24, no prompts
    ((7, 7, 7), (6, 6, 24), 192, 4, 64, (3, 3, 3)),      # decoder with prompts: hd 48
    ((8, 8, 4), (16, 16, 16), 48, 4, 64, (4, 4, 2)),     # yml window
    ((8, 8, 4), (4, 4, 8), 192, 16, 0, (4, 4, 2)),       # divisible axis padded by a full window
])
def test_block_forward_real_sizes(window, dims, C, heads, n_prompt, shift):
    import mivp_amd
    from mivp_amd import swin_ops
    from oracle import swin_ref as S
    from oracle.unetr_ref import _block_state
    gen = torch.Generator().manual_seed(1)
    sd = {}
    _block_state(sd, "", C, heads, list(window), 64, max(n_prompt, 1), n_prompt > 0, gen)
    for k in list(sd):
        if "norm.weight" in k:
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=gen)
        if "norm.bias" in k or k.endswith("proj.bias") or k.endswith("mlp.bias"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=gen)
    sd = _rounded_state(sd)
    x = _bf16_round(torch.randn(2, C, *dims, generator=gen))
    prm = 0.5 * torch.randn(n_prompt, C, generator=gen) if n_prompt else None
    want = S.swin_block(x, prm, sd, "", window, shift, heads)
    dev = torch.device("cuda")
    w = swin_ops.weights_from_state(sd, "", heads, 64, n_prompt, dev)
    xc = x.permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16)
    y, _ = swin_ops.swin_block_forward(xc, None if prm is None else prm.to(dev), w, None, window, shift)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 4, 1, 2, 3)
    err = rel_l2(got, want)
    want16 = S.swin_block(x, prm, sd, "", window, shift, heads, emulate_bf16=True, zero_ref=True)
    err16 = rel_l2(got, want16)
    print(f"[tight] block C={C} heads={heads} dims={dims} win={window}: vs rounding-aware oracle {err16:.3e}, vs fp32 oracle {err:.3e}")
    assert err16 < TIGHT_FWD, err16
    assert err < 6e-3, err
    assert float((got - want).abs().max()) < 8e-2


@pytest.mark.parametrize("gain,shift,n_prompt", [(6.0, (3, 3, 3), 64), (12.0, (0, 0, 0), 0), (12.0, (3, 3, 3), 64),
                                                 (40.0, (0, 0, 0), 0), (40.0, (3, 3, 3), 64)])
def test_block_sharp_softmax(gain, shift, n_prompt):
    """Large logits with a wide dynamic range (to_q / to_k scaled by ``gain``: logits x gain^2, |logit| up to several hundred
    in log2 units): the lazily refreshed reference point of the online softmax has to rescale again and again, nearly
    one-hot rows and fully suppressed keys appear, and the padding keys (excluded through a -30000 bias instead of a test)
    must stay at exactly zero weight.  At gain 40 logits differ by thousands of log2 units within a row: the forward's
    optimistic (test-free) softmax steps overflow and the tile is redone with the tested steps.  Forward against the oracle, then input / prompt gradients through both backward
    passes (which restart from the stored log-sum-exp)."""
    import mivp_amd
    from mivp_amd import swin_ops
    from oracle import swin_ref as S
    from oracle.unetr_ref import _block_state
    window, dims, C, heads = (7, 7, 7), (14, 14, 14), 48, 4
    gen = torch.Generator().manual_seed(7)
    sd = {}
    _block_state(sd, "", C, heads, list(window), 64, max(n_prompt, 1), n_prompt > 0, gen)
    sd["attn.to_q.weight"] = sd["attn.to_q.weight"] * gain
    sd["attn.to_k.weight"] = sd["attn.to_k.weight"] * gain
    sd = _rounded_state(sd)
    x = _bf16_round(torch.randn(1, C, *dims, generator=gen)).requires_grad_(True)
    prm = (0.5 * torch.randn(n_prompt, C, generator=gen)).requires_grad_(True) if n_prompt else None
    want = S.swin_block(x, prm, sd, "", window, shift, heads)
    dy = _bf16_round(torch.randn(want.shape, generator=gen))
    want.backward(dy)
    # conditioning yardstick: how far the fp32 oracle's own outputs move when its input carries bf16-level relative noise
    xn = (x.detach() * (1 + 2.0 ** -9 * torch.randn(x.shape, generator=gen))).requires_grad_(True)
    pn = prm.detach().clone().requires_grad_(True) if n_prompt else None
    wantn = S.swin_block(xn, pn, sd, "", window, shift, heads)
    wantn.backward(dy)
    yard_y = rel_l2(wantn.detach(), want.detach())
    yard_dx = rel_l2(xn.grad, x.grad)
    yard_dp = rel_l2(pn.grad, prm.grad) if n_prompt else 0.0
    dev = torch.device("cuda")
    w = swin_ops.weights_from_state(sd, "", heads, 64, n_prompt, dev, need_bwd=True)
    xc = x.detach().permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16)
    pd = None if prm is None else prm.detach().to(dev)
    y, saved = swin_ops.swin_block_forward(xc, pd, w, None, window, shift, save=True)
    dyc = dy.permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16)
    dx, dp, _ = swin_ops.swin_block_backward(saved, w, pd, dyc, True, n_prompt > 0)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 4, 1, 2, 3)
    assert torch.isfinite(got).all()
    # sharper softmax rows amplify the bf16 rounding of x, q and k (a logit error that grows with the gain decides between
    # near-ties): the bar is 3x the oracle's own sensitivity to 2^-9 input noise, or the unit-gain bars where that is small
    e_y = rel_l2(got, want.detach())
    e_dx = rel_l2(dx.float().cpu().permute(0, 4, 1, 2, 3), x.grad)
    assert e_y < max(1.5e-2, 3 * yard_y), (e_y, yard_y)
    assert e_dx < max(4e-2, 3 * yard_dx), (e_dx, yard_dx)
    if n_prompt:
        e_dp = rel_l2(dp.float().cpu(), prm.grad)
        assert e_dp < max(4e-2, 3 * yard_dp), (e_dp, yard_dp)
    # the forward-only call (no lse: zero-reference optimistic walk, csrc/swin_fwd.hip ZREF) on the same logits: at these gains
    # its row sums over- and underflow and the tiles take the tested walk -- same bar against the oracle as the saved call
    y0, none = swin_ops.swin_block_forward(xc, pd, w, None, window, shift)
    torch.cuda.synchronize()
    assert none is None
    got0 = y0.float().cpu().permute(0, 4, 1, 2, 3)
    assert torch.isfinite(got0).all()
    e_y0 = rel_l2(got0, want.detach())
    assert e_y0 < max(1.5e-2, 3 * yard_y), (e_y0, yard_y)


@pytest.mark.parametrize("shift,n_prompt", [((0, 0, 0), 64), ((3, 3, 3), 64), ((3, 3, 3), 0)])
def test_fp8_attention_variant(shift, n_prompt):
    """The experimental E4M3 forward attention (BASELINE.json configs[4]; csrc/swin_fwd_fp8.hip, off by default -- it is not
    faster, profiles/r02_fp8_attention.json): same semantics as the bf16 kernel, block output within 1e-2 of the fp32 oracle
    (3-bit mantissas on Q', K', V, P; the bf16 kernel meets 6e-3) and the saved log-sum-exp within 0.1 of the bf16 kernel's."""
    import mivp_amd
    from mivp_amd import swin_ops
    from oracle import swin_ref as S
    from oracle.unetr_ref import _block_state
    window, dims, C, heads = (7, 7, 7), (14, 14, 14), 48, 4
    gen = torch.Generator().manual_seed(11)
    sd = {}
    _block_state(sd, "", C, heads, list(window), 64, max(n_prompt, 1), n_prompt > 0, gen)
    sd = _rounded_state(sd)
    x = _bf16_round(torch.randn(2, C, *dims, generator=gen))
    prm = 0.5 * torch.randn(n_prompt, C, generator=gen) if n_prompt else None
    want = S.swin_block(x, prm, sd, "", window, shift, heads)
    dev = torch.device("cuda")
    w = swin_ops.weights_from_state(sd, "", heads, 64, n_prompt, dev)
    xc = x.permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16)
    pd = None if prm is None else prm.to(dev)
    _, ref = swin_ops.swin_block_forward(xc, pd, w, None, window, shift, save=True)
    swin_ops.USE_FP8_ATTN_FWD = True
    try:
        y, sv = swin_ops.swin_block_forward(xc, pd, w, None, window, shift, save=True)
    finally:
        swin_ops.USE_FP8_ATTN_FWD = False
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 4, 1, 2, 3)
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) < 1e-2
    nq = sv.desc.Nq
    assert float((sv.lse[..., :nq] - ref.lse[..., :nq]).abs().max()) < 0.1
    assert rel_l2(sv.o.float().cpu(), ref.o.float().cpu()) < 8e-2          # the attention output itself: E4M3 noise
