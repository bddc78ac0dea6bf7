"""GPU parity at the sizes and configurations BASELINE.json names (VERDICT r1 "parity holes"):

  * configs[1] / configs[2] at the benchmark's own size: batch 1 of a 96^3 volume, 7^3 windows (343-token windows, 343
    windows in stage 0, the 48 -> 49 odd pad with its one-voxel shift, the low-resolution head) against the CPU oracle --
    logits, and for configs[2] every prompt-token / prompt-bias / head gradient;
  * configs[3] (4-channel 128^3, both prompt sides): oracle parity at a reduced 4-channel size + size-independent properties
    at the full 128^3;
  * configs[4]'s shape (batch 8 of 96^3, encoder prompts; bf16 attention -- the fp8 variant is measured in
    profiles/) through the same properties;
  * the second optimizer step really sees the first step's weights (fused AdamW does not bump version counters);
  * Dice of the arg-max segmentation within 1e-4 of the oracle's on a model a few optimisation steps in (north star:
    "Dice within 1e-4"; utils.py:41-64 restated in oracle/loss_ref.py).

The oracle here is the rounding-aware one (``emulate_bf16=True``: the reference's fp32 arithmetic, bf16 rounding wherever the
HIP path stores bf16).  Full-size forwards are checked STAGE BY STAGE (tests/stagewise.py): each stage restarted from the
HIP path's own input must meet a tight bar, while end to end two bf16-storing chains drift apart by ~1e-3 per block."""
from argparse import Namespace

import os

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"

# end-to-end logits, against the rounding-aware oracle (E2E_EMUL) and against the PLAIN fp32 oracle pinned to the reference's
# fixtures (E2E_FP32): the same figure -- the reference's own bf16-autocast gap is 1.1-1.4e-2
E2E_EMUL = 1.5e-2
E2E_FP32 = 1.5e-2


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


def _product(conf, sd, x, train_mode=True):
    import mivp_amd  # noqa: F401
    from mivp_amd.swin_unetr import SwinUnetR
    model = SwinUnetR(conf)
    model.load_state_dict(sd, strict=True)
    model.to(DEV)
    model.train(train_mode)
    return model


def _cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.reshape(-1).double(), b.reshape(-1).double(), dim=0))


# Restarted-stage bars (every stage of the HIP forward against the rounding-aware oracle fed the HIP path's own stage
# input; measured on MI355X at 96^3 / window 7, batch 1 -- tools/stage_err.py, profiles/README.md round 2):
#   Swin blocks 1.1-2.1e-3, patch merging 5-7e-5, convs 2e-5-1.4e-4, upsample+concat 0 (bit-exact), patch embedding
#   1.5e-5, low-resolution head 1.1-1.6e-4 (hi + lo folded weights, fp16 tap planes; 3.7e-3 with single-bf16 folded weights).
STAGE_BARS = {"b0": 2.5e-3, "b1": 2.5e-3, "merge": 3e-4, "conv": 5e-4, "bottleneck": 5e-4, "upcat": 1e-5, "embed": 1e-4,
              "logits": 5e-4}


BWD_BAR = {"dx": 3e-3}    # restarted per-block backward: dx is a stored bf16 tensor (its own rounding: 2.35e-3 measured everywhere)
BWD_PARAM_BAR = 1e-2        # prompt-token / prompt-bias gradients, f32 sums over all windows (measured 1.2e-3 .. 7.7e-3)


def _check_stagewise(conf, sd, x, tag):
    import stagewise
    lines = []
    res = stagewise.stagewise_errors(conf, sd, x, emul=True, report=lines.append)
    print(f"[stagewise {tag}] stage / end-to-end / restarted\n" + "\n".join(lines))
    for name, (e2e, restarted) in res.items():
        bar = STAGE_BARS[name.split(".")[-1]]
        assert restarted < bar, (tag, name, restarted, bar)
    # end to end the per-stage storage-rounding drift adds up (12 blocks x ~1e-3): the reference's own bf16 path is
    # 1.1-1.4e-2 away from its fp32 path
    assert res["logits"][0] < E2E_EMUL, res["logits"]
    return res


@pytest.mark.parametrize("workload", ["cfg1", "cfg2"])
def test_full_size_96_oracle_parity(workload):
    """B = 1, 96^3, window 7 (343 windows of 343 tokens in stage 0, the 48 -> 49 odd pad, the low-resolution head): every
    stage of the forward against the rounding-aware oracle, restarted per stage and end to end; cfg2 additionally all
    trainable gradients (prompt tokens, prompt-bias parameters, head) of sum(logits * g) against autograd over the oracle."""
    from mivp_amd import train
    from oracle.unetr_ref import OracleSwinUnetR, random_state
    conf, size, _ = train.make_conf(workload)
    sd = round_weights(random_state(conf, seed=3))
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(1, conf.input_channels, size, size, size, generator=gen)
    gout = torch.randn(1, conf.output_channels_downstream, size, size, size, generator=gen) / size ** 1.5
    _check_stagewise(conf, sd, x, f"{workload} 96^3")
    # the same forward against the PLAIN fp32 oracle (no rounding emulation: the restatement that the reference's own
    # fixtures pin), end to end
    with torch.no_grad():
        want32, _ = OracleSwinUnetR(conf, sd)(x, training=True)
        got = _product(conf, sd, x)(x.to(DEV))["downstream"].float().cpu()
    e32 = rel_l2(got, want32["downstream"])
    print(f"[{workload} 96^3] logits rel-L2 vs the plain fp32 oracle {e32:.3e} (bar {E2E_FP32})")
    assert e32 < E2E_FP32, e32
    if workload != "cfg2":
        return
    osd = {k: v.clone() for k, v in sd.items()}
    orc = OracleSwinUnetR(conf, osd, emulate_bf16=True)
    keys = orc.trainable_keys()
    for k in keys:
        osd[k].requires_grad_(True)
    want, _ = orc(x, training=True)
    (want["downstream"] * gout).sum().backward()
    model = _product(conf, sd, x)
    out = model(x.to(DEV))["downstream"]
    (out * gout.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    worst = {}
    for k in keys:
        g, w = params[k].grad, osd[k].grad
        assert g is not None and torch.isfinite(g).all(), k
        worst[k] = (rel_l2(g.cpu(), w), _cos(g.cpu(), w))
    ranked = sorted(worst.items(), key=lambda kv: -kv[1][0])
    print("[full-size cfg2] worst gradient errors vs rounding-aware oracle:",
          [(k, f"{e:.2e}", f"{c:.5f}") for k, (e, c) in ranked[:8]])
    for k, (e, c) in worst.items():
        # END TO END (secondary): the prompt gradients of the first encoder stage have crossed 23 stages of backward on top
        # of the forward drift of two bf16-storing chains; the random-init network moves its own gradients by 3-12 % under
        # 2^-9 input noise (round 1, tests/aux/grad_report.py).  The primary pin is the restarted per-block backward below.
        assert c > 0.99 and e < 0.2, (k, e, c)
    import stagewise
    lines = []
    res = stagewise.stagewise_block_backward(conf, sd, x, gout, report=lines.append)
    print("[stagewise backward cfg2 96^3] restarted per block, rel-L2 vs the rounding-aware oracle\n" + "\n".join(lines))
    assert len(res) == 12 and sum("dprompt" in v for v in res.values()) == 6
    for name, errs in res.items():
        for what, e in errs.items():
            assert e < BWD_BAR.get(what, BWD_PARAM_BAR), (name, what, e)


def test_config3_reduced_size_oracle_parity():
    """configs[3]'s model (4 input channels, encoder + decoder prompts, window 7) at 64^3 so that the CPU oracle's
    forward + backward stays at a few seconds: logits and every trainable gradient."""
    from mivp_amd import train
    from oracle.unetr_ref import OracleSwinUnetR, random_state
    conf, _, _ = train.make_conf("cfg3")
    size = 64
    sd = round_weights(random_state(conf, seed=4))
    gen = torch.Generator().manual_seed(10)
    x = torch.rand(1, 4, size, size, size, generator=gen)
    gout = torch.randn(1, conf.output_channels_downstream, size, size, size, generator=gen) / size ** 1.5
    osd = {k: v.clone() for k, v in sd.items()}
    orc = OracleSwinUnetR(conf, osd, emulate_bf16=True)
    keys = orc.trainable_keys()
    for k in keys:
        osd[k].requires_grad_(True)
    want, _ = orc(x, training=True)
    (want["downstream"] * gout).sum().backward()
    model = _product(conf, sd, x)
    out = model(x.to(DEV))["downstream"]
    (out * gout.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    e16 = rel_l2(out.detach().cpu(), want["downstream"].detach())
    print(f"[cfg3 @64^3] logits rel-L2 vs rounding-aware oracle {e16:.3e}")
    assert e16 < E2E_EMUL, e16
    _check_stagewise(conf, sd, x, "cfg3 64^3")
    params = dict(model.named_parameters())
    assert sorted(k for k, q in params.items() if q.requires_grad) == sorted(keys)
    worst = {k: (rel_l2(params[k].grad.cpu(), osd[k].grad), _cos(params[k].grad.cpu(), osd[k].grad)) for k in keys}
    ranked = sorted(worst.items(), key=lambda kv: -kv[1][0])
    print("[cfg3 @64^3] worst gradient errors:", [(k, f"{e:.2e}", f"{c:.4f}") for k, (e, c) in ranked[:6]])
    for k, (e, c) in worst.items():
        assert c > 0.99 and e < 0.2, (k, e, c)                 # end to end: see test_full_size_96_oracle_parity
    import stagewise
    lines = []
    res = stagewise.stagewise_block_backward(conf, sd, x, gout, report=lines.append)
    print("[stagewise backward cfg3 64^3]\n" + "\n".join(lines))
    assert len(res) == 12 and sum("dprompt" in v for v in res.values()) == 12
    for name, errs in res.items():
        for what, e in errs.items():
            assert e < BWD_BAR.get(what, BWD_PARAM_BAR), (name, what, e)


@pytest.fixture
def fp8_attention(request):
    from mivp_amd import swin_ops
    old = swin_ops.USE_FP8_ATTN_FWD
    swin_ops.USE_FP8_ATTN_FWD = bool(request.param)
    yield bool(request.param)
    swin_ops.USE_FP8_ATTN_FWD = old


@pytest.mark.parametrize("workload,batch,fp8_attention", [("cfg3", 2, False), ("cfg4", 8, False), ("cfg4", 8, True)],
                         indirect=["fp8_attention"])
def test_config3_config4_full_size_properties(workload, batch, fp8_attention):
    """configs[3] at 4-ch 128^3 and configs[4]'s shape (96^3, batch 8, encoder prompts) through size-independent
    properties: eval-mode volumes are independent of their batch neighbours (flipping the batch flips the output bit for
    bit, sub-batches agree to rounding), and a training step produces a finite loss and finite gradients for exactly the
    reference's downstream parameter partition.  configs[4] runs twice: in bf16 (the shipped default) and with its NAMED
    arithmetic, the E4M3 MFMA window-attention forward (``swin_ops.USE_FP8_ATTN_FWD``; per-tensor scales, so a sub-batch is
    quantised on another grid than the full batch: same bar, measured 2-3x the bf16 figure)."""
    from mivp_amd import train
    from mivp_amd.swin_unetr import SwinUnetR
    conf, size, _ = train.make_conf(workload)
    torch.manual_seed(2)
    model = SwinUnetR(conf).to(DEV)
    x, y = train.synthetic_batch(conf, batch, size, DEV)
    model.eval()
    with torch.no_grad():
        full = model(x)["downstream"]
        flipped = model(x.flip(0))["downstream"]
        first = model(x[:1])["downstream"]
    torch.cuda.synchronize()
    assert full.shape == (batch, conf.output_channels_downstream, size, size, size) and torch.isfinite(full).all()
    assert torch.equal(flipped, full.flip(0))
    assert rel_l2(full[:1].cpu(), first.cpu()) < 1e-2
    assert float((full[0] - full[1]).abs().max()) > 0
    del full, flipped, first
    model.train()
    opt = train.build_optimizer(model, conf)
    before = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
    loss = train.train_step(model, opt, conf, x, y)
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    want = sorted(n for n, _ in model.named_parameters_downstream())
    moved = [k for k, p in model.named_parameters() if p.requires_grad and not torch.equal(p.detach(), before[k])]
    assert len(moved) == len(want) == len(before)
    for _, p in model.named_parameters_downstream():
        assert p.grad is not None and torch.isfinite(p.grad).all()


def test_second_step_sees_the_first_steps_weights():
    """``torch.optim.AdamW(fused=True)`` does not bump ``p._version``, so a cache keyed on it would keep
    feeding step-0 weights to the kernels.  Two optimizer steps in supervised_learning_all at a large learning rate, then
    the forward must match the oracle loaded with the POST-step state -- and must not match the pre-step state."""
    from mivp_amd import train
    from mivp_amd.swin_unetr import SwinUnetR
    from oracle.unetr_ref import OracleSwinUnetR
    conf, _, _ = train.make_conf("tiny")
    conf.training_mode = "supervised_learning_all"
    conf.use_encoder_prompting = conf.use_decoder_prompting = False
    conf.lr_students_teacher = 2e-2
    size, batch = 32, 2
    torch.manual_seed(5)
    model = SwinUnetR(conf).to(DEV).train()
    x, y = train.synthetic_batch(conf, batch, size, DEV)
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = train.build_optimizer(model, conf)
    for _ in range(2):
        train.train_step(model, opt, conf, x, y)
    torch.cuda.synchronize()
    sd2 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    out = model(x)
    torch.cuda.synchronize()
    want_new, _ = OracleSwinUnetR(conf, round_weights(sd2), emulate_bf16=True)(x.cpu(), training=True)
    want_old, _ = OracleSwinUnetR(conf, round_weights(sd0), emulate_bf16=True)(x.cpu(), training=True)
    for key in ("latent_outputs", "seg_pred"):
        got = out[key].detach().float().cpu()
        e_new, e_old = rel_l2(got, want_new[key]), rel_l2(got, want_old[key])
        print(f"[two-step] {key}: vs post-step oracle {e_new:.3e}, vs pre-step oracle {e_old:.3e}")
        assert e_new < 1.5e-2, (key, e_new)
        assert e_old > 10 * e_new, (key, e_old, e_new)          # the test has power: stale weights would show


def test_dice_difference_to_the_oracle_after_a_few_steps():
    """North star: "Dice within 1e-4" -- asserted on the mean over sixteen unseen volumes (5e-4 per volume).  At random init the two-class logits are near-tied everywhere (the reference's own
    bf16 path agrees with its fp32 path on only 99.5-99.7 % of the voxels), so Dice is compared on a model a
    few optimisation steps in: a 96^3 volume with a bright blob whose mask the head learns, prompt
    tuning (configs[2]) for 80 steps on the HIP path, then HIP logits vs oracle logits on the trained state.  Dice =
    ``DiceCoefficient`` of the reference (utils.py:41-64, restated in oracle/loss_ref.py).  The oracle is the
    fp32 restatement of the reference (no rounding emulation: the strict reading).  Evaluated on the training volume and
    sixteen unseen noise realisations; the bar applies to the mean (the validation metric), with a per-volume sanity bound."""
    from mivp_amd import train
    from mivp_amd.swin_unetr import SwinUnetR
    from oracle.unetr_ref import OracleSwinUnetR
    from oracle.loss_ref import dice_coefficient
    conf, size, _ = train.make_conf("cfg2")
    conf.lr_downstream = 1e-2
    torch.manual_seed(11)
    model = SwinUnetR(conf).to(DEV).train()
    gen = torch.Generator().manual_seed(12)
    ax = torch.arange(size, dtype=torch.float32)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    blob = (((zz - 40) / 22) ** 2 + ((yy - 52) / 16) ** 2 + ((xx - 44) / 26) ** 2 < 1).float()
    blob = torch.maximum(blob, (((zz - 70) / 9) ** 2 + ((yy - 24) / 12) ** 2 + ((xx - 72) / 10) ** 2 < 1).float())
    img = (0.25 * torch.rand(size, size, size, generator=gen) + 0.2 + 0.5 * blob).clamp(0, 1)
    x = img[None, None].to(DEV)
    y = blob[None, None].contiguous().to(DEV)
    opt = train.build_optimizer(model, conf)
    losses = [float(train.train_step(model, opt, conf, x, y)) for _ in range(int(os.environ.get("MIVP_DICE_STEPS", "80")))]
    torch.cuda.synchronize()
    assert losses[-1] < losses[0]
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.eval()
    orc = OracleSwinUnetR(conf, round_weights(sd))
    diffs = []
    seeds = (12,) + tuple(range(101, 117))                     # the training volume, then sixteen unseen noise realisations
    for seed in seeds:
        gen = torch.Generator().manual_seed(seed)
        img = (0.25 * torch.rand(size, size, size, generator=gen) + 0.2 + 0.5 * blob).clamp(0, 1)
        xe = img[None, None].to(DEV)
        with torch.no_grad():
            got = model(xe)["downstream"].float().cpu()
            want, _ = orc(xe.cpu(), training=False)
        want = want["downstream"]
        d_hip = float(dice_coefficient(got, y.cpu(), conf.output_channels_downstream))
        d_ref = float(dice_coefficient(want, y.cpu(), conf.output_channels_downstream))
        agree = float((got.argmax(1) == want.argmax(1)).float().mean())
        print(f"[dice] seed {seed}: HIP {d_hip:.6f}  oracle {d_ref:.6f}  diff {d_hip - d_ref:+.2e}  argmax agreement {agree:.6f}  "
              f"loss {losses[0]:.4f} -> {losses[-1]:.4f}  logits rel-L2 {rel_l2(got, want):.3e}")
        assert d_ref > 0.6                                     # the segmentation is non-trivial on both sides
        diffs.append(d_hip - d_ref)
    # "Dice" is the validation metric: the mean over the evaluation volumes (the reference averages DiceCoefficient over its
    # validation loader, segmentation.py:204-300).  Per volume the two paths' bf16-vs-fp32 logit differences (~4.7e-3 rel-L2
    # end to end) flip ~1e-4 of the voxels either way: single-volume differences scatter by about +-2e-4 (round 2: the
    # six-volume mean moved between +1.8e-5 and +1.2e-4 from build to build: trajectory noise).  Round 3 evaluates sixteen
    # UNSEEN volumes and asserts the north star's own figure on their mean; the per-volume bound stays.
    unseen = diffs[1:]
    mean_diff = sum(unseen) / len(unseen)
    print(f"[dice] mean signed difference over {len(unseen)} unseen volumes {mean_diff:+.2e} (training volume {diffs[0]:+.1e}); "
          f"per volume {[f'{v:+.1e}' for v in unseen]}")
    assert abs(mean_diff) <= 1e-4, diffs
    assert max(abs(v) for v in diffs) <= 5e-4, diffs
