"""GPU tests of the students/teacher side (SURVEY 8 a19 second half, 8f N1 / N2): ClusteredPrototypeLoss against the
fixtures the REFERENCE produced (tests/golden/proto_*.npz), the multi-tensor AdamW against torch.optim.AdamW, the fused EMA
against the reference's MomentumModel fixture, and one full step of BASELINE.json configs[0] (self_supervised_learning_all,
1-channel 32^3, batch 2: two students 32^3 + 24^3 and the EMA teacher) against the oracle."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import load_fixture, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tag", ["proto_a", "proto_b", "proto_c"])
def test_prototype_loss_against_reference_fixture(tag):
    """Loss value and the students' input gradients against what the reference's own ClusteredPrototypeLoss returned for the
    same embeddings, coordinates and jitter.  Gradients come back through the bf16 channels-last gradient tensor the
    sampling kernel writes (2^-9 rounding).  ``detach_teacher=False`` additionally checks the teacher-side gradient."""
    import mivp_amd  # noqa: F401
    from mivp_amd.losses import ClusteredPrototypeLoss
    fx = load_fixture(tag)
    m = fx.meta
    n = m["n_students"]
    C = fx["in"]["emb_t"].shape[1]
    for detach in (True, False):
        emb_t = fx["in"]["emb_t"].to(DEV).requires_grad_(True)
        emb_s = [fx["in"][f"emb_s{i}"].to(DEV).requires_grad_(True) for i in range(n)]
        loss_fn = ClusteredPrototypeLoss(m["reduction_factor"], m["k_means_iterations"], m["fwhm"], detach_teacher=detach)
        loss = loss_fn(emb_s, emb_t, [fx["in"][f"coord_s{i}"].to(DEV) for i in range(n)], fx["in"]["coord_t"].to(DEV),
                       temp_s=m["temp_s"], temp_t=m["temp_t"], jitters=[fx["in"][f"jitter{i}"].tolist() for i in range(n)])
        want = float(fx["out"]["loss"])
        assert abs(float(loss) - want) < 1e-4 * max(1.0, abs(want)), (float(loss), want)
        if C % 8:
            continue                                            # the sampling backward works on 8-channel groups
        loss.backward()
        torch.cuda.synchronize()
        for i in range(n):
            e = rel_l2(emb_s[i].grad.float().cpu(), fx["grad"][f"emb_s{i}"])
            assert e < 4e-3, (tag, i, e)
        if detach:
            assert emb_t.grad is None
        else:
            assert rel_l2(emb_t.grad.float().cpu(), fx["grad"]["emb_t"]) < 4e-3


def test_sampling_kernel_on_channels_last_bf16_latent():
    """The layout the model hands over: a channels-first VIEW of channels-last bf16 storage, with a jitter crop; forward
    against the oracle's index-form sampler, backward against autograd over it."""
    import mivp_amd  # noqa: F401
    from mivp_amd.losses import sample_points
    from oracle import proto_ref as P
    g = torch.Generator().manual_seed(2)
    base = torch.randn(2, 20, 16, 24, 48, generator=g).to(torch.bfloat16)          # [B, H, W, D, C]
    jit = (1, 2, 0, 3, 2, 1)
    out_dims = (5, 4, 6)
    ref_in = base.float().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    want = P.sample_volume(ref_in, out_dims, jit).flatten(2).transpose(1, 2)
    gout = torch.randn(want.shape, generator=g)
    want.backward(gout)
    vol = base.to(DEV).requires_grad_(True)
    got = sample_points(vol.permute(0, 4, 1, 2, 3), out_dims, jit)
    got.backward(gout.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(got.cpu(), want.detach()) < 1e-6
    assert rel_l2(vol.grad.float().cpu().permute(0, 4, 1, 2, 3), ref_in.grad) < 3e-3      # bf16 gradient tensor


def test_fused_adamw_matches_torch_adamw():
    """mivp_amd.optim.FusedAdamW against torch.optim.AdamW (foreach, f32 on the same device): two parameter groups with
    their own lr / weight decay, a WarmupCosineSchedule on each, eight steps, a parameter that gets no gradient; then the
    state dicts are interchangeable."""
    import mivp_amd  # noqa: F401
    from mivp_amd.optim import FusedAdamW, WarmupCosineSchedule
    g = torch.Generator().manual_seed(0)
    shapes = [(48, 48), (48,), (3, 5, 7, 2), (1030,), (2, 1024), (96, 3, 3, 3, 3), (7,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    own_p = [torch.nn.Parameter(q.detach().clone()) for q in ref_p]

    def groups(ps):
        return [{"params": ps[:4], "lr": 5e-3, "weight_decay": 0.1}, {"params": ps[4:], "lr": 1e-2, "weight_decay": 0.0}]

    ref = torch.optim.AdamW(groups(ref_p), lr=5e-3, weight_decay=0.1, foreach=True)
    own = FusedAdamW(groups(own_p), lr=5e-3, weight_decay=0.1)
    s_ref = WarmupCosineSchedule(ref, 3, 20)
    s_own = WarmupCosineSchedule(own, 3, 20)
    for step in range(8):
        for a, b in zip(ref_p, own_p):
            gr = torch.randn(a.shape, generator=g).to(DEV) * (1 + step)
            a.grad, b.grad = gr.clone(), gr.clone()
        ref_p[-1].grad = own_p[-1].grad = None                  # a parameter without gradient is skipped
        ref.step(); own.step(); s_ref.step(); s_own.step()
    torch.cuda.synchronize()
    worst = 0.0
    exact = 0
    for a, b in zip(ref_p, own_p):
        worst = max(worst, float((a - b).abs().max() / a.abs().max()))
        exact += int(torch.equal(a, b))
    print(f"[adamw] worst relative difference after 8 steps {worst:.2e}; {exact}/{len(ref_p)} tensors bit-equal")
    assert worst < 2e-6
    assert torch.equal(ref_p[-1], own_p[-1])
    sd = own.state_dict()
    other = torch.optim.AdamW(groups([torch.nn.Parameter(q.detach().clone()) for q in own_p]), lr=5e-3, weight_decay=0.1)
    other.load_state_dict(sd)                                   # AdamW's layout: step / exp_avg / exp_avg_sq
    own2 = FusedAdamW(groups(own_p), lr=5e-3, weight_decay=0.1)
    own2.load_state_dict(ref.state_dict())
    assert own2.state_dict()["param_groups"][1]["lr"] == ref.state_dict()["param_groups"][1]["lr"]


def test_fused_ema_matches_reference_momentum_model():
    """update_teacher on the device against the two EMA steps the reference's MomentumModel took (fixture), through this
    package's MomentumModel with the same toy architecture."""
    import mivp_amd  # noqa: F401
    from mivp_amd.students_teacher import MomentumModel
    fx = load_fixture("momentum_model")

    class Toy(torch.nn.Module):
        def __init__(self, conf):
            super().__init__()
            self.a = torch.nn.Linear(5, 7)
            self.b = torch.nn.Conv3d(2, 3, 3)
            self.n = torch.nn.BatchNorm3d(3)

    mm = MomentumModel(Namespace(tau=fx.meta["tau"]), Toy).to(DEV)
    order = fx.meta["param_order"]
    assert [k for k, _ in mm.net_student.named_parameters()] == order
    with torch.no_grad():
        for k, p in mm.net_student.named_parameters():
            p.copy_(fx["student0"][k])
        for k, p in mm.net_teacher.named_parameters():
            p.copy_(fx["teacher0"][k])
    ptrs = [p.data_ptr() for p in mm.net_teacher.parameters()]
    mm.update_teacher()
    for k, p in mm.net_teacher.named_parameters():
        assert torch.allclose(p.cpu(), fx["teacher1"][k], rtol=1e-6, atol=1e-7), k
    with torch.no_grad():
        for k, p in mm.net_student.named_parameters():
            p.copy_(fx["student1"][k])
    mm.update_teacher()
    for k, p in mm.net_teacher.named_parameters():
        assert torch.allclose(p.cpu(), fx["teacher2"][k], rtol=1e-6, atol=1e-7), k
    assert ptrs == [p.data_ptr() for p in mm.net_teacher.parameters()]          # in place: no re-allocation per step
    mm.copy_state_dict()
    assert all(torch.equal(a, b) for a, b in zip(mm.net_student.parameters(), mm.net_teacher.parameters()))
    assert all(not p.requires_grad for p in mm.net_teacher.parameters())


def test_frozen_teacher_forward_sees_every_ema_update():
    """``copy_state_dict()`` freezes the teacher (students_teacher.py:136) and ``update_teacher`` rewrites it
    through a raw kernel (no version bump, same storage): the packed weight images of the teacher must be rebuilt anyway.
    After an EMA update the teacher's forward equals the forward of a FRESH model loaded with the EMA'd state, bit for bit,
    and differs from its forward before the update."""
    import mivp_amd  # noqa: F401
    from mivp_amd import train, students_teacher as ST
    from mivp_amd.swin_unetr import SwinUnetR
    conf, size, batch = train.make_conf("cfg0")
    torch.manual_seed(3)
    mm = ST.MomentumModel(conf, SwinUnetR).to(DEV).train()
    mm.copy_state_dict()
    assert all(not p.requires_grad for p in mm.net_teacher.parameters())
    x = torch.rand(1, conf.input_channels, size, size, size, device=DEV)
    fwd = lambda net: {k: v.detach().clone() for k, v in net(x).items() if torch.is_tensor(v)}
    with torch.no_grad():
        y0 = fwd(mm.net_teacher)
        for p in mm.net_student.parameters():
            p.add_(0.5 * torch.randn_like(p) * p.abs().mean().clamp_min(1e-3))
        for _ in range(3):                                      # tau = 0.99..: a few updates so that bf16 images move
            mm.update_teacher()
        y1 = fwd(mm.net_teacher)
        fresh = SwinUnetR(conf).to(DEV).train()
        fresh.load_state_dict(mm.net_teacher.state_dict(), strict=True)
        y_ref = fwd(fresh)
    torch.cuda.synchronize()
    assert set(y1) == set(y_ref) and len(y1) > 0
    for k in y1:
        assert torch.equal(y1[k], y_ref[k]), f"teacher forward does not see the EMA'd weights ({k})"
    assert any(not torch.equal(y0[k], y1[k]) for k in y1), "the perturbation did not reach the output: test is vacuous"


def test_config0_students_teacher_step_against_oracle():
    """BASELINE.json configs[0] as the reference runs it (students_teacher.py:150-207): MomentumModel(conf, SwinUnetR) with
    the yml's channel widths, 1-channel 32^3, batch 2, students 32^3 and 24^3, teacher 32^3.  One step on the HIP path vs
    the oracle (rounding-aware model + oracle/proto_ref.py loss + autograd): EMA'd teacher weights, the loss value, the
    student's parameter gradients, and that the optimizer moved exactly the reference's parameter partition."""
    import mivp_amd  # noqa: F401
    from mivp_amd import train, students_teacher as ST
    from mivp_amd.losses import ClusteredPrototypeLoss
    from mivp_amd.swin_unetr import SwinUnetR
    from oracle.unetr_ref import OracleSwinUnetR, random_state
    from oracle import proto_ref as P
    from test_hip_configs import round_weights
    conf, size, batch = train.make_conf("cfg0")
    sd_s = round_weights(random_state(conf, seed=31))
    sd_t = round_weights(random_state(conf, seed=32))
    mm = ST.MomentumModel(conf, SwinUnetR)
    mm.net_student.load_state_dict(sd_s, strict=True)
    mm.net_teacher.load_state_dict(sd_t, strict=True)
    mm.to(DEV).train()
    views = ST.synthetic_views(conf, batch, size, DEV, student_sizes=[32, 24])
    assert [tuple(v.shape[2:]) for v in views["image_st"]] == [(32, 32, 32), (24, 24, 24)]
    jit = [[1, 0, 2, 1, 0, 3], [0, 2, 1, 1, 3, 0]]
    loss_fn = ClusteredPrototypeLoss(float(conf.reduction_factor), int(conf.k_means_iterations), float(conf.fwhm))
    opt = train.build_optimizer(mm, conf)
    sched = train.build_scheduler(opt, conf)
    before = {k: p.detach().clone() for k, p in mm.net_student.named_parameters()}
    # --- oracle ---
    keys = OracleSwinUnetR(conf, sd_s).trainable_keys()
    tsd = {k: v.clone() for k, v in sd_t.items()}
    for k in keys:
        tsd[k] = P.ema_update(sd_t[k], sd_s[k], float(conf.tau))
    osd = {k: v.clone() for k, v in sd_s.items()}
    for k in keys:
        osd[k].requires_grad_(True)
    orc = OracleSwinUnetR(conf, osd, emulate_bf16=True)
    outs = [orc(v.cpu(), training=True)[0]["latent_outputs"] for v in views["image_st"]]
    with torch.no_grad():
        out_t = OracleSwinUnetR(conf, round_weights(tsd), emulate_bf16=True)(views["image"].cpu(), training=True)[0]["latent_outputs"]
    want = P.clustered_prototype_loss(outs, out_t, [c.cpu() for c in views["coord_st"]], views["coord"].cpu(), jit,
                                      float(conf.reduction_factor), int(conf.k_means_iterations), float(conf.fwhm))
    want.backward()
    # --- product: one step ---
    mm.update_teacher()
    for k, p in mm.net_teacher.named_parameters():
        assert torch.allclose(p.cpu(), tsd[k], rtol=1e-6, atol=1e-7), k
    out_sts, out_tch = mm(views["image_st"], views["image"])
    assert not out_tch["latent_outputs"].requires_grad
    got = loss_fn([o["latent_outputs"] for o in out_sts], out_tch["latent_outputs"], views["coord_st"], views["coord"], jitters=jit)
    opt.zero_grad(set_to_none=True)
    got.backward()
    torch.cuda.synchronize()
    print(f"[cfg0 step] loss HIP {float(got):.6f} oracle {float(want):.6f}")
    assert abs(float(got) - float(want)) < 2e-2 * abs(float(want))
    params = dict(mm.net_student.named_parameters())
    cos = {}
    for k in keys:
        g, w = params[k].grad, osd[k].grad
        if w is None or float(w.norm()) < 1e-9:
            continue
        assert g is not None and torch.isfinite(g).all(), k
        # a per-channel constant in front of a training-mode BatchNorm (conv / mlp / LayerNorm biases of the last block of
        # a stage) has a TRUE gradient of zero: what both sides hold there is rounding noise, small against the sibling weight
        sib = osd.get(k[:-4] + "weight") if k.endswith(".bias") else None
        if sib is not None and sib.grad is not None and float(w.norm()) < 1e-3 * float(sib.grad.norm()):
            assert float(g.norm()) < 2e-2 * float(sib.grad.norm()), k
            continue
        cos[k] = float(torch.nn.functional.cosine_similarity(g.cpu().reshape(-1).double(), w.reshape(-1).double(), dim=0))
    worst = sorted(cos.items(), key=lambda kv: kv[1])[:5]
    print("[cfg0 step] lowest gradient cosines vs the oracle:", [(k, f"{c:.4f}") for k, c in worst])
    # an ill-conditioned objective end to end (soft arg-max prototype assignments on a random-init net): direction only
    assert np.median(list(cos.values())) > 0.97 and sum(c < 0.8 for c in cos.values()) <= max(2, len(cos) // 50)
    opt.step()
    sched.step()
    # the reference's schedule starts at lr = base * 0 / warmup (utils.py:81-83): the first step moves nothing
    assert all(torch.equal(p.detach(), before[k]) for k, p in mm.net_student.named_parameters())
    assert abs(opt.param_groups[0]["lr"] - float(conf.lr_students_teacher) * 1 / conf.warmup_steps_students_teacher) < 1e-12
    # the packaged step (second iteration: EMA from the student, fused optimizer at lr = base / warmup, schedule)
    l2 = ST.students_teacher_step(mm, opt, sched, loss_fn, conf, views, jitters=jit)
    torch.cuda.synchronize()
    assert torch.isfinite(l2)
    moved = {k for k, p in mm.net_student.named_parameters() if not torch.equal(p.detach(), before[k])}
    assert moved == {k for k in keys if params[k].grad is not None} and len(moved) > 200
    assert abs(opt.param_groups[0]["lr"] - float(conf.lr_students_teacher) * 2 / conf.warmup_steps_students_teacher) < 1e-12
