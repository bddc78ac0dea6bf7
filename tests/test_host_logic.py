"""CPU tests of host-side logic that needs no GPU: the weight-cache invalidation rule, schedule arithmetic."""
import torch


def test_weight_cache_sees_fused_optimizer_steps():
    """torch.optim.AdamW(fused=True) changes parameters without bumping ``_version``: the cache stamp
    carries a process-wide optimizer-step epoch for trainable parameters, frozen ones are packed once."""
    import mivp_amd
    from mivp_amd import functional as Fn
    train_p = torch.nn.Parameter(torch.randn(8))
    frozen_p = torch.nn.Parameter(torch.randn(8), requires_grad=False)
    cache = Fn.WeightCache()
    calls = {"t": 0, "f": 0}

    def get_t():
        return cache.get("t", [train_p], lambda: (calls.__setitem__("t", calls["t"] + 1), train_p.detach().clone())[1])

    def get_f():
        return cache.get("f", [frozen_p], lambda: (calls.__setitem__("f", calls["f"] + 1), frozen_p.detach().clone())[1])

    a = get_t(); get_t(); get_f(); get_f()
    assert calls == {"t": 1, "f": 1}
    opt = torch.optim.AdamW([train_p], lr=0.1, fused=True)
    train_p.grad = torch.ones(8)
    v0 = train_p._version
    opt.step()
    assert train_p._version == v0, "this torch bumps versions in fused AdamW: the epoch is then redundant, not wrong"
    b = get_t(); get_f()
    assert calls == {"t": 2, "f": 1}
    assert not torch.equal(a, b) and torch.equal(b, train_p.detach())
    # in-place edits with a version bump and storage swaps (EMA teacher: ``p.data = ...``) are seen without a step
    with torch.no_grad():
        frozen_p.add_(1.0)
    get_f()
    frozen_p.data = frozen_p.data * 0.5
    get_f()
    assert calls["f"] == 3
    # raw writes behind autograd's back (EMA teacher kernel, graph replays) invalidate FROZEN parameters' copies too
    # (the frozen teacher kept its first packed weights)
    Fn.invalidate_weight_caches()
    get_t(); get_f()
    assert calls == {"t": 3, "f": 4}
    get_t(); get_f()
    assert calls == {"t": 3, "f": 4}


def test_bn_momentum_none_is_cumulative_average():
    import mivp_amd
    from mivp_amd import functional as Fn
    bn = torch.nn.BatchNorm3d(4, momentum=None)
    assert Fn.bn_momentum(bn) == 1.0
    bn.num_batches_tracked += 3
    assert Fn.bn_momentum(bn) == 0.25
    assert Fn.bn_momentum(torch.nn.BatchNorm3d(4)) == 0.1


def test_reference_momentum_model_accepts_the_product_class():
    """Drop-in: the REFERENCE's MomentumModel (momentum_model/momentum_model.py:4-36) builds two product SwinUnetR instances
    through ``architecture(conf=conf)`` and pairs their parameters by zip order.  Runs where /root/reference is mounted (the
    build container); construction, the EMA update and copy_state_dict need no GPU."""
    import os
    import sys
    import types
    import pytest
    ref = "/root/reference/src/modules"
    if not os.path.isdir(ref):
        pytest.skip("reference not mounted")
    import mivp_amd
    from mivp_amd import train
    from mivp_amd.swin_unetr import SwinUnetR
    if "refmodules" not in sys.modules:
        parent = types.ModuleType("refmodules")
        parent.__path__ = [ref]
        sys.modules["refmodules"] = parent
    from refmodules.momentum_model import MomentumModel
    conf, _, _ = train.make_conf("tiny")
    conf.training_mode = "self_supervised_learning_all"
    mm = MomentumModel(conf, SwinUnetR)
    names_s = [k for k, _ in mm.net_student.named_parameters()]
    assert names_s == [k for k, _ in mm.net_teacher.named_parameters()] and len(names_s) > 200
    t0 = [p.detach().clone() for p in mm.net_teacher.parameters()]
    mm.update_teacher()
    for a, t, s in zip(t0, mm.net_teacher.parameters(), mm.net_student.parameters()):
        assert torch.allclose(t, conf.tau * a + (1 - conf.tau) * s.detach())
    mm.copy_state_dict()
    assert all(torch.equal(a, b) for a, b in zip(mm.net_student.parameters(), mm.net_teacher.parameters()))
    # the optimizer recipe of the reference's trainer (students_teacher.py:27-68) resolves against the product's helpers
    n = (sum(p.numel() for _, p in mm.net_student.named_parameters_decoder())
         + sum(p.numel() for _, p in mm.net_student.named_parameters_encoder()))
    assert n > 0


def test_checkpoint_wire_format_round_trip(tmp_path):
    """The reference's checkpoint dict (segmentation.py:145-154, students_teacher.py:234-244): keys, file name, epoch + 1,
    safe loading, and interchange of the optimizer state with torch.optim.AdamW (CPU: the model is only a state holder)."""
    import mivp_amd
    from mivp_amd import checkpoint as CK, train
    from mivp_amd.optim import WarmupCosineSchedule
    from mivp_amd.swin_unetr import SwinUnetR
    conf, _, _ = train.make_conf("tiny")
    torch.manual_seed(0)
    model, teacher = SwinUnetR(conf), SwinUnetR(conf)
    params = [p for _, p in model.named_parameters_downstream()]
    opt = torch.optim.AdamW(params, lr=1e-3)
    for p in params:
        p.grad = torch.randn_like(p)
    opt.step()
    sched = WarmupCosineSchedule(opt, 5, 50)
    sched.step()
    path = CK.save_checkpoint(tmp_path, 19, model, opt, sched, teacher=teacher)
    assert path.endswith("0019.pt")
    ck = CK.read_checkpoint(path)
    assert list(ck) == ["current_epoch", "model_state_dict", "teacher_state_dict", "optimizer_state_dict", "scheduler_state_dict"]
    assert ck["current_epoch"] == 20
    assert list(ck["model_state_dict"]) == list(model.state_dict())
    m2, t2 = SwinUnetR(conf), SwinUnetR(conf)
    p2 = [p for _, p in m2.named_parameters_downstream()]
    o2 = torch.optim.AdamW(p2, lr=1e-3)
    s2 = WarmupCosineSchedule(o2, 5, 50)
    assert CK.resume(path, m2, o2, s2, teacher=t2) == 20
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert all(torch.equal(a, b) for a, b in zip(teacher.state_dict().values(), t2.state_dict().values()))
    assert o2.state_dict()["state"][0]["exp_avg"].equal(opt.state_dict()["state"][0]["exp_avg"])
    assert s2.last_epoch == sched.last_epoch and o2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    # backbone load into a model with different prompting flags: the matching entries really arrive (the reference's
    # own code writes into a copy of state_dict() and loads nothing: SURVEY Appendix F)
    conf3, _, _ = train.make_conf("cfg1")
    conf3.hidden_channels, conf3.num_heads_encoder, conf3.num_heads_decoder = conf.hidden_channels, conf.num_heads_encoder, conf.num_heads_decoder
    m3 = SwinUnetR(conf3)
    n = CK.load_backbone(path, m3)
    assert 0 < n < len(model.state_dict())
    assert torch.equal(m3.state_dict()["bottleneck.weight"], model.state_dict()["bottleneck.weight"])


def test_sliding_window_grid_matches_the_reference_formulation():
    """inference.sliding_windows against the unfold chain of segmentation.py:232-253 written out on the CPU, incl. volumes
    the stride grid does not cover exactly (centre crop) and the window order."""
    import mivp_amd
    from mivp_amd import inference as I
    g = torch.Generator().manual_seed(1)
    for size, roi in (((20, 17, 9), (8, 8, 4)), ((16, 16, 8), (8, 8, 8)), ((13, 30, 11), (6, 10, 4))):
        x = torch.randn(1, 2, *size, generator=g)
        stride = [r // 2 for r in roi]
        adjusted, slc = [0, 0, 0], [None, None, None]
        for i in range(3):
            adjusted[i] = (size[i] - roi[i]) // stride[i] * stride[i] + roi[i]
            start = (size[i] - adjusted[i]) // 2
            slc[i] = slice(start, start + adjusted[i])
        ax = x[:, :, slc[0], slc[1], slc[2]]
        want = ax.unfold(2, roi[0], stride[0]).unfold(3, roi[1], stride[1]).unfold(4, roi[2], stride[2]) \
                 .flatten(2, 4).permute(2, 1, 0, 3, 4, 5).squeeze(2).contiguous()
        got = I.sliding_windows(x, roi)
        assert torch.equal(got, want)
        _, _, count = I.window_grid(size, roi)
        assert got.shape[0] == count[0] * count[1] * count[2]
    assert I.summarize([0.5, 0.7]) == (0.6, 0.09999999999999998)


def test_halo_brick_cost_model_picks_the_measured_bricks():
    """ops.halo_brick (host logic, no GPU): the brick geometry per decoder conv of the 96^3 / batch-4 model as measured in
    round 3 (tools/ab_conv_bricks.py): exact 2 x 8-tile bricks at the 12 x 12 x 24 and 6 x 6 x 24 stages, 6 x 6 x 16 where the volume
    fills three / one full rounds of them, the im2col kernel for tiny volumes."""
    import mivp_amd
    from mivp_amd import ops
    assert ops.halo_brick(4, (12, 12, 24), 192) == 66
    assert ops.halo_brick(4, (6, 6, 24), 384) == 36
    assert ops.halo_brick(4, (24, 24, 24), 96) == 6
    assert ops.halo_brick(4, (48, 48, 48), 48) == 6
    assert ops.halo_brick(1, (4, 4, 8), 48) == 0
    for code, (bh, bw, bd, tiles) in ops._HALO_BRICKS.items():
        assert bd in (8, 16) and tiles >= 1 and bh * bw * bd % 16 == 0, code
