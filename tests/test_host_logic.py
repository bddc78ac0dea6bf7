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
    Fn.invalidate_weight_caches()
    get_t(); get_f()
    assert calls == {"t": 3, "f": 3}


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
