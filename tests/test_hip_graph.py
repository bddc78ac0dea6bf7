"""GPU tests of the recorded-step mode (mivp_amd.train.GraphedStep): a HIP graph of forward + loss + backward + optimizer
launch, replayed, must leave the SAME bits in every parameter, buffer and optimizer state as the eager step -- the kernels
are the same kernels in the same order; what differs is who issues them.  What changes per step on the host side
(learning rate and AdamW bias corrections, the prototype loss's jitter) reaches the replay through device memory."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _same_bits(a, b):
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    return bad


def test_capturable_adamw_is_bit_equal_to_the_plain_launch():
    """FusedAdamW(capturable=True) reads lr / bias corrections from device memory (mivp_adamw_multi_dev); the plain form
    takes them as kernel arguments: same arithmetic, same bits, schedules included."""
    import mivp_amd  # noqa: F401
    from mivp_amd.optim import FusedAdamW, WarmupCosineSchedule
    g = torch.Generator().manual_seed(3)
    shapes = [(48, 48), (1030,), (3, 5, 7, 2), (2, 1024)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    pb = [torch.nn.Parameter(q.detach().clone()) for q in pa]
    grp = lambda ps: [{"params": ps[:2], "lr": 5e-3, "weight_decay": 0.1}, {"params": ps[2:], "lr": 1e-2, "weight_decay": 0.0}]
    oa, ob = FusedAdamW(grp(pa)), FusedAdamW(grp(pb), capturable=True)
    sa, sb = WarmupCosineSchedule(oa, 2, 12), WarmupCosineSchedule(ob, 2, 12)
    for step in range(6):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
    assert oa.state_dict()["state"][0]["step"] == ob.state_dict()["state"][0]["step"] == 6


@pytest.mark.parametrize("workload", ["tiny", "sup_all_small", "cfg1_small"])
def test_graphed_train_step_equals_the_eager_step(workload):
    """Six steps of the single-network trainer on a small volume: eager vs (two eager warm-up steps + four replays)."""
    import mivp_amd  # noqa: F401
    from mivp_amd import train
    from mivp_amd.swin_unetr import SwinUnetR
    if workload == "tiny":
        conf, size, batch = train.make_conf("tiny")
    elif workload == "cfg1_small":                              # frozen decoder without prompts: the concat-free up-block path
        conf, size, batch = train.make_conf("cfg1")
        size, batch = 32, 2
    else:                                                       # every parameter trains (weight-gradient kernels of all blocks)
        conf, size, batch = train.make_conf("sup_all")
        size, batch = 32, 2
    torch.manual_seed(5)
    ref = SwinUnetR(conf).to(DEV).train()
    own = copy.deepcopy(ref)
    x, y = train.synthetic_batch(conf, batch, size, DEV)
    o_ref = train.build_optimizer(ref, conf)
    o_own = train.build_optimizer(own, conf, capturable=True)
    losses_ref = [float(train.train_step(ref, o_ref, conf, x, y)) for _ in range(6)]
    step = train.graphed_train_step(own, o_own, conf, x, y, warmup=2)
    losses_own = [float(step()) for _ in range(4)]
    torch.cuda.synchronize()
    print(f"[graph {workload}] eager losses {losses_ref[2:]}, replayed {losses_own}")
    assert losses_own == losses_ref[2:]
    assert _same_bits(dict(ref.state_dict()), dict(own.state_dict())) == []
    sa, sb = o_ref.state_dict()["state"], o_own.state_dict()["state"]
    assert all(torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"])
               and float(sa[k]["step"]) == float(sb[k]["step"]) == 6 for k in sa)
    # an eager forward between replays sees the replayed parameters (the packed-weight caches were marked stale)
    own.eval(); ref.eval()
    with torch.no_grad():
        key = "downstream" if conf.training_mode == "downstream" else "seg_pred"
        assert torch.equal(own(x)[key], ref(x)[key])
    # a new batch is a copy into the recorded input tensors
    own.train(); ref.train()
    x2, y2 = train.synthetic_batch(conf, batch, size, DEV, rank=1)
    x.copy_(x2); y.copy_(y2)
    l_own = float(step())
    l_ref = float(train.train_step(ref, o_ref, conf, x, y))
    assert l_own == l_ref
    assert _same_bits(dict(ref.state_dict()), dict(own.state_dict())) == []


def test_graphed_students_teacher_step_equals_the_eager_step():
    """configs[0] (two students 32^3 / 24^3 + EMA teacher, prototype loss, AdamW with two groups under the warm-up schedule):
    five steps with a different jitter each -- eager vs (two warm-up steps + three replays); the jitter reaches the replay as
    table CONTENT (losses.JitterSlot), lr and bias corrections through the optimizer's device table."""
    import mivp_amd  # noqa: F401
    from mivp_amd import train, students_teacher as ST
    from mivp_amd.losses import ClusteredPrototypeLoss
    from mivp_amd.swin_unetr import SwinUnetR
    conf, size, batch = train.make_conf("cfg0")
    conf.warmup_steps_students_teacher = 3                      # the learning rate moves every step
    torch.manual_seed(9)
    ref = ST.MomentumModel(conf, SwinUnetR).to(DEV).train()
    ref.copy_state_dict()
    own = copy.deepcopy(ref)
    views = ST.synthetic_views(conf, batch, size, DEV, student_sizes=[32, 24])
    jit = [[[1, 0, 2, 1, 0, 3], [0, 2, 1, 1, 3, 0]], [[0, 0, 0, 0, 0, 0], [3, 3, 3, 3, 3, 3]], [[2, 1, 0, 3, 1, 2], [1, 1, 2, 0, 0, 1]],
           [[3, 0, 0, 2, 2, 1], [0, 0, 1, 3, 2, 2]], [[1, 1, 1, 1, 1, 1], [2, 0, 3, 1, 0, 2]]]
    mk = lambda static: ClusteredPrototypeLoss(float(conf.reduction_factor), int(conf.k_means_iterations), float(conf.fwhm),
                                               static_jitter=static)
    o_ref = train.build_optimizer(ref, conf)
    s_ref = train.build_scheduler(o_ref, conf)
    l_ref = [float(ST.students_teacher_step(ref, o_ref, s_ref, mk(False), conf, views, jitters=j)) for j in jit]
    o_own = train.build_optimizer(own, conf, capturable=True)
    s_own = train.build_scheduler(o_own, conf)
    feed = iter(jit)
    step = ST.graphed_students_teacher_step(own, o_own, s_own, mk(True), conf, views, jitters=lambda: next(feed), warmup=2)
    l_own = [float(step()) for _ in range(3)]
    torch.cuda.synchronize()
    print(f"[graph cfg0] eager losses {l_ref[2:]}, replayed {l_own}")
    assert l_own == l_ref[2:]
    assert o_own.param_groups[0]["lr"] == o_ref.param_groups[0]["lr"]
    assert _same_bits(dict(ref.state_dict()), dict(own.state_dict())) == []
    # after replays an EAGER forward of the (frozen, EMA-rewritten) teacher uses the current weights
    with torch.no_grad():
        fresh = SwinUnetR(conf).to(DEV).train()
        fresh.load_state_dict(own.net_teacher.state_dict(), strict=True)
        a = own.net_teacher(views["image"])["latent_outputs"]
        b = fresh(views["image"])["latent_outputs"]
    assert torch.equal(a, b)


def test_recording_refuses_plain_optimizers():
    import mivp_amd  # noqa: F401
    from mivp_amd import train
    from mivp_amd.swin_unetr import SwinUnetR
    conf, size, batch = train.make_conf("tiny")
    model = SwinUnetR(conf).to(DEV).train()
    x, y = train.synthetic_batch(conf, batch, size, DEV)
    with pytest.raises(ValueError, match="capturable"):
        train.graphed_train_step(model, train.build_optimizer(model, conf), conf, x, y)
    torch.cuda.synchronize()
    # the failed recording left the process usable
    assert torch.isfinite(train.train_step(model, train.build_optimizer(model, conf), conf, x, y))


def test_graphed_step_draws_fresh_dropout_masks_every_replay():
    """The yml's default ``attn_drop = proj_drop = 0.1`` (example_configs.yml:18-19) inside a recorded step:
    the host-drawn seeds are frozen with the descriptors, the recording increments the device's dropout epoch word and the
    kernels fold it into their seeds.  With a zero learning rate the loss of a replay depends on the masks alone:
    * consecutive replays give different losses (fresh masks);
    * a replay is a pure function of the epoch word: resetting the word reproduces a loss bit for bit;
    * forward and backward of one replay use the same masks: gradients of a replay equal those of an eager step that is
      forced onto the same seeds and epoch (checked through the loss AND the prompt gradients)."""
    import mivp_amd  # noqa: F401
    from mivp_amd import train, functional as Fn
    from mivp_amd.swin_unetr import SwinUnetR
    conf, size, batch = train.make_conf("tiny", dropout=0.3)
    conf.lr_downstream = 0.0
    conf.lr_prompt_tokens = 0.0
    conf.weight_decay_downstream = conf.weight_decay_prompt_tokens = 0.0
    torch.manual_seed(5)
    model = SwinUnetR(conf).to(DEV).train()
    x, y = train.synthetic_batch(conf, batch, size, DEV)
    opt = train.build_optimizer(model, conf, capturable=True)
    step = train.graphed_train_step(model, opt, conf, x, y, warmup=1)
    ep = Fn.dropout_epoch(torch.device(DEV))
    ep.fill_(100)
    l1 = float(step()); g1 = torch.cat([p.grad.reshape(-1).clone() for p in step.params])
    l2 = float(step()); g2 = torch.cat([p.grad.reshape(-1).clone() for p in step.params])
    l3 = float(step())
    assert int(ep.item()) == 103
    assert len({l1, l2, l3}) == 3, (l1, l2, l3)                   # three replays, three sets of masks
    assert not torch.equal(g1, g2)
    ep.fill_(100)
    l1b = float(step()); g1b = torch.cat([p.grad.reshape(-1).clone() for p in step.params])
    assert l1b == l1 and torch.equal(g1b, g1)                     # the epoch word alone decides the masks
    # the masks really are dropout at the configured rate: the loss scatters around the dropout-free loss, which it never equals
    model.eval()
    with torch.no_grad():
        l_eval = float(train.step_loss(model(x), conf, y))
    assert all(abs(l - l_eval) > 0 for l in (l1, l2, l3)) and all(abs(l - l_eval) < 0.5 * max(1.0, abs(l_eval)) for l in (l1, l2, l3))
