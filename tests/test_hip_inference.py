"""GPU tests of the evaluation / checkpoint rows: on-device metric counts against the fixture the
reference's MeanIoU / DiceCoefficient produced, the sliding-window ``test()`` loop against the oracle, and a training run
that is saved, resumed and continued bit-identically."""
import pytest
import torch

from conftest import load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("ncls", [2, 5])
def test_metric_counts_match_reference_fixture(ncls):
    import mivp_amd  # noqa: F401
    from mivp_amd.inference import SegMetrics
    fx = load_fixture("utils_metrics_schedule")[f"metrics{ncls}"]
    m = SegMetrics(ncls, DEV)
    for step in range(2):
        preds, target = fx[f"preds{step}"].to(DEV), fx[f"target{step}"].to(DEV)
        if step == 0:
            m.update(preds, target)                                             # contiguous channels-first
        else:
            cl = preds.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)   # the model's view of channels-last storage
            m.update(cl, target)
    iou, dice = m.compute()
    assert abs(iou - float(fx["miou"])) < 1e-6 and abs(dice - float(fx["dice"])) < 1e-6
    m.reset()
    assert int(m.counts.sum()) == 0


def test_sliding_window_evaluation_against_oracle():
    """SegmentationTrainer.test's loop (segmentation.py:204-300) on one synthetic volume: windows of roi with half-roi
    stride, sub-batches of ten, eval-mode model, MeanIoU / Dice over all windows -- HIP model + device counts against the
    oracle model + oracle/loss_ref metrics on the same windows."""
    import mivp_amd  # noqa: F401
    from mivp_amd import train, inference as I
    from mivp_amd.swin_unetr import SwinUnetR
    from oracle.unetr_ref import OracleSwinUnetR
    from oracle.loss_ref import dice_coefficient, mean_iou
    from test_hip_configs import round_weights
    conf, _, _ = train.make_conf("tiny")
    torch.manual_seed(4)
    model = SwinUnetR(conf)
    sd = round_weights({k: v.clone() for k, v in model.state_dict().items()})
    sd["extra_heads.downstream.1.bias"] = torch.tensor([0.3, -0.3])              # a margin, so that arg-max ties do not decide
    model.load_state_dict(sd)
    model.to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1, 1, 56, 48, 40, generator=g)
    seg = torch.randint(0, 2, (1, 1, 56, 48, 40), generator=g).float()
    roi = (32, 32, 32)
    iou, dice = I.test_volume(model, x.to(DEV), seg.to(DEV), roi, 2)
    xw, sw = I.sliding_windows(x, roi), I.sliding_windows(seg, roi)
    assert xw.shape[0] == 2 * 2 * 1
    want, _ = OracleSwinUnetR(conf, sd, emulate_bf16=True)(xw, training=False)
    w_iou, w_dice = float(mean_iou(want["downstream"], sw, 2)), float(dice_coefficient(want["downstream"], sw, 2))
    print(f"[test()] IoU {iou:.5f} / oracle {w_iou:.5f}; Dice {dice:.5f} / oracle {w_dice:.5f}")
    assert abs(iou - w_iou) < 2e-3 and abs(dice - w_dice) < 2e-3                  # random-init logits: near-ties flip a few voxels


def test_save_resume_continues_bit_identically(tmp_path):
    """Three steps, checkpoint, two more steps  ==  resume from the checkpoint in fresh objects + the same two steps: the
    FusedAdamW / scheduler / BatchNorm running statistics all travel through the reference's checkpoint dict."""
    import mivp_amd  # noqa: F401
    from mivp_amd import train, checkpoint as CK
    from mivp_amd.optim import WarmupCosineSchedule
    from mivp_amd.swin_unetr import SwinUnetR
    conf, size, batch = train.make_conf("tiny")

    def fresh():
        torch.manual_seed(9)
        m = SwinUnetR(conf).to(DEV).train()
        o = train.build_optimizer(m, conf)
        return m, o, WarmupCosineSchedule(o, 2, 20)

    x, y = train.synthetic_batch(conf, batch, size, DEV)
    m1, o1, s1 = fresh()
    for _ in range(3):
        train.train_step(m1, o1, conf, x, y); s1.step()
    path = CK.save_checkpoint(tmp_path, 0, m1, o1, s1)
    tail1 = []
    for _ in range(2):
        tail1.append(float(train.train_step(m1, o1, conf, x, y))); s1.step()
    m2, o2, s2 = fresh()
    assert CK.resume(path, m2, o2, s2, map_location=DEV) == 1
    tail2 = []
    for _ in range(2):
        tail2.append(float(train.train_step(m2, o2, conf, x, y))); s2.step()
    torch.cuda.synchronize()
    assert tail1 == tail2
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
