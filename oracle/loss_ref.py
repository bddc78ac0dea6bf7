"""Oracle (test infrastructure): training-step loss and evaluation metrics.

The reference's downstream loss is MONAI ``DiceFocalLoss(include_background,
to_onehot_y=True, softmax=True, gamma=4.0)`` (segmentation.py:44-50).  MONAI is
not in the image and the reference pins no version, so this is a restatement of
MONAI's documented formulas and is **parity unpinned** (no reference fixture
covers it):

* Dice term: softmax over channels, one-hot target, per (batch, class)
  ``1 - (2*sum(p*t) + 1e-5) / (sum(p) + sum(t) + 1e-5)``, mean over batch x class.
* Focal term: MONAI's FocalLoss defaults to the *sigmoid* form even inside
  DiceFocalLoss (``use_softmax`` defaults to False): per element
  ``bce(x,t) * exp(gamma * logsigmoid(-x*(2t-1)))``, mean over all elements.
* ``include_background=False`` drops class 0 from both terms.

Metrics restate modules/utils.py:14-64 (importing that file needs cv2, absent).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def dice_focal_loss(logits: Tensor, target: Tensor, include_background: bool = True,
                    gamma: float = 4.0) -> Tensor:
    """``logits [B,C,H,W,D]`` float, ``target [B,1,H,W,D]`` float class indices."""
    C = logits.shape[1]
    onehot = F.one_hot(target[:, 0].long(), C).permute(0, 4, 1, 2, 3).to(logits.dtype)
    prob = logits.softmax(dim=1)
    x = logits
    if not include_background:
        onehot, prob, x = onehot[:, 1:], prob[:, 1:], logits[:, 1:]
    dims = (2, 3, 4)
    inter = (prob * onehot).sum(dims)
    denom = prob.sum(dims) + onehot.sum(dims)
    dice = (1.0 - (2.0 * inter + 1e-5) / (denom + 1e-5)).mean()
    bce = x - x * onehot - F.logsigmoid(x)
    inv = F.logsigmoid(-x * (onehot * 2 - 1))
    focal = ((inv * gamma).exp() * bce).mean()
    return dice + focal


def dice_loss(logits: Tensor, target: Tensor, include_background: bool = True) -> Tensor:
    """MONAI ``DiceLoss(include_background, to_onehot_y=True, softmax=True)`` as documented -- the segmentation term of the
    students / teacher trainer's supervised modes (modules/students_teacher.py:96-100, used at :190-197): the Dice term of
    ``dice_focal_loss`` above, alone.  Parity unpinned (MONAI absent, no reference fixture)."""
    C = logits.shape[1]
    onehot = F.one_hot(target[:, 0].long(), C).permute(0, 4, 1, 2, 3).to(logits.dtype)
    prob = logits.softmax(dim=1)
    if not include_background:
        onehot, prob = onehot[:, 1:], prob[:, 1:]
    dims = (2, 3, 4)
    inter = (prob * onehot).sum(dims)
    denom = prob.sum(dims) + onehot.sum(dims)
    return (1.0 - (2.0 * inter + 1e-5) / (denom + 1e-5)).mean()


def _counts(preds: Tensor, target: Tensor, num_classes: int):
    pred = preds.argmax(dim=1, keepdim=True)
    inter, psum, tsum = [], [], []
    for c in range(num_classes):
        p = (pred == c).float()
        t = (target == c).float()
        inter.append((p * t).sum())
        psum.append(p.sum())
        tsum.append(t.sum())
    return torch.stack(inter), torch.stack(psum), torch.stack(tsum)


def dice_coefficient(preds: Tensor, target: Tensor, num_classes: int) -> Tensor:
    """modules/utils.py:41-64: arg-max prediction, per class 2*|P&T| / (|P|+|T|+1e-6), mean."""
    inter, psum, tsum = _counts(preds, target, num_classes)
    return (2 * inter / (psum + tsum + 1e-6)).mean()


def mean_iou(preds: Tensor, target: Tensor, num_classes: int) -> Tensor:
    """modules/utils.py:14-38: per class |P&T| / (|P|+|T|-|P&T|+1e-6), mean."""
    inter, psum, tsum = _counts(preds, target, num_classes)
    return (inter / (psum + tsum - inter + 1e-6)).mean()
