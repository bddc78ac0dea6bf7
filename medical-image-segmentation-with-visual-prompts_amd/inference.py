"""Evaluation side of the reference's ``SegmentationTrainer.test`` on the device (segmentation.py:204-300):
the fixed sliding windows, sub-batches of ten, per-volume MeanIoU / DiceCoefficient -- with the windows cut on the device
and the metric counts accumulated by one kernel per sub-batch (no ``.item()`` inside the loop; the reference syncs
2 x classes times per update, utils.py:26-35,52-62)."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import torch

from . import _lib as L


def window_grid(image_size: Sequence[int], roi: Sequence[int]) -> Tuple[List[slice], List[int], List[int]]:
    """segmentation.py:232-241: stride = roi // 2, the volume is centre-cropped to the largest size the stride grid covers.
    Returns (crop slices, stride, windows per axis)."""
    stride = [r // 2 for r in roi]
    slc, count = [], []
    for n, w, s in zip(image_size, roi, stride):
        adjusted = (n - w) // s * s + w
        start = (n - adjusted) // 2
        slc.append(slice(start, start + adjusted))
        count.append((adjusted - w) // s + 1)
    return slc, stride, count


def sliding_window_view(x: torch.Tensor, roi: Sequence[int]) -> torch.Tensor:
    """``x [1, C, H, W, D]`` -> ``[N, C, r0, r1, r2]`` in the reference's window order (segmentation.py:242-253)."""
    if x.shape[0] != 1:
        raise ValueError("the reference's test() unfolds one volume at a time (it squeezes the batch axis)")
    slc, stride, _ = window_grid(x.shape[2:], roi)
    a = x[:, :, slc[0], slc[1], slc[2]]
    u = a.unfold(2, roi[0], stride[0]).unfold(3, roi[1], stride[1]).unfold(4, roi[2], stride[2])
    # a VIEW over the volume ([n0, n1, n2] windows of [C, roi]): with stride roi / 2 the materialised windows are ~8x the
    # volume (the reference keeps them on the CPU and moves ten at a time); ``window_batch`` copies one sub-batch
    return u.squeeze(0).permute(1, 2, 3, 0, 4, 5, 6)


def sliding_windows(x: torch.Tensor, roi: Sequence[int]) -> torch.Tensor:
    """All windows materialised, ``[N, C, roi]`` in the reference's flatten order (tests; small volumes)."""
    v = sliding_window_view(x, roi)
    return v.reshape(-1, *v.shape[3:]).contiguous()


def window_batch(win_view: torch.Tensor, begin: int, end: int) -> torch.Tensor:
    """Windows ``begin .. end`` (row-major over the window grid, the reference's flatten order) as one contiguous batch."""
    flat = win_view.reshape(-1, *win_view.shape[3:]) if win_view.is_contiguous() else None
    if flat is not None:
        return flat[begin:end]
    n0, n1, n2 = win_view.shape[:3]
    idx = torch.arange(begin, min(end, n0 * n1 * n2), device=win_view.device)
    return win_view[idx // (n1 * n2), (idx // n2) % n1, idx % n2].contiguous()


class SegMetrics:
    """MeanIoU and DiceCoefficient (utils.py:14-64) from device-resident counts."""

    def __init__(self, num_classes: int, device):
        self.num_classes = num_classes
        self.counts = torch.zeros((num_classes, 3), dtype=torch.int64, device=device)

    def reset(self):
        self.counts.zero_()

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        """preds [B, C, ...] float logits (the model's channels-first view of channels-last storage, or contiguous);
        target [B, 1, ...] float class indices."""
        if not preds.is_cuda:
            raise RuntimeError("SegMetrics runs on the GPU (the CPU restatement is oracle/loss_ref.py)")
        Cn = preds.shape[1]
        if Cn != self.num_classes:
            raise ValueError("class count mismatch")
        nd = preds.dim()
        base = preds.permute(0, *range(2, nd), 1)
        if base.is_contiguous() and preds.dtype == torch.float32:
            src, clast = base, 1
        else:
            src, clast = preds.float().contiguous(), 0
        tgt = target.float().contiguous()
        vol = 1
        for n in preds.shape[2:]:
            vol *= int(n)
        L.call("mivp_seg_counts", L.ptr(src), L.ptr(tgt), C.c_int64(preds.shape[0] * vol), C.c_int32(Cn), C.c_int32(clast),
               C.c_int64(vol), L.ptr(self.counts), L.stream())

    def compute(self) -> Tuple[float, float]:
        """(mean IoU, mean Dice) -- the one host read."""
        c = self.counts.to(torch.float64).cpu()
        inter, psum, tsum = c[:, 0], c[:, 1], c[:, 2]
        iou = (inter / (psum + tsum - inter + 1e-6)).mean()
        dice = (2 * inter / (psum + tsum + 1e-6)).mean()
        return float(iou), float(dice)


@torch.no_grad()
def test_volume(model, x: torch.Tensor, seg: torch.Tensor, roi: Sequence[int], num_classes: int, sub_batch: int = 10):
    """One volume of segmentation.py:225-286: windows, sub-batches of ten through ``model`` (eval mode), metrics over all of
    the volume's windows.  ``x [1, C, H, W, D]``, ``seg [1, 1, H, W, D]`` (already mapped to class indices).  Returns
    (mean IoU, mean Dice) of this volume."""
    dev = x.device
    xw = sliding_window_view(x, roi)
    sw = sliding_window_view(seg, roi)
    m = SegMetrics(num_classes, dev)
    n = xw.shape[0] * xw.shape[1] * xw.shape[2]
    for i in range(0, n, sub_batch):
        out = model(window_batch(xw, i, i + sub_batch))["downstream"]
        m.update(out, window_batch(sw, i, i + sub_batch))
    return m.compute()


def summarize(values: List[float]) -> Tuple[float, float]:
    """mean and (population) standard deviation over the volumes, as logged by segmentation.py:297-300."""
    mean = sum(values) / len(values)
    return mean, (sum((v - mean) ** 2 for v in values) / len(values)) ** 0.5
