"""Evaluation step right after the hot path: the reference's IoUMetric
(mmseg/evaluation/metrics/iou_metric.py) on the device.  `process` consumes the uint8 argmax masks that
the fused last resize of LEDHead.predict writes (segmentor.predict -> pred_sem_seg) and the label maps,
one histogram kernel per batch (ledn_iou_hist); `compute_metrics` is the reference's O(classes) host
arithmetic (total_area_to_metrics, :201-296) on the four accumulated histograms."""
from collections import OrderedDict

import torch

from . import _lib
from .ops import LednError, _check, _p, _run


def intersect_and_union(pred_label, label, num_classes, ignore_index=255, out=None):
    """IoUMetric.intersect_and_union (:163-199) for a whole batch: pred_label uint8 [..], label int64
    [..] -> (area_intersect, area_union, area_pred_label, area_label), float32 [num_classes] on the
    device.  out: optional [3, num_classes] f32 accumulator (zeroed by the caller) to keep summing
    across batches on the device."""
    lib = _lib.get_lib()
    if pred_label.dtype != torch.uint8 or label.dtype != torch.int64:
        raise LednError('intersect_and_union: uint8 prediction and int64 label expected')
    if pred_label.numel() != label.numel() or not 0 < num_classes <= 256:
        raise LednError('intersect_and_union: shape mismatch or num_classes out of range')
    pred_label, label = pred_label.contiguous(), label.contiguous()
    hist = out if out is not None else torch.zeros((3, num_classes), dtype=torch.float32, device=label.device)
    if tuple(hist.shape) != (3, num_classes) or hist.dtype != torch.float32:
        raise LednError('intersect_and_union: accumulator must be float32 [3, num_classes]')
    _check(lib, pred_label, label, hist)
    _run(lib, 'ledn_iou_hist', label, _p(pred_label), _p(label), label.numel(), num_classes, ignore_index, _p(hist))
    return hist[0], hist[1] + hist[2] - hist[0], hist[1], hist[2]


def total_area_to_metrics(inter, union, pred, label, metrics=('mIoU',), nan_to_num=None, beta=1):
    """IoUMetric.total_area_to_metrics (:201-296), float32 torch arithmetic as in the reference."""
    if isinstance(metrics, str):
        metrics = [metrics]
    if not set(metrics) <= {'mIoU', 'mDice', 'mFscore'}:
        raise KeyError(f'metrics {metrics} is not supported')
    inter, union, pred, label = (t.detach().float().cpu() for t in (inter, union, pred, label))
    ret = OrderedDict(aAcc=inter.sum() / label.sum())
    prec, rec = inter / pred, inter / label
    fscore = (1 + beta ** 2) * (prec * rec) / ((beta ** 2 * prec) + rec)
    for m in metrics:
        if m == 'mIoU':
            ret['IoU'], ret['Acc'] = inter / union, inter / label
        elif m == 'mDice':
            ret['Dice'], ret['Acc'] = 2 * inter / (pred + label), inter / label
        else:
            ret['Fscore'], ret['Precision'], ret['Recall'] = fscore, prec, rec
    ret = {k: v.numpy() for k, v in ret.items()}
    if nan_to_num is not None:
        import numpy as np
        ret = {k: np.nan_to_num(v, nan=nan_to_num) for k, v in ret.items()}
    return ret


class IoUMetric:
    """Same constructor arguments and result keys as the reference's IoUMetric (:18-61,102-157):
    aAcc / mIoU / mAcc (/ mDice / mFscore / mPrecision / mRecall) in percent, rounded to 2 digits."""

    def __init__(self, num_classes, ignore_index=255, iou_metrics=('mIoU',), nan_to_num=None, beta=1):
        self.num_classes, self.ignore_index = num_classes, ignore_index
        self.metrics = [iou_metrics] if isinstance(iou_metrics, str) else list(iou_metrics)
        self.nan_to_num, self.beta = nan_to_num, beta
        self.hist = None

    def process(self, pred_masks, labels):
        """pred_masks: uint8 [N,H,W] (or list of [H,W] / [1,H,W]); labels: int64 of the same sizes."""
        if isinstance(pred_masks, (list, tuple)):
            pred_masks = torch.stack([p.reshape(p.shape[-2:]) for p in pred_masks])
            labels = torch.stack([l.reshape(l.shape[-2:]) for l in labels])
        if self.hist is None:
            self.hist = torch.zeros((3, self.num_classes), dtype=torch.float32, device=labels.device)
        intersect_and_union(pred_masks, labels, self.num_classes, self.ignore_index, out=self.hist)

    def compute_metrics(self):
        import numpy as np
        h = self.hist
        per_class = total_area_to_metrics(h[0], h[1] + h[2] - h[0], h[1], h[2], self.metrics, self.nan_to_num,
                                          self.beta)
        out = OrderedDict()
        for k, v in per_class.items():
            out[k if k == 'aAcc' else 'm' + k] = round(float(np.nanmean(np.asarray(v, dtype=np.float64))) * 100, 2)
        return out, per_class
