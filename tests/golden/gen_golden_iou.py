#!/usr/bin/env python3
"""Golden vectors for the evaluation step right after the hot path: the
reference's IoUMetric.intersect_and_union / total_area_to_metrics
(mmseg/evaluation/metrics/iou_metric.py:163-199, 201-296), IMPORTED from /root/reference in this
container.  Only data is written (tests/golden/g13_*.npz); re-run: python tests/golden/gen_golden_iou.py

Import plumbing: parent packages are empty modules whose __path__ points at the reference
directories (no reference __init__ runs); the module-level imports of iou_metric.py that are
absent here and irrelevant to the two static methods get NAME-ONLY stand-ins: mmengine.dist
.is_main_process, mmengine.evaluator.BaseMetric, mmengine.logging.MMLogger/print_log,
mmengine.utils.mkdir_or_exist, prettytable.PrettyTable, mmseg.registry.METRICS.register_module.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _pkg(name, path=None, **names):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    for k, v in names.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    _pkg('mmseg', f'{REF}/mmseg')
    _pkg('mmseg.evaluation', f'{REF}/mmseg/evaluation')
    _pkg('mmseg.evaluation.metrics', f'{REF}/mmseg/evaluation/metrics')

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls
    _pkg('mmseg.registry', METRICS=_Reg())
    _pkg('mmengine')
    _pkg('mmengine.dist', is_main_process=lambda: True)
    _pkg('mmengine.evaluator', BaseMetric=object)
    _pkg('mmengine.logging', MMLogger=object, print_log=print)
    _pkg('mmengine.utils', mkdir_or_exist=lambda p: None)
    _pkg('prettytable', PrettyTable=object)


def main():
    install()
    from mmseg.evaluation.metrics.iou_metric import IoUMetric     # iou_metric.py:18
    g = torch.Generator().manual_seed(304)
    for name, ncls, shapes in [('g13_iou_c2', 2, [(37, 53), (64, 48)]), ('g13_iou_c19', 19, [(41, 29), (32, 80)])]:
        d, totals = {}, None
        for i, (h, w) in enumerate(shapes):
            pred = torch.randint(0, ncls, (h, w), generator=g)
            label = torch.randint(0, ncls, (h, w), generator=g)
            label[torch.rand((h, w), generator=g) < 0.15] = 255          # ignored pixels
            if i == 1:
                label[:, : w // 2] = pred[:, : w // 2]                   # a mostly-correct image
            res = IoUMetric.intersect_and_union(pred, label, ncls, 255)   # :163-199
            d[f'in/pred{i}'], d[f'in/label{i}'] = pred.numpy().astype(np.uint8), label.numpy()
            for k, v in zip(('intersect', 'union', 'pred_area', 'label_area'), res):
                d[f'out/{k}{i}'] = v.numpy()
            totals = res if totals is None else tuple(a + b for a, b in zip(totals, res))
        met = IoUMetric.total_area_to_metrics(*totals, ['mIoU', 'mDice', 'mFscore'], None, 1)   # :201-296
        for k, v in met.items():
            d['out/metric_' + k] = np.asarray(v, dtype=np.float64)
        d['meta'] = np.asarray(json.dumps(dict(kind='IoUMetric', num_classes=ncls, ignore_index=255,
                                               n_images=len(shapes), shim='names only')))
        np.savez(os.path.join(OUT, name + '.npz'), **d)
        print(name, {k: np.round(v, 4).tolist() for k, v in met.items() if k in ('aAcc', 'IoU')})


if __name__ == '__main__':
    main()
