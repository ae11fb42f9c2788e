"""Evaluation histograms / IoU metrics: oracle vs the golden vectors generated
from the reference's IoUMetric (tests/golden/gen_golden_iou.py), HIP kernel vs oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import spec

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
_DEV = [torch.device('cpu')]


@pytest.fixture(autouse=True)
def _track_device(request):
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    yield


def D(t):
    return t.to(_DEV[0])


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return z, json.loads(str(z['meta']))


@pytest.mark.parametrize('name', ['g13_iou_c2', 'g13_iou_c19'])
def test_oracle_iou_vs_reference_golden(name):
    z, meta = _load(name)
    C = meta['num_classes']
    tot = None
    for i in range(meta['n_images']):
        res = spec.intersect_and_union(torch.from_numpy(z[f'in/pred{i}']), torch.from_numpy(z[f'in/label{i}']), C, 255)
        for k, v in zip(('intersect', 'union', 'pred_area', 'label_area'), res):
            assert np.array_equal(v.numpy(), z[f'out/{k}{i}']), (name, i, k)       # integer counts: exact
        tot = res if tot is None else tuple(a + b for a, b in zip(tot, res))
    met = spec.total_area_to_metrics(*tot)
    for k, v in met.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float64), z['out/metric_' + k], rtol=1e-6, equal_nan=True)


@pytest.mark.parametrize('name', ['g13_iou_c2', 'g13_iou_c19'])
def test_iou_hist_kernel_golden(be, name):
    """ledn_iou_hist on the golden inputs: bit-exact counts; IoUMetric end to end gives the reference's
    per-class metrics."""
    import led_net_amd as L
    from led_net_amd import metrics as M
    z, meta = _load(name)
    C = meta['num_classes']
    metric = L.IoUMetric(C, 255, ['mIoU', 'mDice', 'mFscore'])
    for i in range(meta['n_images']):
        pred, label = D(torch.from_numpy(z[f'in/pred{i}'])), D(torch.from_numpy(z[f'in/label{i}']))
        res = M.intersect_and_union(pred, label, C, 255)
        for k, v in zip(('intersect', 'union', 'pred_area', 'label_area'), res):
            assert np.array_equal(v.cpu().numpy(), z[f'out/{k}{i}']), (name, i, k)
        metric.process(pred[None], label[None])
    summary, per_class = metric.compute_metrics()
    for k, v in per_class.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float64), z['out/metric_' + k], rtol=1e-6, equal_nan=True)
    assert summary['mIoU'] == round(float(np.nanmean(z['out/metric_IoU'])) * 100, 2)


@pytest.mark.parametrize('C,shape,p_ign', [(2, (3, 257, 131), 0.1), (19, (2, 64, 96), 0.5), (5, (1, 33, 7), 1.0),
                                           (150, (1, 300, 500), 0.0)])
def test_iou_hist_kernel_vs_oracle(be, C, shape, p_ign):
    """ragged sizes, everything ignored (all-zero histograms), stray labels >= num_classes (dropped like
    torch.histc drops them), 150 classes, accumulation across calls."""
    from led_net_amd import metrics as M
    g = torch.Generator().manual_seed(C)
    pred = torch.randint(0, C, shape, generator=g).to(torch.uint8)
    label = torch.randint(0, C + 2, shape, generator=g)         # C and C+1: out of range, not ignored
    label[torch.rand(shape, generator=g) < p_ign] = 255
    want = spec.intersect_and_union(pred, label, C, 255)
    acc = D(torch.zeros((3, C)))
    got = M.intersect_and_union(D(pred), D(label), C, 255, out=acc)
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
    M.intersect_and_union(D(pred), D(label), C, 255, out=acc)    # accumulates
    assert torch.equal(acc[1].cpu(), 2 * want[2])


def test_iou_hist_rejects_bad_arguments(be):
    from led_net_amd import metrics as M, ops
    with pytest.raises(ops.LednError):
        M.intersect_and_union(D(torch.zeros(4, dtype=torch.int64)), D(torch.zeros(4, dtype=torch.int64)), 2)
    with pytest.raises(ops.LednError):
        M.intersect_and_union(D(torch.zeros(4, dtype=torch.uint8)), D(torch.zeros(5, dtype=torch.int64)), 2)
