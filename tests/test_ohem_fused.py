"""Both OhemCrossEntropy losses of LEDHead.loss_by_feat in one launch set (csrc/ohem_fused.hip) against (1) the
oracle's OhemCrossEntropy on the explicitly resized logits (losses/ohem_cross_entropy_loss.py:52-90 via oracle.spec:
threshold bit-exact = the k-th order statistic, selected count, loss) and (2) the two single-loss calls
(ledn_ohem_ce_up_fwd / _bwd): thresholds, counts and accuracy identical, losses and gradients to f32 rounding."""
import pytest
import torch
import torch.nn.functional as F

from oracle import spec

_DEV = [torch.device('cpu')]


@pytest.fixture(autouse=True)
def _track_device(request):
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    yield


def D(t):
    return t.to(_DEV[0])


CASES = [  # N, Hs, Ws, (min_kept0, min_kept1), (thres0, thres1), ignore pattern
    (2, 12, 14, (50, 400), (0.9, 0.7), 'border'),
    (1, 9, 10, (1, 100000), (0.6, 0.9), 'none'),          # min_kept beyond the pixel count: index clamp
    (3, 16, 8, (131072, 20), (0.9, 0.9), 'rows'),
    (1, 5, 6, (10, 10), (0.9, 0.9), 'all'),                # every pixel ignored: loss 0, nothing selected
]


def _inputs(N, Hs, Ws, ignore, seed):
    g = torch.Generator().manual_seed(seed)
    s0 = torch.randn(N, Hs, Ws, 2, generator=g) * 1.5
    s1 = torch.randn(N, Hs, Ws, 2, generator=g) * 0.7 + 0.2
    H, W = 2 * Hs, 2 * Ws
    y = torch.randint(0, 2, (N, H, W), dtype=torch.int64, generator=g)
    if ignore == 'border':
        y[:, :3], y[:, :, -2:] = 255, 255
    elif ignore == 'rows':
        y[:, ::5] = 255
    elif ignore == 'all':
        y[:] = 255
    return s0, s1, y


@pytest.mark.parametrize('N,Hs,Ws,kept,thr,ignore', CASES)
def test_fused_pair_vs_oracle_and_single_calls(be, N, Hs, Ws, kept, thr, ignore):
    from led_net_amd import ops_train as T
    s0, s1, y = _inputs(N, Hs, Ws, ignore, 7 + N)
    cfg = [(thr[0], kept[0], 1.0), (thr[1], kept[1], 0.4)]
    out, work = T.ohem2_up_fwd(D(s0), D(s1), D(y), cfg[0], cfg[1], 255)
    out = out.cpu()
    H, W = 2 * Hs, 2 * Ws
    for k, s in enumerate((s0, s1)):
        # (1) the oracle on the explicitly resized logits
        up = F.interpolate(s.permute(0, 3, 1, 2), size=(H, W), mode='bilinear', align_corners=False)
        spec.TRACE = {}
        try:
            want = spec.ohem_ce(up, y, cfg[k][0], cfg[k][1], cfg[k][2], 255)
            thr_o = spec.TRACE.get('ohem_thr', [None])[0]
        finally:
            spec.TRACE = None
        if ignore == 'all':
            assert float(out[k, 0]) == 0.0 and float(out[k, 3]) == 0.0
        else:
            assert abs(float(out[k, 0]) - float(want)) <= 2e-5 * abs(float(want)) + 1e-7, (k, out[k], want)
            # the threshold is max(k-th smallest probability, thres): equal up to the f32 rounding of the resize
            assert abs(float(out[k, 2]) - thr_o) <= 2e-6, (float(out[k, 2]), thr_o)
        # (2) the single-loss kernels on the same source
        o1, w1 = T.ohem_ce_up_fwd(D(s), D(y), cfg[k][0], cfg[k][1], cfg[k][2], 255)
        o1 = o1.cpu()
        assert float(o1[2]) == float(out[k, 2]) and float(o1[3]) == float(out[k, 3]), (o1, out[k])    # bit-exact
        if ignore != 'all':
            assert abs(float(o1[0]) - float(out[k, 0])) <= 2e-6 * abs(float(o1[0])) + 1e-8
        if k == 0:
            assert float(o1[1]) == float(out[0, 1])          # accuracy of the context output
    if ignore == 'all':
        return
    g0, g1 = torch.tensor([0.7]), torch.tensor([1.3])
    d0, d1 = T.ohem2_up_bwd(D(s0), D(s1), (H, W), work, D(out), D(g0), D(g1), cfg[0][2], cfg[1][2], 255)
    for k, (s, g, d) in enumerate(((s0, g0, d0), (s1, g1, d1))):
        o1, w1 = T.ohem_ce_up_fwd(D(s), D(y), cfg[k][0], cfg[k][1], cfg[k][2], 255)
        want = T.ohem_ce_up_bwd(D(s), D(y), w1, o1, D(g), cfg[k][2], 255)
        torch.testing.assert_close(d.cpu(), want.cpu(), rtol=1e-5, atol=1e-8)
        # and autograd through the oracle
        sr = s.clone().requires_grad_(True)
        up = F.interpolate(sr.permute(0, 3, 1, 2), size=(H, W), mode='bilinear', align_corners=False)
        (spec.ohem_ce(up, y, cfg[k][0], cfg[k][1], cfg[k][2], 255) * float(g)).backward()
        torch.testing.assert_close(d.cpu(), sr.grad, rtol=2e-4, atol=1e-7)
