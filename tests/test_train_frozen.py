"""Deterministic whole-step gradient check: the discrete decisions of the step (the SEAM percentile edge mask,
tools/speed/ddrnet_speed.py:282-338 + PDF eq.1; the OHEM selection threshold, losses/ohem_cross_entropy_loss.py:
83-89) are taken from the oracle's pass and FROZEN in both the oracle and the product, so that what is compared is
the chain rule through the whole network and not which pixel a 1e-7 perturbation selects.  Then every parameter's
gradient must agree per parameter (relative L2), instead of the distribution bound of test_train.py.

Stated tolerance (measured x ~2, profiles/r03_frozen_step_gradients.txt; parameters holding > 1e-3 of the gradient norm):
  f32 activations (VALU kernels): per-parameter rel-L2 <= 5e-2 on the emulator (batch 3: measured worst 2.3e-2, median
      1.2e-2; the head's parameters 1e-5, the rise to 1e-2 happens in the backward through Muti_AFF, whose global
      branch normalises N = 3 values per channel) and <= 2.5e-2 on the MI355X (batch 8: 5.0e-3 and 1.0e-2 in two runs).  What is left is not
      arithmetic error: every block reproduces its golden gradients to 1e-6 (tests/test_train.py); a ReLU / PReLU
      pre-activation within 1e-6 of zero picks the other branch under another f32 summation order, and a fraction f of
      flipped derivative masks is a relative L2 error of sqrt(f) -- the ORACLE moves by as much against itself under
      1e-6 input noise (tools/diag_train_parity.py).
  bf16 activations (the bench path): the same effect with f ~ 2e-3 per activation layer (8-bit significands): the
      gradient the bf16 step computes is the exact gradient of the bf16-rounded forward function, whose masks differ
      from the f32 function's in 0.2 % of the elements per layer -- sqrt(f) = 4.5 % per layer, ~75 activation layers in
      series: 40-60 % per parameter against the f32 oracle at random initialisation (measured median 0.58 at batch 3;
      blocks in isolation 2-8 %, tests/test_bf16_blocks.py).  The comparison with the f32 oracle is therefore kept
      as a coarse bound on the GPU only (median <= 0.8, head parameters <= 0.25), and the tight whole-step check of
      the bf16 backward is SELF-consistency at identical masks: the same step with the gradient fan-in chains
      (EPI_RAW_ACC, dz_add / dres_add, pool / combine addends) switched off, i.e. every multi-consumer gradient
      summed by plain elementwise adds, must give the same parameter gradients to bf16 rounding (median <= 3e-2, 90th
      percentile <= 6e-2, worst <= 2e-1 with one image -- measured 1.3e-2 / 3e-2 / 9.9e-2; two images: worst 1.7e-2):
      a mis-chained or doubled addend is O(1) on every parameter upstream of it.  Emulator only: on the MI355X two
      passes of the SAME bf16 step differ by a median 25 % per parameter (f32 atomics order of the small-map
      statistics -> bf16 rounding -> activation masks), measured r3b.
  Measured on the MI355X, batch 8 (gpurun r3b): f32 median 1.6e-3, worst 5.0e-3 over 317 parameters; bf16 median 0.54,
  90th percentile 0.65, head parameters <= 0.12."""
import os

import pytest
import torch

from oracle import spec
from test_train import _randomize

_DEV = [torch.device('cpu')]


@pytest.fixture(autouse=True)
def _track_device(request):
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    yield


def D(t):
    return t.to(_DEV[0])


def _slow_on_emu():
    """the whole-step tests take 2.5-5 minutes each on the CPU emulator: there they run only with LEDN_EMU_SLOW=1
    (the -m gpu variants always run); the CPU suite keeps the block-level tests of the same kernels"""
    if _DEV[0].type == 'cpu' and not int(os.environ.get('LEDN_EMU_SLOW', '0')):
        pytest.skip('whole-step test on the emulator: set LEDN_EMU_SLOW=1')


def _frozen_step(dtype):
    import led_net_amd as L
    from led_net_amd import train as TR
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(os.path.dirname(__file__), 'data', 'lednet_test_config.py'))
    model = L.MODELS.build(cfg['model'])
    _randomize(model, 3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    nb, hw = (8, 320) if _DEV[0].type != "cpu" else (3, 320)     # (the emulator runs a smaller batch; 320 = 5 x 64:
    # the 1/64-resolution GETB reflect-pads 5 -> 8, and 256 / 64 = 4 would need a pad as large as the map)
    g = torch.Generator().manual_seed(21)
    img = torch.randint(0, 256, (nb, 3, hw, hw), dtype=torch.uint8, generator=g)
    # images of a batch differ in brightness / contrast / colour cast (as photographs do): with i.i.d. noise images
    # the per-image global pools of Muti_AFF's global_att branch are nearly EQUAL across the batch, and its
    # BatchNorm over those N values (classification/model_utils.py:369-376) divides by their tiny spread: a 1e-6
    # forward difference becomes 1e-3, which is conditioning of the test input, not of the implementation
    gain = torch.linspace(0.35, 1.0, nb).view(nb, 1, 1, 1) * (0.8 + 0.4 * torch.rand(nb, 3, 1, 1, generator=g))
    off = 90.0 * torch.rand(nb, 3, 1, 1, generator=g)
    img = (img.float() * gain + off).clamp(0, 255).to(torch.uint8)
    lab = torch.randint(0, 2, (nb, 1, hw, hw), dtype=torch.int64, generator=g)
    lab[:, :, :5, :] = 255
    lab[:, :, :, -3:] = 255
    kept = nb * hw * hw // 6
    # ---- oracle, pass 1: the decisions
    spec.TRACE = {}
    try:
        with torch.no_grad():
            spec.loss(spec.preprocess(img), lab, dict(sd), loss_cfg=((0.9, kept, 1.0), (0.9, kept, 0.4)))
        trace = spec.TRACE
    finally:
        spec.TRACE = None
    t0, t1 = trace['ohem_thr']
    frozen_cfg = ((t0, 1, 1.0), (t1, 1, 0.4))          # threshold = max(2nd smallest prob, t*) = t*
    # ---- oracle, pass 2: gradients under the frozen thresholds (the edge mask is a function of the stem: the same)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k}
    sd2 = {k: leaves.get(k, v.clone()) for k, v in sd.items()}
    want = spec.loss(spec.preprocess(img), lab, sd2, loss_cfg=frozen_cfg)
    (want['decode.loss_context'] + want['decode.loss_spatial']).backward()
    # ---- product
    for c, t in zip(cfg['model']['decode_head']['loss_decode'], (t0, t1)):
        c['thres'], c['min_kept'] = t, 1
    model = L.MODELS.build(cfg['model'])
    model.load_state_dict(sd)
    if dtype == torch.bfloat16:
        model.set_act_dtype(torch.bfloat16)
    model.to(_DEV[0])
    samples = [L.SegDataSample(gt=D(lab[i])) for i in range(nb)]
    TR.TEST_HOOKS['edge'] = trace['edge'].permute(0, 2, 3, 1).contiguous()
    try:
        tr = L.Trainer(model, cfg, max_iters=100, lr=0.0)          # lr = 0: the first step only attaches the buffers
        tr.train_step(D(img), samples)
        got = tr.forward_backward(D(img), samples)
        grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    finally:
        TR.TEST_HOOKS.pop('edge', None)
    return want, got, leaves, grads


def _product_grads_bf16(fanin_chain, monkeypatch, edge=None):
    """parameter gradients of one bf16 step of the product (no oracle), with / without the fan-in chains;
    edge: the SEAM edge mask of the other run (on the GPU the f32 atomics order of two identical forwards differs,
    and one flipped percentile pixel would be compared instead of the addend chains)"""
    import led_net_amd as L
    from led_net_amd import train as TR
    monkeypatch.setattr(TR, 'FANIN_CHAIN', fanin_chain)
    cap = {}
    monkeypatch.setitem(TR.TEST_HOOKS, 'capture', cap)
    monkeypatch.setitem(TR.TEST_HOOKS, 'edge', edge)
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(os.path.dirname(__file__), 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = 20000
    model = L.MODELS.build(cfg['model'])
    _randomize(model, 3)
    model.set_act_dtype(torch.bfloat16)
    model.to(_DEV[0])
    nb = 1          # (one image keeps the emulator run at ~2.5 minutes; every chain is exercised regardless)
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (nb, 3, 320, 320), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (nb, 1, 320, 320), dtype=torch.int64, generator=g)
    lab[:, :, :5, :] = 255
    samples = [L.SegDataSample(gt=D(lab[i])) for i in range(nb)]
    tr = L.Trainer(model, cfg, max_iters=100, lr=0.0)
    tr.train_step(D(img), samples)
    TR._Acc.counters.update(chained=0, added=0)
    out = tr.forward_backward(D(img), samples)
    counters = dict(TR._Acc.counters)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    return {k: float(v.reshape(-1)[0]) for k, v in out.items()}, grads, counters, cap['edge']


def test_bf16_fanin_chains_match_plain_adds(be, monkeypatch):
    """bf16 whole step: gradients with the fan-in addend chains == gradients with plain elementwise gradient adds
    (identical forward, identical activation masks: what differs is only where the partial gradients are added)"""
    from conftest import slow_on_emu
    slow_on_emu(_DEV[0])          # (4.5 minutes on the emulator; since round 4 the GPU twin runs the same body in deterministic mode)
    import led_net_amd as L
    # two passes of the SAME bf16 step on the GPU differ by a median 25 % per parameter in the default mode (f32 atomics
    # order of the small-map statistics -> bf16 rounding -> activation masks: measured r3b), so round 3 could run this on
    # the emulator only; deterministic mode (round 4) makes the forward bit-reproducible on the MI355X too
    L.set_deterministic(True)
    try:
        _fanin_chains_body(monkeypatch)
    finally:
        L.set_deterministic(False)


def _fanin_chains_body(monkeypatch):
    o1, g1, c1, edge = _product_grads_bf16(True, monkeypatch)
    o0, g0, c0, _ = _product_grads_bf16(False, monkeypatch, edge)
    assert c1['chained'] >= 15 and c0['chained'] == 0, (c1, c0)
    gn = max(v.norm().item() for v in g0.values())
    rows = sorted(((g1[k] - v).norm().item() / v.norm().item(), k) for k, v in g0.items() if v.norm().item() > 1e-3 * gn)
    print(f'bf16 fan-in chains vs plain adds: {len(rows)} parameters, median {rows[len(rows) // 2][0]:.2e}, worst {rows[-1]}; '
          f'forward {o1} vs {o0}')
    # same forward (the emulator is deterministic; the MI355X is in deterministic mode)
    for k in o1:
        assert abs(o1[k] - o0[k]) <= 1e-5 * abs(o0[k]) + 1e-6, (k, o1[k], o0[k])
    # (one image: the 1/64-resolution BatchNorms see 25 values per channel, their 16..64-element parameter vectors are
    #  the noisy tail; with two images the worst parameter is 1.7e-2 and the median 8.8e-3)
    med, p90 = rows[len(rows) // 2][0], rows[int(len(rows) * 0.9)][0]
    assert med <= 3e-2 and p90 <= 6e-2 and rows[-1][0] <= 2e-1, (med, p90, rows[-5:])


def _per_param(leaves, grads):
    gn = max(v.grad.norm().item() for v in leaves.values() if v.grad is not None)
    rows = []
    for k, v in leaves.items():
        if v.grad is None:
            assert k not in grads or grads[k].abs().max().item() == 0.0, k
            continue
        a, b = grads[k].flatten(), v.grad.flatten()
        if b.norm().item() < 1e-3 * gn:         # analytically-zero / negligible gradients: rounding noise
            continue
        rows.append(((a - b).norm().item() / b.norm().item(), (a @ b / (b @ b)).item(), k))
    return sorted(rows)


def test_frozen_step_gradients_f32(be):
    _slow_on_emu()
    want, got, leaves, grads = _frozen_step(torch.float32)
    for k in ('decode.loss_context', 'decode.loss_spatial'):
        a, b = float(got[k].reshape(-1)[0]), float(want[k].reshape(-1)[0])
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (k, a, b)
    rows = _per_param(leaves, grads)
    print(f'frozen step f32: {len(rows)} parameters, median rel-L2 {rows[len(rows) // 2][0]:.2e}, worst {rows[-1]}')
    assert rows[-1][0] <= (5e-2 if _DEV[0].type == 'cpu' else 2.5e-2), rows[-5:]    # MI355X, batch 8: measured 5.0e-3 and 1.0e-2 in two runs (f32 atomics order), median 1.6e-3 / 5.4e-3


def test_frozen_step_gradients_bf16(be):
    if _DEV[0].type == 'cpu':
        pytest.skip('statistical bound on a batch of 8: GPU only (the emulator runs the self-consistency test)')
    want, got, leaves, grads = _frozen_step(torch.bfloat16)
    for k in ('decode.loss_context', 'decode.loss_spatial'):
        a, b = float(got[k].reshape(-1)[0]), float(want[k].reshape(-1)[0])
        assert abs(a - b) <= 2e-2 * abs(b), (k, a, b)
    rows = _per_param(leaves, grads)
    med, p90 = rows[len(rows) // 2][0], rows[int(len(rows) * 0.9)][0]
    print(f'frozen step bf16: {len(rows)} parameters, median rel-L2 {med:.2e}, p90 {p90:.2e}, worst {rows[-1]}')
    head = [r for r in rows if r[2].startswith('decode_head.')]
    print('  head parameters worst', max(head))
    assert med <= 0.8 and max(head)[0] <= 0.25, (med, p90, max(head))
