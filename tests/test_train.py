"""Training path: blocks in train mode (batch-statistics BN, hand-written backward
kernels through autograd) against the golden vectors from the reference modules
(outputs, input grads, parameter grads, running-stat updates), then the whole
LED-Net train step against the CPU oracle."""
import math
import os

import pytest
import torch

from conftest import Fixture, golden_names
from oracle import spec

_DEV = [torch.device('cpu')]


@pytest.fixture(autouse=True)
def _track_device(request):
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    yield


def D(t):
    return t.to(_DEV[0])


def nhwc(t):
    return D(t.detach().permute(0, 2, 3, 1).contiguous())


def nchw(t):
    return t.detach().permute(0, 3, 1, 2).contiguous().cpu()


def close(a, b, rt=1e-3, at=1e-4, what='', rel_bound=1e-2):
    """Elementwise tolerance, OR (for gradients) a relative-L2 bound of 1%: a ReLU /
    ReLU6 / PReLU kink whose pre-activation is ~1e-7 can flip between two fp32
    summation orders, which moves a handful of gradient elements by O(1) while a
    genuine chain-rule error moves the whole tensor (rel-L2 ~ 1)."""
    a, b = a.detach().cpu().float(), b.cpu().float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if torch.allclose(a, b, rtol=rt, atol=at):
        return
    rel = ((a - b).norm() / (b.norm() + 1e-12)).item()
    assert rel < rel_bound, f'{what}: rel-L2 {rel:.3e}, max abs diff {(a - b).abs().max().item():.3e}'


def train_names(prefix):
    return [n for n in golden_names(prefix) if n.endswith('_train')]


def _check_block(fx, m, fn, n_in=1):
    """fn(module, *nhwc inputs) -> nhwc output; compares y, dx, param grads, running stats."""
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0]).train()
    ins = [nhwc(v).requires_grad_(True) for v in fx.ins.values()]
    y = fn(m, *ins)
    close(nchw(y), fx.outs['y'], 5e-4, 5e-5, fx.name + ' y')
    (y * nhwc(fx.outs['cot'])).sum().backward()
    for t, (k, g) in zip(ins, fx.gin.items()):
        close(nchw(t.grad), g, 2e-3, 2e-4, f'{fx.name} d/d{k}')
    params = dict(m.named_parameters())
    for k, g in fx.gp.items():
        got = params[k].grad
        assert got is not None, k
        scale = max(1.0, float(g.abs().max()))
        # MFAF's global_att branch normalises the 1x1 global pool over the BATCH: two values per channel
        # in these fixtures, so x_hat = +-1 and the gradient divides by |a - b|.  The f32 summation order
        # of the pool (atomics) moves it by 2e-3..1.3e-2 rel-L2 from run to run on the GPU
        # (tools/diag_mfaf_golden.py; every other gradient of the block sits at 1e-6): bound 5e-2 there,
        # a wrong chain rule is ~1.
        rb = 5e-2 if k.startswith('global_att.') else 1e-2
        close(got, g, 3e-3, 1e-3 * scale, f'{fx.name} grad {k}', rel_bound=rb)   # atol: grads that are analytically 0 (bias before BN) are fp32 noise ~1e-4
    sd = m.state_dict()
    for k, v in fx.outs.items():
        if k.startswith('post/'):
            close(sd[k[5:]], v, 1e-3, 1e-4, f'{fx.name} {k}')


@pytest.mark.parametrize('name', train_names('g1_') + train_names('g2_') + train_names('g3_') + train_names('g4'))
def test_sesp_train_golden(be, name):
    from led_net_amd.blocks import SESP
    from led_net_amd import train as TR
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    m = SESP(kw['nIn'], kw['nOut'], kw['stride'], 4, kw.get('r_lim', 7), kw['Spatial'])
    _check_block(fx, m, TR.sesp)


@pytest.mark.parametrize('name', [train_names('g1_')[0], train_names('g4')[0], train_names('g11_')[0], train_names('g11_')[-1]])
def test_bn_folded_into_conv_golden(be, name, monkeypatch):
    """train.FUSE_BN_INTO_CONV = 2: BatchNorm + (P)ReLU folded into the consuming convolution's input
    staging in BasicBlock (conv1 -> conv2) and SESP (cat -> expansion), forward and backward, against
    the same golden fixtures as the unfused path (the default folds the LEDHead modules only)."""
    from led_net_amd.blocks import SESP, BasicBlock
    from led_net_amd import train as TR
    monkeypatch.setattr(TR, 'FUSE_BN_INTO_CONV', 2)
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    if name.startswith('g11_'):
        m = BasicBlock(kw['in_channels'], kw['channels'], kw['stride'], kw.get('downsample', False),
                       kw.get('act_out', True))
        _check_block(fx, m, TR.basic_block)
    else:
        m = SESP(kw['nIn'], kw['nOut'], kw['stride'], 4, kw.get('r_lim', 7), kw['Spatial'])
        _check_block(fx, m, TR.sesp)


@pytest.mark.parametrize('name', train_names('g5_'))
def test_getb_train_golden(be, name):
    from led_net_amd.blocks import GETB
    from led_net_amd import train as TR
    fx = Fixture(name)
    _check_block(fx, GETB(128, 8, 8), TR.getb)


@pytest.mark.parametrize('name', train_names('g6_'))
def test_mfaf_train_golden(be, name):
    from led_net_amd.blocks import MFAF
    from led_net_amd import train as TR
    fx = Fixture(name)
    _check_block(fx, MFAF(64, 4), TR.mfaf)


@pytest.mark.parametrize('name', train_names('g16_'))
def test_bottleneck_train_golden(be, name):
    from led_net_amd.blocks import Bottleneck
    from led_net_amd import train as TR
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    m = Bottleneck(kw['in_channels'], kw['channels'], kw['stride'], kw.get('downsample', False), kw.get('act_out', False))
    _check_block(fx, m, TR.bottleneck)


@pytest.mark.parametrize('name', train_names('g11_'))
def test_basic_block_train_golden(be, name):
    from led_net_amd.blocks import BasicBlock
    from led_net_amd import train as TR
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    m = BasicBlock(kw['in_channels'], kw['channels'], kw['stride'], kw.get('downsample', False),
                   kw.get('act_out', True))
    _check_block(fx, m, TR.basic_block)


@pytest.mark.parametrize('name', train_names('g14_'))
def test_ppm_train_golden(be, name):
    """DAPPM / PAPPM context tail in training mode: output, input gradient, every parameter gradient, running stats"""
    from led_net_amd.blocks import PPM
    from led_net_amd import train as TR
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    m = PPM(kw['in_channels'], kw['branch_channels'], kw['out_channels'], fx.meta['kind'].lower(), kw['num_scales'])
    _check_block(fx, m, TR.ppm)


def test_led_head_train_golden(be):
    """LEDHead.forward (train) + loss_by_feat: logits, losses, accuracy, grads wrt the
    four backbone features and every head parameter (fixture g10, mmcv shim)."""
    import led_net_amd as L
    fx = Fixture('g10_ledhead_train')
    kw = fx.meta['kwargs']
    m = L.LEDHead(**kw)
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0]).train()
    # backbone features arrive as NCHW *views* of NHWC storage (what LEDNet returns)
    ins = {k: nhwc(fx.ins[k]).requires_grad_(True) for k in ('c3', 'c5', 'x1', 'x2')}
    label = D(fx.ins['label'])
    samples = [L.SegDataSample(gt=label[i]) for i in range(label.shape[0])]
    out = m.loss(tuple(ins[k].permute(0, 3, 1, 2) for k in ('c3', 'c5', 'x1', 'x2')), samples)
    for k in ('loss_context', 'loss_spatial', 'acc_seg'):
        close(out[k].reshape(-1), fx.outs[k].reshape(-1), 1e-4, 1e-6, k)
    (out['loss_context'] + out['loss_spatial']).backward()
    for k, g in fx.gin.items():
        close(nchw(ins[k].grad), g, 2e-3, 1e-7, f'd/d{k}')
    params = dict(m.named_parameters())
    for k, g in fx.gp.items():
        close(params[k].grad, g, 3e-3, 1e-6, f'grad {k}')


def _randomize(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and n.endswith('weight') and 'act' not in n:
                p.copy_(0.7 + 0.6 * torch.rand(p.shape, generator=g))
            elif p.dim() == 1 and n.endswith('bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif 'relative_position_bias_table' in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))


def test_whole_train_step_vs_oracle(be):
    """mode='loss' + backward + one SGD step on 2 x 3 x 320 x 320 vs oracle.spec.loss
    autograd + torch.optim.SGD: losses, accuracy, updated weights, running stats."""
    import led_net_amd as L
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(os.path.dirname(__file__), 'data', 'lednet_test_config.py'))
    cfg['model']['decode_head']['loss_decode'][0]['min_kept'] = 20000
    cfg['model']['decode_head']['loss_decode'][1]['min_kept'] = 20000
    model = L.MODELS.build(cfg['model'])
    _randomize(model, 3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(_DEV[0])
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (2, 3, 320, 320), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (2, 1, 320, 320), dtype=torch.int64, generator=g)
    lab[:, :, :6, :] = 255
    lab[:, :, :, -5:] = 255

    # ---- oracle
    init = {k: v.clone() for k, v in sd.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and 'running_' not in k}
    want = spec.loss(spec.preprocess(img), lab, sd, loss_cfg=((0.9, 20000, 1.0), (0.9, 20000, 0.4)))
    (want['decode.loss_context'] + want['decode.loss_spatial']).backward()
    used = [k for k, v in leaves.items() if v.grad is not None]
    opt = torch.optim.SGD([leaves[k] for k in used], lr=0.01, momentum=0.9, weight_decay=5e-4)
    opt.step()

    # ---- product
    tr = L.Trainer(model, cfg, max_iters=80000)
    samples = [L.SegDataSample(gt=D(lab[i])) for i in range(2)]
    got = tr.train_step(D(img), samples)
    for k in ('decode.loss_context', 'decode.loss_spatial', 'decode.acc_seg'):
        close(got[k].reshape(-1), want[k].detach().reshape(-1), 2e-3, 1e-4, k)
    new = model.state_dict()
    # The update (lr * (grad + wd*p)) of every parameter vs the oracle's.  The whole-net
    # gradient is a discontinuous function of the input (ReLU/PReLU kinks in ~100 layers,
    # the SEAM percentile binarisation, the OHEM selection): the ORACLE's own gradients
    # move by 0.8 % (median over parameters; 2.4 % max) under 1e-6 relative input noise
    # (tools/diag_train_parity.py), so per-parameter agreement is statistical here; the
    # tight checks are the block-level golden tests above.  A wrong chain rule shows up as
    # rel ~ 1 on the affected parameters.
    rels = []
    for k in used:
        a, b = new[k].detach().cpu(), leaves[k].detach()
        upd = (b - init[k]).norm().item()
        err = (a - b).norm().item()
        if upd > 1e-4:      # skip analytically-zero gradients (biases in front of a BatchNorm)
            rels.append((err / upd, k))
    rels.sort()
    med, p90, worst = rels[len(rels) // 2][0], rels[int(len(rels) * 0.9)][0], rels[-1]
    # (one flipped SEAM edge pixel / OHEM selection shifts every parameter's gradient by a
    #  few % at once, hence a bound on the whole distribution rather than per parameter)
    assert med < 0.15 and p90 < 0.3 and worst[0] < 0.6, (med, p90, worst)
    # parameters without gradient: SEAM conv_1 (non-differentiable edge map) and the unused
    # module_act of stride-2 context SESP blocks (eesp.py:110-111 returns before it);
    # the product must skip exactly the same set (torch.optim.SGD skips grad=None)
    dead = sorted(k for k in leaves if k not in used)
    assert all('seam.conv_1' in k or k.endswith('.1.module_act.weight') for k in dead), dead
    live_names = {n for n, p in model.named_parameters() if any(p is q for q in tr.live)}
    assert live_names == set(used)
    for k in dead:
        close(new[k], init[k], 0, 0, k)
    for k, v in sd.items():
        if 'running_' in k:
            close(new[k], v, 2e-3, 2e-4, k)


def test_direct_gradient_sinks_match_autograd_accumulation(be):
    """From the second step on conv / BatchNorm / PReLU backward kernels reduce straight into the
    Trainer's flat gradient buffer (train._Sinks) and small zeroed scratch comes from one arena
    fill.  At ONE model state, forward+backward through the sinks must give the gradients that the
    autograd-accumulated path gives; the run-to-run noise floor of the latter (f32 atomics order ->
    ReLU kinks, SEAM percentile) is measured with a second reference pass and bounds the check."""
    import led_net_amd as L
    from conftest import slow_on_emu
    slow_on_emu(_DEV[0])
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(os.path.dirname(__file__), 'data', 'lednet_test_config.py'))
    cfg['model']['decode_head']['loss_decode'][0]['min_kept'] = 5000
    cfg['model']['decode_head']['loss_decode'][1]['min_kept'] = 5000
    model = L.MODELS.build(cfg['model'])
    _randomize(model, 5)
    model.to(_DEV[0])
    g = torch.Generator().manual_seed(12)
    # batch 8 on the GPU: MFAF's global branch normalises one value per image over the batch, and with two
    # images that BatchNorm is singular enough (x_hat = +-1) for f32 summation-order noise to swing the whole
    # upstream gradient by 20-30 % between two passes of the same code (tools/diag_sinks.py); the emulator is
    # deterministic and keeps the cheap batch of 2
    nb = 2 if _DEV[0].type == 'cpu' else 8
    img = torch.randint(0, 256, (nb, 3, 320, 320), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (nb, 1, 320, 320), dtype=torch.int64, generator=g)
    lab[:, :, :4, :] = 255
    samples = [L.SegDataSample(gt=D(lab[i])) for i in range(nb)]
    tr = L.Trainer(model, cfg, max_iters=100)
    tr.train_step(D(img), samples)                        # attaches the flat gradient views / sinks
    assert tr._sink_map and float(tr.flat_grad.abs().max()) == 0.0   # SGD re-zeroed the buffer
    state = {k: v.clone() for k, v in model.state_dict().items()}
    sinks = tr._sink_map

    def grads(use_sinks):
        model.load_state_dict(state)                      # running statistics move in every forward
        tr.flat_grad.zero_()
        tr._sink_map = sinks if use_sinks else {}
        out = tr.forward_backward(D(img), samples)
        return tr.flat_grad.detach().cpu().clone(), {k: float(v.reshape(-1)[0]) for k, v in out.items()}
    g_ref, o_ref = grads(False)
    g_ref2, o_ref2 = grads(False)
    g_snk, o_snk = grads(True)
    tr._sink_map = sinks
    for k in o_ref:
        # the forward is the same code in all three passes: what differs is GPU summation order (and, for the
        # accuracy, which near-tie pixels flip), so only a coarse bound applies here
        tol = max(2e-3 * abs(o_ref[k]) + 1e-4, 10 * abs(o_ref2[k] - o_ref[k]))
        assert abs(o_snk[k] - o_ref[k]) <= tol, (k, o_snk[k], o_ref[k], o_ref2[k])
    rel, floor, off = [], [], 0
    gmax = 0.0
    spans = []
    for p_ in tr.params:
        n = p_.numel()
        if any(p_ is q for q in tr.live):
            spans.append((off, n))
            gmax = max(gmax, g_ref[off:off + n].norm().item())
        off += n
    for off, n in spans:
        a, b, c = g_snk[off:off + n], g_ref[off:off + n], g_ref2[off:off + n]
        # analytically-zero gradients (a bias in front of a BatchNorm) are pure rounding noise: measure
        # them against the largest gradient instead of against themselves
        den = max(b.norm().item(), 1e-4 * gmax) + 1e-12
        rel.append((a - b).norm().item() / den)
        floor.append((c - b).norm().item() / den)
    rel_s, floor_s = sorted(rel), sorted(floor)
    med, p90, worst = rel_s[len(rel_s) // 2], rel_s[int(len(rel_s) * 0.9)], rel_s[-1]
    fmed, fp90, fworst = floor_s[len(floor_s) // 2], floor_s[int(len(floor_s) * 0.9)], floor_s[-1]
    print(f'sinks vs autograd: median rel-L2 {med:.2e} p90 {p90:.2e} worst {worst:.2e}; '
          f'noise floor {fmed:.2e} / {fp90:.2e} / {fworst:.2e}')
    # The step is chaotic on the GPU (f32 atomics order -> SEAM percentile pixels / ReLU kinks flip between
    # two passes of the SAME code), so the bound is on the distribution over parameters: a missed or doubled
    # accumulation is rel ~ 1 on a whole class of parameters (every conv weight, every BN gamma, ...) and
    # moves the median / 90th percentile; single small-gradient parameters reach ~0.8 by noise alone.
    assert med <= max(1e-3, 10 * fmed) and p90 <= max(5e-2, 10 * fp90), (med, p90, worst, fmed, fp90, fworst)
    n_bad, f_bad = sum(r > 0.5 for r in rel), sum(f > 0.5 for f in floor)
    assert n_bad <= max(4, 3 * f_bad + 2), (n_bad, f_bad)     # a small class (the ~30 PReLU slopes) would show here
