"""Product blocks (led-net_amd/blocks.py etc.) executed on the CPU emulation build
and checked against the golden vectors from the reference modules (eval mode)
and against the oracle for the assembled network."""
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
    return D(t.permute(0, 2, 3, 1).contiguous())


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def close(a, b, rt=2e-4, at=2e-5):
    torch.testing.assert_close(a.cpu(), b.cpu(), rtol=rt, atol=at)


def eval_names(prefix):
    return [n for n in golden_names(prefix) if n.endswith('_eval')]


@pytest.mark.parametrize('name', eval_names('g1_') + eval_names('g2_') + eval_names('g3_') + eval_names('g4'))
def test_sesp_eval_golden(be, name):
    from led_net_amd.blocks import SESP
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    m = SESP(kw['nIn'], kw['nOut'], kw['stride'], 4, kw.get('r_lim', 7), kw['Spatial']).eval()
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0])
    close(nchw(m(nhwc(fx.ins['x']))), fx.outs['y'])


@pytest.mark.parametrize('name', eval_names('g5_'))
def test_getb_eval_golden(be, name):
    from led_net_amd.blocks import GETB
    fx = Fixture(name)
    m = GETB(128, 8, 8).eval()
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0])
    close(nchw(m(nhwc(fx.ins['x']))), fx.outs['y'], 5e-4, 5e-5)


@pytest.mark.parametrize('name', eval_names('g6_'))
def test_mfaf_eval_golden(be, name):
    from led_net_amd.blocks import MFAF
    fx = Fixture(name)
    m = MFAF(64, 4).eval()
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0])
    close(nchw(m(nhwc(fx.ins['x']), nhwc(fx.ins['residual']))), fx.outs['y'])


@pytest.mark.parametrize('name', eval_names('g11_'))
def test_basic_block_eval_golden(be, name):
    from led_net_amd.blocks import BasicBlock
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    m = BasicBlock(kw['in_channels'], kw['channels'], kw['stride'], kw.get('downsample', False),
                   kw.get('act_out', True)).eval()
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0])
    close(nchw(m(nhwc(fx.ins['x']))), fx.outs['y'])


@pytest.mark.parametrize('name', eval_names('g16_'))
def test_bottleneck_eval_golden(be, name):
    from led_net_amd.blocks import Bottleneck
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    m = Bottleneck(kw['in_channels'], kw['channels'], kw['stride'], kw.get('downsample', False),
                   kw.get('act_out', False)).eval()
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0])
    close(nchw(m(nhwc(fx.ins['x']))), fx.outs['y'])


def test_led_head_eval_golden(be):
    from led_net_amd import LEDHead
    fx = Fixture('g10_ledhead_eval')
    kw = fx.meta['kwargs']
    m = LEDHead(**kw).eval()
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0])
    ins = tuple(D(fx.ins[k]) for k in ('c5', 'x1', 'x2'))
    xc, h1, h2 = m.forward(ins)
    close(xc, fx.outs['x_c'])
    close(h1, fx.outs['head_x1'])
    close(h2, fx.outs['head_x2'])
    fused = m.predict(ins)
    close(fused, fx.outs['fused'])
    # predict_by_feat on reference-shaped (NCHW) logits
    close(m.predict_by_feat(tuple(D(fx.outs[k]) for k in ('x_c', 'head_x1', 'head_x2'))), fx.outs['fused'])


@pytest.mark.parametrize('name', golden_names('g9_'))
def test_fuse_predict_golden(be, name):
    from led_net_amd import LEDHead
    fx = Fixture(name)
    got, mask = LEDHead.fuse_predict(nhwc(fx.ins['x_c']), nhwc(fx.ins['head_x1']), nhwc(fx.ins['head_x2']),
                                     argmax=True)
    close(got, fx.outs['y'], 1e-5, 1e-6)
    want_mask = fx.outs['y'].argmax(1)
    assert (mask.long().cpu() != want_mask).float().mean().item() < 1e-4


def _randomize(model, seed):
    """non-trivial BN statistics / PReLU slopes so that folding errors show."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, b in model.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif n.endswith('running_var'):
                b.copy_(0.6 + 0.8 * torch.rand(b.shape, generator=g))
        for n, p in model.named_parameters():
            if p.dim() == 1 and ('bn' in n or 'norm' in n or n.split('.')[-2].isdigit()) and n.endswith('weight'):
                p.copy_(0.7 + 0.6 * torch.rand(p.shape, generator=g))
            elif p.dim() == 1 and n.endswith('bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif 'relative_position_bias_table' in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))


@pytest.mark.parametrize('hw', [(320, 328)])
def test_whole_network_predict_vs_oracle(be, hw):
    import led_net_amd as L
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(os.path.dirname(__file__), 'data', 'lednet_test_config.py'))
    model = L.MODELS.build(cfg['model']).eval()
    _randomize(model, 1)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(_DEV[0])
    img = torch.randint(0, 256, (1, 3, *hw), dtype=torch.uint8)
    x = spec.preprocess(img)
    with torch.no_grad():
        want_logits, want_mask = spec.predict(x, sd)
        feats_want = spec.lednet(x, sd, training=False, p='backbone.')
        feats = model.extract_feat(D(img))
        for got, want, nm in zip(feats, feats_want, ('c5', 'x1', 'x2')):
            assert got.shape == want.shape, nm
            close(got.contiguous(), want, 2e-3, 2e-4)
        out = model(D(img), mode='predict')
    logits = torch.stack([o.seg_logits.data for o in out]).cpu()
    mask = torch.cat([o.pred_sem_seg.data for o in out]).long().cpu()
    close(logits, want_logits, 2e-3, 2e-4)
    margin = (want_logits[:, 0] - want_logits[:, 1]).abs()
    bad = (mask != want_mask) & (margin > 1e-3)
    assert bad.sum().item() == 0
    # mode='tensor' returns the raw 3-tuple of logits (encoder_decoder.py:224-239)
    t = model(D(img), mode='tensor')
    assert [tuple(a.shape[1:]) for a in t] == [(2, math.ceil(hw[0] / 8), math.ceil(hw[1] / 8)),
                                               (2, hw[0] // 2, hw[1] // 2), (2, hw[0] // 4, hw[1] // 4)]


@pytest.mark.parametrize('name', eval_names('g14_'))
def test_ppm_eval_golden(be, name):
    """blocks.PPM (DAPPM / PAPPM) on the HIP kernels vs the fixtures generated from utils/ppm.py"""
    from led_net_amd.blocks import PPM
    fx = Fixture(name)
    kw = fx.meta['kwargs']
    m = PPM(kw['in_channels'], kw['branch_channels'], kw['out_channels'], fx.meta['kind'].lower(), kw['num_scales']).eval()
    m.load_state_dict(fx.sd, strict=True)
    m.to(_DEV[0])
    close(nchw(m(nhwc(fx.ins['x']))), fx.outs['y'], 5e-4, 5e-5)


VARIANTS = [
    dict(context_tail='pappm'),
    dict(context_tail='dappm'),
    dict(cespb_depth=(1, 3)),
    dict(seam_mode='fixed', seam_threshold=0.1),
    dict(getb_stage3=False),
]


@pytest.mark.parametrize('flags', VARIANTS, ids=lambda f: '-'.join(f'{k}={v}' for k, v in f.items()))
def test_reconstruction_flags_vs_oracle(be, flags):
    """every open choice of the backbone reconstruction is a constructor flag (CESPB cascade depth,
    the context tail -- GETB or the DAPPM / PAPPM pooling pyramid --, SEAM binarisation rule, GETB placement);
    each variant: inference features + fused logits and the training-mode loss against the oracle with the same
    flags."""
    import led_net_amd as L
    if flags not in (VARIANTS[0], VARIANTS[2]):       # (emulator default: the pooling tail and the cascade depth; GPU: all five)
        from conftest import slow_on_emu
        slow_on_emu(be.dev)
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(os.path.dirname(__file__), 'data', 'lednet_test_config.py'))
    cfg['model']['backbone'].update(flags)
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = 5000
    model = L.MODELS.build(cfg['model']).eval()
    _randomize(model, 2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(_DEV[0])
    okw = dict(cespb_depth=model.backbone.cespb_depth, context_tail=model.backbone.context_tail,
               getb_stage3=model.backbone.getb_stage3,
               seam_threshold=('p80' if model.backbone.seam_mode == 'percentile' else flags.get('seam_threshold', 0.1)))
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (2, 3, 320, 320), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (2, 1, 320, 320), generator=g)
    x = spec.preprocess(img)
    with torch.no_grad():
        want_logits, want_mask = spec.predict(x, sd, **okw)
        out = model(D(img), mode='predict')
    logits = torch.stack([o.seg_logits.data for o in out]).cpu()
    close(logits, want_logits, 2e-3, 5e-4)
    mask = torch.cat([o.pred_sem_seg.data for o in out]).long().cpu()
    margin = (want_logits[:, 0] - want_logits[:, 1]).abs()
    assert ((mask != want_mask) & (margin > 1e-3)).sum().item() == 0
    model.train()
    sd_t = {k: v.clone() for k, v in sd.items()}
    want = spec.loss(x, lab, sd_t, loss_cfg=((0.9, 5000, 1.0), (0.9, 5000, 0.4)), **okw)
    got = model(D(img), [L.SegDataSample(gt=D(lab)[i]) for i in range(2)], mode='loss')
    for k in want:
        # (random-init net on noise: a SEAM pixel within rounding of its hard threshold flips the whole gate at
        #  that pixel -- the fixed-threshold variant moved the loss by 2.5e-3; a mis-wired variant is >> 1e-2)
        assert abs(float(got[k].detach().reshape(-1)[0]) - float(want[k])) <= 1e-2 * abs(float(want[k])) + 1e-4, k
    # the state_dict surface of the variant is the reference's for the pooling pyramids (utils/ppm.py child names)
    if flags.get('context_tail') in ('pappm', 'dappm'):
        keys = set(sd)
        assert 'backbone.spp.scales.0.conv.weight' in keys and 'backbone.spp.scales.4.1.bn.weight' in keys
        assert 'backbone.spp.compression.conv.weight' in keys and 'backbone.spp.shortcut.bn.running_mean' in keys
        assert ('backbone.spp.processes.conv.weight' in keys) == (flags['context_tail'] == 'pappm')
        assert ('backbone.spp.processes.3.conv.weight' in keys) == (flags['context_tail'] == 'dappm')
        assert not any(k.startswith('backbone.getb2.') for k in keys)


def test_reconstruction_flag_errors():
    import led_net_amd as L
    with pytest.raises(ValueError):
        L.LEDNet(context_tail='aspp')
    with pytest.raises(ValueError):
        L.LEDNet(cespb_depth=1)          # the context branch cannot grow channels and stride in one SESP
    with pytest.raises(ValueError):
        L.LEDNet(seam_mode='otsu')
