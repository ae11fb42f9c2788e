#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference's surviving
LED-Net building blocks from /root/reference (this container only).

Only data leaves this script: seeded inputs, the module state_dict, outputs and
autograd gradients, written as .npz under tests/golden/.  No reference source
is copied.  Re-run with:  python tests/golden/gen_golden.py

How the reference is imported:
  * parent packages (mmseg, mmseg.models, ...) are registered as EMPTY modules
    whose __path__ points at the reference directories, so none of the
    reference's __init__.py files execute (mmseg/models/__init__.py cannot: it
    star-imports the withheld lednet.py placeholder);
  * `timm.models.layers` (absent here) is replaced by a 3-name stand-in:
    DropPath is never instantiated at drop_path=0, to_2tuple is unused on the
    GETB path, trunc_normal_ only initialises the bias table (we overwrite
    every parameter with seeded values afterwards anyway);
  * for the *_shim fixtures (LEDHead, BaseDecodeHead.predict_by_feat,
    BasicBlock) `mmcv.cnn.ConvModule/build_norm_layer/build_activation_layer`
    and `mmengine.model.BaseModule` (third-party, un-vendored, mmcv>=2.0.0rc4
    <2.2.0 / mmengine>=0.5.0,<1.0.0 per requirements/mminstall.txt) are
    restated below following their published semantics.  Those fixtures carry
    meta['shim']=True: parity for them is pinned only modulo that restatement.

Fixture layout (np.savez): keys  sd/<state_dict key>, in/<name>, out/<name>,
gin/<name> (grad wrt input), gp/<param key> (grad wrt parameter), meta (json).
"""
import json
import math
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- #
# import plumbing
# --------------------------------------------------------------------------- #
def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def install_stubs():
    _pkg('mmseg', f'{REF}/mmseg')
    _pkg('mmseg.models', f'{REF}/mmseg/models')
    for sub in ('backbones', 'classification', 'nn_layers', 'losses',
                'decode_heads'):
        _pkg(f'mmseg.models.{sub}', f'{REF}/mmseg/models/{sub}')
    # mmseg.models.utils: only wrappers.resize / basic_block are wanted; its
    # real __init__ pulls in mmcv-heavy siblings.
    u = _pkg('mmseg.models.utils', f'{REF}/mmseg/models/utils')

    # ---- timm stand-in (see module docstring)
    timm = _pkg('timm')
    tm = _pkg('timm.models')
    tl = _pkg('timm.models.layers')
    tl.DropPath = nn.Identity
    tl.to_2tuple = lambda x: (x, x)
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models = tm
    tm.layers = tl

    # ---- mmseg.registry.MODELS mini registry (build() is used by
    # BaseDecodeHead.__init__ for the loss list and BasicBlock for act_cfg_out)
    class _Registry:
        def __init__(self):
            self.d = {'ReLU': nn.ReLU}

        def register_module(self, name=None, module=None, force=False):
            def deco(cls):
                self.d[cls.__name__] = cls
                return cls
            return deco

        def build(self, cfg):
            cfg = dict(cfg)
            return self.d[cfg.pop('type')](**cfg)

    reg = _pkg('mmseg.registry')
    reg.MODELS = _Registry()
    ut = _pkg('mmseg.utils')
    ut.ConfigType = ut.OptConfigType = ut.SampleList = object
    st = _pkg('mmseg.structures')
    st.build_pixel_sampler = lambda cfg, **kw: None

    # ---- mmengine / mmcv restatement (shim fixtures only)
    me = _pkg('mmengine')
    mm = _pkg('mmengine.model')

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg
    mm.BaseModule = BaseModule
    mm.Sequential = nn.Sequential
    me.model = mm

    def build_norm_layer(cfg, num_features, postfix=''):
        t = cfg['type']
        assert t in ('BN', 'SyncBN')
        bn = nn.BatchNorm2d(num_features, eps=cfg.get('eps', 1e-5))
        return 'bn' + str(postfix), bn

    def build_activation_layer(cfg):
        cfg = dict(cfg)
        t = cfg.pop('type')
        return {'ReLU': nn.ReLU, 'ReLU6': nn.ReLU6}[t](**cfg)

    class ConvModule(nn.Module):
        """conv(bias='auto' -> not with_norm) / norm / act applied in `order`;
        with order=('norm','act','conv') the norm is over in_channels."""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                     padding=0, dilation=1, groups=1, bias='auto',
                     conv_cfg=None, norm_cfg=None,
                     act_cfg=dict(type='ReLU'), inplace=True,
                     order=('conv', 'norm', 'act')):
            super().__init__()
            self.order = order
            self.with_norm = norm_cfg is not None
            self.with_activation = act_cfg is not None
            if bias == 'auto':
                bias = not self.with_norm
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size,
                                  stride, padding, dilation, groups, bias)
            if self.with_norm:
                nf = out_channels if order.index('norm') > order.index(
                    'conv') else in_channels
                _, self.bn = build_norm_layer(norm_cfg, nf)
            if self.with_activation:
                a = dict(act_cfg)
                a.setdefault('inplace', inplace)
                self.activate = build_activation_layer(a)

        def forward(self, x):
            for layer in self.order:
                if layer == 'conv':
                    x = self.conv(x)
                elif layer == 'norm' and self.with_norm:
                    x = self.bn(x)
                elif layer == 'act' and self.with_activation:
                    x = self.activate(x)
            return x

    mc = _pkg('mmcv')
    mcc = _pkg('mmcv.cnn')
    mcc.ConvModule = ConvModule
    mcc.build_norm_layer = build_norm_layer
    mcc.build_activation_layer = build_activation_layer
    mc.cnn = mcc


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
def seeded_init(mod, seed):
    """Overwrite every parameter/buffer with seeded, well-conditioned values
    (non-trivial BN statistics, PReLU slopes, biases) so nothing is hidden by
    default inits (gamma=1, beta=0, running_mean=0 ...)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(fan_in))
            elif name.endswith('bn.weight') or 'norm' in name and name.endswith('weight') \
                    or (name.split('.')[-1] == 'weight' and p.dim() == 1 and 'act' not in name):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif name.endswith('act.weight') or name.endswith('module_act.weight'):
                p.copy_(0.05 + 0.4 * torch.rand(p.shape, generator=g))
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for name, b in mod.named_buffers():
            if name.endswith('running_mean'):
                b.copy_(0.3 * torch.randn(b.shape, generator=g))
            elif name.endswith('running_var'):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))


def save(name, mod, inputs, outputs, gin=None, gp=None, meta=None):
    d = {}
    if mod is not None:
        for k, v in mod.state_dict().items():
            d['sd/' + k] = v.detach().numpy()
    for k, v in inputs.items():
        d['in/' + k] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
    for k, v in outputs.items():
        d['out/' + k] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
    for k, v in (gin or {}).items():
        d['gin/' + k] = v.detach().numpy()
    for k, v in (gp or {}).items():
        d['gp/' + k] = v.detach().numpy()
    d['meta'] = np.asarray(json.dumps(meta or {}))
    path = os.path.join(OUT, name + '.npz')
    np.savez(path, **d)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def run_module_case(name, mod, xs, seed, meta, train_too=True, call=None):
    """eval forward; then (optionally) train-mode forward + backward of
    sum(out * cotangent) with a seeded cotangent."""
    call = call or (lambda m, *a: m(*a))
    seeded_init(mod, seed)
    sd0 = {k: v.clone() for k, v in mod.state_dict().items()}
    mod.eval()
    with torch.no_grad():
        y = call(mod, *[x.clone() for x in xs.values()])
    save(name + '_eval', mod, xs, {'y': y}, meta=dict(meta, mode='eval'))
    if not train_too:
        return
    mod.load_state_dict(sd0)
    mod.train()
    xin = {k: v.clone().requires_grad_(True) for k, v in xs.items()}
    y = call(mod, *xin.values())
    g = torch.Generator().manual_seed(seed + 1)
    cot = torch.randn(y.shape, generator=g)
    (y * cot).sum().backward()
    gp = {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}
    gin = {k: v.grad for k, v in xin.items()}
    # state dict AFTER the step holds the updated running stats: store the
    # pre-step one under sd/ and the post-step BN buffers under out/.
    post = {('post/' + k): v for k, v in mod.state_dict().items()
            if 'running_' in k}
    mod2_sd = sd0
    d_out = {'y': y, 'cot': cot}
    d_out.update(post)
    # save with pre-step state
    holder = SimpleNamespace(state_dict=lambda: mod2_sd)
    save(name + '_train', holder, xs, d_out, gin=gin, gp=gp,
         meta=dict(meta, mode='train'))


# --------------------------------------------------------------------------- #
def main():
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(4)
    from mmseg.models.nn_layers.eesp import SESP                      # eesp.py:15
    from mmseg.models.backbones.UNetFormer_GETB import GETBBlock     # UNetFormer_GETB.py:209
    from mmseg.models.classification.model_utils import Muti_AFF     # model_utils.py:356
    from mmseg.models.losses.ohem_cross_entropy_loss import OhemCrossEntropy
    from mmseg.models.losses.accuracy import accuracy
    from mmseg.models.utils.wrappers import resize
    sys.modules['mmseg.models.losses'].accuracy = accuracy
    sys.modules['mmseg.models.utils'].resize = resize

    g = torch.Generator().manual_seed(304)

    # ---- G1-G4 SESP variants (eesp.py:15-118)
    sesp_cases = [
        ('g1_sesp_64_64_spatial', dict(nIn=64, nOut=64, stride=1, Spatial=True), (2, 64, 16, 24)),
        ('g2_sesp_64_64_ctx_r9', dict(nIn=64, nOut=64, stride=1, Spatial=False, r_lim=9), (2, 64, 16, 24)),
        ('g3_sesp_64_128_spatial', dict(nIn=64, nOut=128, stride=1, Spatial=True), (2, 64, 16, 24)),
        ('g4_sesp_128_128_s2_ctx_r9', dict(nIn=128, nOut=128, stride=2, Spatial=False, r_lim=9), (2, 128, 16, 24)),
        ('g4b_sesp_64_128_ctx_r9', dict(nIn=64, nOut=128, stride=1, Spatial=False, r_lim=9), (2, 64, 15, 21)),
    ]
    for i, (name, kw, shp) in enumerate(sesp_cases):
        x = torch.randn(shp, generator=g)
        run_module_case(name, SESP(**kw), {'x': x}, 100 + i, dict(kind='SESP', kwargs=kw))

    # ---- G5 GETBBlock (UNetFormer_GETB.py:97-226), incl. odd (reflect-pad) size
    for i, (name, shp) in enumerate([('g5_getb_128_16x24', (1, 128, 16, 24)),
                                     ('g5_getb_128_13x27', (1, 128, 13, 27)),
                                     ('g5_getb_128_b2_8x8', (2, 128, 8, 8))]):
        x = torch.randn(shp, generator=g)
        kw = dict(dim=128, num_heads=8, window_size=8)
        run_module_case(name, GETBBlock(**kw), {'x': x}, 200 + i, dict(kind='GETB', kwargs=kw))

    # ---- G6 Muti_AFF (classification/model_utils.py:356-429)
    for i, (name, shp) in enumerate([('g6_mfaf_64_24x40', (2, 64, 24, 40)),
                                     ('g6_mfaf_64_19x21', (2, 64, 19, 21))]):
        x = torch.randn(shp, generator=g)
        r = torch.randn(shp, generator=g)
        kw = dict(channels=64, r=4)
        run_module_case(name, Muti_AFF(**kw), {'x': x, 'residual': r}, 300 + i,
                        dict(kind='Muti_AFF', kwargs=kw))

    # ---- G7 OhemCrossEntropy (ohem_cross_entropy_loss.py:52-90) + G8 accuracy
    ohem_cases = [
        ('g7_ohem_k1000', dict(thres=0.9, min_kept=1000, loss_weight=1.0), (2, 2, 64, 64), 0.1, 3.0),
        ('g7_ohem_k131072', dict(thres=0.9, min_kept=131072, loss_weight=0.4), (2, 2, 64, 64), 0.1, 3.0),
        ('g7_ohem_k100_confident', dict(thres=0.7, min_kept=100, loss_weight=1.0), (1, 2, 48, 40), 0.0, 8.0),
        ('g7_ohem_c5', dict(thres=0.9, min_kept=500, loss_weight=1.0), (2, 5, 32, 32), 0.2, 2.0),
        ('g7_ohem_all_ignored', dict(thres=0.9, min_kept=1000, loss_weight=1.0), (1, 2, 16, 16), 1.0, 1.0),
    ]
    for name, kw, shp, p_ign, scale in ohem_cases:
        n, c, h, w = shp
        score = (scale * torch.randn(shp, generator=g)).requires_grad_(True)
        tgt = torch.randint(0, c, (n, h, w), generator=g)
        ign = torch.rand((n, h, w), generator=g) < p_ign
        tgt[ign] = 255
        crit = OhemCrossEntropy(**kw)
        loss = crit(score, tgt)
        gin = {}
        if loss.requires_grad:
            loss.backward()
            gin = {'score': score.grad}
        acc = accuracy(score.detach(), tgt, ignore_index=255)
        save(name, None, {'score': score.detach(), 'target': tgt},
             {'loss': loss.detach(), 'acc': acc}, gin=gin,
             meta=dict(kind='OhemCrossEntropy', kwargs=kw))

    # ---- G9/G10/G11: shim-dependent fixtures
    from mmseg.models.decode_heads.decode_head import BaseDecodeHead   # decode_head.py:341-379
    sys.modules['mmseg.models.decode_heads'].decode_head = sys.modules[
        'mmseg.models.decode_heads.decode_head']
    from mmseg.models.decode_heads.led_head import LEDHead             # led_head.py:15
    from mmseg.models.utils.basic_block import BasicBlock              # basic_block.py:13

    # G9 predict_by_feat pyramid, even and odd (ceil path) sizes
    fake = SimpleNamespace(align_corners=False)
    for name, hw1 in [('g9_predict_by_feat_even', (36, 68)), ('g9_predict_by_feat_odd', (35, 67))]:
        h1, w1 = hw1
        size = (2 * h1, 2 * w1)
        h2, w2 = math.ceil(size[0] / 4), math.ceil(size[1] / 4)
        xc = torch.randn((2, 2, math.ceil(h2 / 2), math.ceil(w2 / 2)), generator=g)
        hx1 = torch.relu(torch.randn((2, 2, h1, w1), generator=g))
        hx2 = torch.relu(torch.randn((2, 2, h2, w2), generator=g))
        y = BaseDecodeHead.predict_by_feat(fake, (xc, hx1, hx2), [dict(img_shape=size)])
        save(name, None, {'x_c': xc, 'head_x1': hx1, 'head_x2': hx2}, {'y': y},
             meta=dict(kind='predict_by_feat', shim=True))

    # G10 LEDHead forward (eval+train), loss_by_feat with grads
    head_kw = dict(in_channels=128, channels=64, num_classes=2,
                   norm_cfg=dict(type='BN', requires_grad=True),
                   dropout_ratio=0., align_corners=False,
                   loss_decode=[dict(type='OhemCrossEntropy', thres=0.9, min_kept=1000, loss_weight=1.0),
                                dict(type='OhemCrossEntropy', thres=0.9, min_kept=1000, loss_weight=0.4)])
    head = LEDHead(**head_kw)
    seeded_init(head, 400)
    sd0 = {k: v.clone() for k, v in head.state_dict().items()}
    H, W = 64, 96
    c3 = torch.randn((2, 64, H // 8, W // 8), generator=g)
    c5 = torch.randn((2, 128, H // 8, W // 8), generator=g)
    x1 = torch.randn((2, 32, H // 2, W // 2), generator=g)
    x2 = torch.randn((2, 32, H // 4, W // 4), generator=g)
    head.eval()
    with torch.no_grad():
        xc, h1, h2 = head((c5, x1, x2))
        fused = head.predict_by_feat((xc, h1, h2), [dict(img_shape=(H, W))])
    save('g10_ledhead_eval', head, {'c5': c5, 'x1': x1, 'x2': x2},
         {'x_c': xc, 'head_x1': h1, 'head_x2': h2, 'fused': fused},
         meta=dict(kind='LEDHead', kwargs=head_kw, shim=True, mode='eval'))
    head.load_state_dict(sd0)
    head.train()
    ins = {k: v.clone().requires_grad_(True) for k, v in
           dict(c3=c3, c5=c5, x1=x1, x2=x2).items()}
    logits = head((ins['c3'], ins['c5'], ins['x1'], ins['x2']))
    label = torch.randint(0, 2, (2, 1, H, W), generator=g)
    label[:, :, :4, :] = 255
    label[:, :, :, -4:] = 255
    samples = [SimpleNamespace(gt_sem_seg=SimpleNamespace(data=label[i])) for i in range(2)]
    losses = head.loss_by_feat(logits, samples)
    total = losses['loss_context'] + losses['loss_spatial']
    total.backward()
    outs = {'x_c': logits[0], 'x_s': logits[1], 'head_x1': logits[2], 'head_x2': logits[3],
            'loss_context': losses['loss_context'], 'loss_spatial': losses['loss_spatial'],
            'acc_seg': losses['acc_seg']}
    holder = SimpleNamespace(state_dict=lambda: sd0)
    save('g10_ledhead_train', holder, dict(c3=c3, c5=c5, x1=x1, x2=x2, label=label), outs,
         gin={k: v.grad for k, v in ins.items()},
         gp={k: p.grad for k, p in head.named_parameters() if p.grad is not None},
         meta=dict(kind='LEDHead', kwargs=head_kw, shim=True, mode='train'))

    # G11 BasicBlock (basic_block.py:13-75): plain and strided+downsample
    nc = dict(type='BN')
    bb = BasicBlock(32, 32, norm_cfg=nc)
    x = torch.randn((2, 32, 12, 20), generator=g)
    run_module_case('g11_basicblock_32', bb, {'x': x}, 500,
                    dict(kind='BasicBlock', kwargs=dict(in_channels=32, channels=32, stride=1), shim=True))
    ds = nn.Sequential(nn.Conv2d(32, 64, 1, 2, bias=False), nn.BatchNorm2d(64))
    bb2 = BasicBlock(32, 64, stride=2, downsample=ds, norm_cfg=nc, act_cfg_out=None)
    x = torch.randn((2, 32, 13, 21), generator=g)
    run_module_case('g11_basicblock_32_64_s2', bb2, {'x': x}, 501,
                    dict(kind='BasicBlock', kwargs=dict(in_channels=32, channels=64, stride=2,
                                                        downsample=True, act_out=False), shim=True))

    # G12 SEAM prototype op chain (tools/speed/ddrnet_speed.py:24-37,88-99,282-338)
    # re-run op by op with torch primitives on a seeded 1-channel map: the
    # prototype file itself cannot be imported (broken `Block` import, .cuda()).
    lap = torch.tensor([-1, -1, -1, -1, 8, -1, -1, -1, -1.]).reshape(1, 1, 3, 3)
    fk = torch.tensor([[6. / 10], [3. / 10], [1. / 10]]).reshape(1, 3, 1, 1)
    thr = 0.1
    for name, shp in [('g12_seam_chain_24x36', (1, 1, 24, 36)), ('g12_seam_chain_23x37', (1, 1, 23, 37))]:
        seg = torch.randn(shp, generator=g)
        mn, mx = seg.min(), seg.max()
        segn = (seg - mn) / (mx - mn)                                # normalize_tensor :24-37
        b1 = (F.conv2d(segn, lap, padding=1).clamp(min=0) > thr).float()
        b2 = F.conv2d(segn, lap, stride=2, padding=1).clamp(min=0)
        b4 = F.conv2d(segn, lap, stride=4, padding=1).clamp(min=0)
        b2u = (F.interpolate(b2, b1.shape[2:], mode='nearest') > thr).float()
        b4u = (F.interpolate(b4, b1.shape[2:], mode='nearest') > thr).float()
        pyr = torch.stack((b1, b2u, b4u), dim=1).squeeze(2)
        edge = (F.conv2d(pyr, fk) > thr).float()
        save(name, None, {'seg': seg},
             {'norm': segn, 'b1': b1, 'b2u': b2u, 'b4u': b4u, 'edge': edge},
             meta=dict(kind='SEAM', threshold=thr))


if __name__ == '__main__':
    main()
