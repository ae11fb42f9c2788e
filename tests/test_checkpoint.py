"""Checkpoint interop: mmengine-format files, DDP prefix, meta handling as in
mmseg/apis/inference.py:59-75."""
import os
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py')


def test_checkpoint_roundtrip_and_meta(tmp_path):
    import led_net_amd as L
    torch.manual_seed(1)
    cfg = L.load_config(CFG)
    a = L.MODELS.build(cfg['model'])
    path = str(tmp_path / 'iter_1.pth')
    L.save_checkpoint(a, path, meta={'dataset_meta': {'classes': ('bg', 'lane'), 'palette': [[0, 0, 0], [255, 0, 0]]}})
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {'meta', 'state_dict'}
    # the reference's key surface
    for k in ('decode_head.head.0.bn.weight', 'decode_head.head.0.conv.weight', 'decode_head.head.1.weight',
              'decode_head.conv_seg.weight', 'decode_head.aux_cls_seg.bias', 'decode_head.head_x1.0.conv.weight'):
        assert k in ck['state_dict'], k
    assert any(k.startswith('backbone.aff1.local_att.') for k in ck['state_dict'])
    assert any(k.startswith('backbone.layer5_.spp_dw.') for k in ck['state_dict'])
    m = L.init_model(CFG, path, device='cpu')
    assert not m.training and m.dataset_meta['classes'] == ('bg', 'lane')
    for (k, v), (k2, v2) in zip(a.state_dict().items(), m.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)


def test_checkpoint_ddp_prefix_bare_dict_and_old_meta(tmp_path):
    import led_net_amd as L
    torch.manual_seed(2)
    cfg = L.load_config(CFG)
    a = L.MODELS.build(cfg['model'])
    b = L.MODELS.build(cfg['model'])
    p1 = str(tmp_path / 'ddp.pth')
    torch.save({'state_dict': {'module.' + k: v for k, v in a.state_dict().items()},
                'meta': {'CLASSES': ('bg', 'lane'), 'PALETTE': [[0, 0, 0], [1, 1, 1]]}}, p1)
    ck = L.load_checkpoint(b, p1, strict=True)
    assert ck['meta']['CLASSES'] == ('bg', 'lane')
    assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
    m = L.init_model(CFG, p1, device='cpu')
    assert m.dataset_meta == {'classes': ('bg', 'lane'), 'palette': [[0, 0, 0], [1, 1, 1]]}
    p2 = str(tmp_path / 'bare.pth')
    sd = dict(a.state_dict())
    sd.pop('decode_head.conv_seg.bias')
    torch.save(sd, p2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        L.load_checkpoint(b, p2)                       # non-strict: reported, not fatal
    assert any('missing keys' in str(x.message) for x in w)
    with pytest.raises(RuntimeError):
        L.load_checkpoint(b, p2, strict=True)


def test_resume_restores_momentum_and_schedule(tmp_path):
    """tools/train.py --resume: the checkpoint carries mmengine's 'optimizer' (torch.optim.SGD.state_dict() layout over
    model.parameters() order) and 'param_schedulers'; a fresh Trainer resumed from it holds the same momentum buffers
    and PolyLR position, and torch.optim.SGD itself accepts the optimizer state."""
    import led_net_amd as L
    from conftest import bind_emu
    torch.manual_seed(5)
    cfg = L.load_config(CFG)
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = 5000
    g = torch.Generator().manual_seed(9)
    img = torch.randint(0, 256, (2, 3, 320, 320), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (2, 1, 320, 320), generator=g)
    with bind_emu():
        a = L.MODELS.build(cfg['model'])
        ta = L.Trainer(a, cfg, max_iters=100)
        samples = [L.SegDataSample(gt=lab[i]) for i in range(2)]
        ta.train_step(img, samples)
        ta.train_step(img, samples)
        path = str(tmp_path / 'iter_2.pth')
        L.save_checkpoint(a, path, meta=dict(iter=2), trainer=ta)
        ck = torch.load(path, weights_only=False)
        assert {'meta', 'state_dict', 'optimizer', 'param_schedulers'} <= set(ck)
        opt = ck['optimizer']
        n_params = len(list(a.parameters()))
        assert opt['param_groups'][0]['params'] == list(range(n_params)) and opt['param_groups'][0]['momentum'] == 0.9
        # parameters that never get a gradient (SEAM conv_1: non-differentiable edge map) have no state, as in torch
        assert 0 < len(opt['state']) < n_params
        b = L.MODELS.build(cfg['model'])
        ckb = L.load_checkpoint(b, path)
        tb = L.resume(L.Trainer(b, cfg, max_iters=100), ckb)
        assert tb.iter == 2 and abs(tb.lr() - ta.lr()) < 1e-12
        pa = {id(p): i for i, p in enumerate(ta.params)}
        for p_a, p_b, (name, _) in zip(a.parameters(), b.parameters(), a.named_parameters()):
            ma = ta.moms[pa[id(p_a)]]
            mb = tb.moms[{id(p): i for i, p in enumerate(tb.params)}[id(p_b)]]
            assert torch.equal(ma, mb), name
        assert ta.flat_mom.abs().sum() > 0
    # format check against the real thing
    sgd = torch.optim.SGD(b.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    sgd.load_state_dict(opt)
    k = next(iter(opt['state']))
    assert torch.equal(sgd.state[list(b.parameters())[k]]['momentum_buffer'], opt['state'][k]['momentum_buffer'])
