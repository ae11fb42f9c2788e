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
