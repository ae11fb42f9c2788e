#!/usr/bin/env python3
"""Parameter / multiply-accumulate accounting of the LED-Net reconstruction against the published
1.661 M parameters and 9.206 GMAC @ 1280 x 720 (supplementary PDF p.17, Table 8; mmengine's counter, as
tools/analysis_tools/get_flops.py:38 of the reference runs it).  Test-side analysis script (it drives the CPU
oracle with a counting hook on conv2d / matmul); prints the table DESIGN.md quotes.
    python tests/count_complexity.py
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import led_net_amd as L  # noqa: E402
from oracle import spec  # noqa: E402

H, W = 720, 1280


def macs_of(flags):
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    cfg['model']['backbone'].update(flags)
    torch.manual_seed(0)
    model = L.MODELS.build(cfg['model']).eval()
    sd = {k: v for k, v in model.state_dict().items()}
    params = sum(p.numel() for p in model.parameters())
    tot = {'conv': 0, 'matmul': 0}
    conv0, mm0, bmm0 = F.conv2d, torch.matmul, torch.Tensor.__matmul__

    def conv(x, w, *a, **k):
        y = conv0(x, w, *a, **k)
        tot['conv'] += y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        return y

    def mm(a, b):
        y = mm0(a, b)
        tot['matmul'] += y.numel() * a.shape[-1]
        return y
    F.conv2d, torch.matmul, torch.Tensor.__matmul__ = conv, mm, (lambda a, b: mm(a, b))
    try:
        bb = model.backbone
        okw = dict(cespb_depth=bb.cespb_depth, context_tail=bb.context_tail, getb_stage3=bb.getb_stage3)
        with torch.no_grad():
            spec.predict(torch.zeros(1, 3, H, W), sd, **okw)
    finally:
        F.conv2d, torch.matmul, torch.Tensor.__matmul__ = conv0, mm0, bmm0
    return params, tot['conv'], tot['matmul']


if __name__ == '__main__':
    print(f'published: 1.661 M parameters, 9.206 GMAC @ {W}x{H} = {9.206e9 / (H * W):.0f} MAC / input pixel')
    for flags in (dict(), dict(cespb_depth=(2, 3)), dict(cespb_depth=(3, 3)), dict(cespb_depth=(3, 2)),
                  dict(context_tail='pappm'), dict(context_tail='dappm'), dict(getb_stage3=False)):
        p, c, m = macs_of(flags)
        print(f'{str(flags):42s} {p / 1e6:7.4f} M params ({p / 1.661e6 * 100:5.1f} %)  conv {c / 1e9:6.3f} G + attention '
              f'{m / 1e9:5.3f} G = {(c + m) / 1e9:6.3f} GMAC ({(c + m) / 9.206e9 * 100:5.1f} %)  {(c + m) / (H * W):6.0f} MAC/px')
