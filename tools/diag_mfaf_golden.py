#!/usr/bin/env python3
"""rel-L2 of every MFAF parameter gradient vs the golden fixture, three passes (GPU run-to-run spread)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import conftest
from conftest import Fixture
import led_net_amd
from led_net_amd.blocks import MFAF
from led_net_amd import train as TR
dev = torch.device('cuda:0')
nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
for name in ('g6_mfaf_64_19x21_train', 'g6_mfaf_64_24x40_train'):
    fx = Fixture(name)
    for rep in range(3):
        m = MFAF(64, 4); m.load_state_dict(fx.sd, strict=True); m.to(dev).train()
        ins = [nhwc(v).requires_grad_(True) for v in fx.ins.values()]
        y = TR.mfaf(m, *ins)
        (y * nhwc(fx.outs['cot'])).sum().backward()
        rels = {k: float(((p.grad.cpu() - fx.gp[k]).norm() / (fx.gp[k].norm() + 1e-12))) for k, p in m.named_parameters() if k in fx.gp and k.endswith('weight')}
        worst = sorted(rels.items(), key=lambda kv: -kv[1])[:4]
        print(name, rep, ' '.join(f'{k}={v:.2e}' for k, v in worst), flush=True)
