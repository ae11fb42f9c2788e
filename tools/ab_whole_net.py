#!/usr/bin/env python3
"""Whole-network A/B of LEDN_OPT_STREAM_FAST masks: logits, losses and per-parameter gradient cosines of one bf16 training
step between two masks (and the run-to-run noise floor of each mask with itself).  python tools/ab_whole_net.py [H W]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import led_net_amd as L  # noqa: E402
from led_net_amd import _lib  # noqa: E402
from test_blocks import _randomize  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (352, 488)
dev = torch.device('cuda:0')
lib = _lib.get_lib()
g = torch.Generator().manual_seed(H + W)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
img = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g).to(dev)
lab = torch.randint(0, 2, (B, 1, H, W), dtype=torch.int64, generator=g).to(dev)


def run(mask):
    lib.set_option(2, mask)
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = 20000
    model = L.MODELS.build(cfg['model'])
    _randomize(model, 5)
    model.set_act_dtype(torch.bfloat16)
    model.to(dev).train()
    tr = L.Trainer(model, cfg, max_iters=1000)
    tr.base_lr = 0.0
    losses = tr.train_step(img, [L.SegDataSample(gt=lab[i]) for i in range(B)])
    name_of = {id(p): k for k, p in model.named_parameters()}
    grads = {name_of[id(p)]: m.detach().float().cpu().clone() for p, m in zip(tr.params, tr.moms)}
    lib.set_option(2, -1)
    return {k: float(v.float().reshape(-1)[0]) for k, v in losses.items()}, grads


def cmp(a, b, tag):
    rows = []
    for k in a[1]:
        x, y = a[1][k].flatten(), b[1][k].flatten()
        if y.norm() < 1e-9:
            continue
        rows.append((float(x @ y / (x.norm() * y.norm() + 1e-30)), k))
    rows.sort()
    med = rows[len(rows) // 2][0]
    print(f'{tag}: losses {a[0]} | {b[0]}')
    print(f'   cosine: median {med:.3f}, p10 {rows[len(rows) // 10][0]:.3f}, worst {rows[:4]}')
    for key in ('backbone.stem.0.conv.weight', 'decode_head.head_x1.0.conv.weight', 'decode_head.head_x2.0.conv.weight', 'backbone.stem.2.0.conv1.conv.weight'):
        print('   ', key, [round(c, 3) for c, k in rows if k == key])


r91a, r91b, r11a, r11b = run(91), run(91), run(11), run(11)
cmp(r91a, r91b, '91 vs 91')
cmp(r11a, r11b, '11 vs 11')
cmp(r91a, r11a, '91 vs 11')
