#!/usr/bin/env python3
"""Training-trajectory fixture: 100 SGD steps of the CPU oracle (oracle/spec.py, fp32 torch ops) on ONE fixed
4 x 3 x 320 x 320 batch with learnable labels, from seeded weights, with the LED-Net config's optimiser
(configs/LED_Net/LEDNet_80k_cityscapes-1024x1024.py:61-75: SGD lr 0.01, momentum 0.9, weight decay 5e-4, PolyLR power
0.9 over 80 000 iterations).  tests/test_trajectory.py runs the same 100 steps on the MI355X (f32 and bf16) and
compares the loss curves.  The batch and the initial weights are rebuilt from seeds by `problem()` on both sides; the
fixture stores only the oracle's per-step losses (and a checksum of the batch).
    python tests/golden/gen_trajectory.py            (about 2 minutes on 8 cores)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g18_trajectory_oracle.npz')
STEPS, LR, MOM, WD, MAX_ITERS, POWER = 100, 0.01, 0.9, 5e-4, 80000, 0.9
MIN_KEPT = 50000


def problem():
    """the fixed batch: smooth random images, label = a thresholded blur of the image (learnable), an ignored border"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(1804)
    N, H, W = 4, 320, 320
    base = torch.rand(N, 3, H // 8, W // 8, generator=g)
    img = F.interpolate(base, size=(H, W), mode='bilinear', align_corners=False)
    img = (img + 0.08 * torch.randn(N, 3, H, W, generator=g)).clamp(0, 1)
    img_u8 = (img * 255).round().to(torch.uint8)
    score = F.avg_pool2d(img[:, 0:1] - img[:, 2:3], 9, 1, 4)
    lab = (score > score.flatten(1).median(1).values.view(N, 1, 1, 1)).to(torch.int64)
    lab[:, :, :6, :] = 255
    lab[:, :, :, -5:] = 255
    return img_u8, lab


def randomize(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and n.endswith('weight') and 'act' not in n:
                p.copy_(0.7 + 0.6 * torch.rand(p.shape, generator=g))
            elif p.dim() == 1 and n.endswith('bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif 'relative_position_bias_table' in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))


def build_model():
    import led_net_amd as L
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = MIN_KEPT
    model = L.MODELS.build(cfg['model'])
    randomize(model, 3)
    return model, cfg


def poly_lr(it):
    return LR * (1.0 - min(it, MAX_ITERS) / MAX_ITERS) ** POWER


def oracle_curve(steps=STEPS, log=None):
    from oracle import spec
    model, _ = build_model()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    img, lab = problem()
    x = spec.preprocess(img)
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k}
    opt, rows = None, []
    for it in range(steps):
        for v in leaves.values():
            v.grad = None
        out = spec.loss(x, lab, sd, loss_cfg=((0.9, MIN_KEPT, 1.0), (0.9, MIN_KEPT, 0.4)))
        (out['decode.loss_context'] + out['decode.loss_spatial']).backward()
        if opt is None:         # parameters without a gradient are skipped, as torch.optim.SGD skips grad=None
            opt = torch.optim.SGD([v for v in leaves.values() if v.grad is not None], lr=LR, momentum=MOM, weight_decay=WD)
        for gr in opt.param_groups:
            gr['lr'] = poly_lr(it)
        opt.step()
        rows.append([float(out[k].detach()) for k in ("decode.loss_context", "decode.loss_spatial", "decode.acc_seg")])
        if log:
            log(it, rows[-1])
    return np.asarray(rows, np.float64)


if __name__ == '__main__':
    import time
    t0 = time.time()
    curve = oracle_curve(log=lambda it, r: print(it, ['%.5f' % v for v in r], f'{time.time() - t0:.0f}s', flush=True) if it % 10 == 0 or it == STEPS - 1 else None)
    img, lab = problem()
    meta = dict(steps=STEPS, lr=LR, momentum=MOM, weight_decay=WD, max_iters=MAX_ITERS, power=POWER, min_kept=MIN_KEPT,
                batch=list(img.shape), img_sum=int(img.sum(dtype=torch.int64)), lab_sum=int(lab.sum()),
                torch=torch.__version__, threads=torch.get_num_threads(),
                ref='configs/LED_Net/LEDNet_80k_cityscapes-1024x1024.py:61-75; oracle/spec.py loss()')
    np.savez_compressed(OUT, curve=curve, meta=json.dumps(meta))
    print('wrote', OUT, curve[0], curve[-1])
