#!/usr/bin/env python3
"""Measured forward parity of the HIP path against the CPU oracle (GPU box; the numbers DESIGN.md and the
tolerances of tests/test_bf16.py / tests/test_parity_argmax.py quote):
  f32 activations: max |logit error|, argmax flips by oracle-margin band;
  bf16 activations: max / mean |logit error| relative to the logit scale, argmax disagreement by margin band.
    python tools/measure_parity.py [--size 512] [--batch 2] > gpurun_out/parity.json
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def measure(dtype, size, batch, seed):
    import led_net_amd as L
    from oracle import spec                      # measurement tool = checker side, not the product path
    from test_blocks import _randomize
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    model = L.MODELS.build(cfg['model']).eval()
    _randomize(model, seed)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.set_act_dtype(dtype)
    model.to('cuda:0')
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (batch, 3, size, size), dtype=torch.uint8, generator=g)
    with torch.no_grad():
        want, want_mask = spec.predict(spec.preprocess(img), sd)
        out = model(img.cuda(), mode='predict')
    logits = torch.stack([o.seg_logits.data for o in out]).cpu()
    mask = torch.cat([o.pred_sem_seg.data for o in out]).long().cpu()
    err = (logits - want).abs()
    scale = want.abs().max().item()
    margin = (want[:, 0] - want[:, 1]).abs()
    flips = mask != want_mask
    rec = dict(dtype=str(dtype)[6:], size=size, batch=batch, seed=seed, pixels=int(mask.numel()), logit_scale=scale,
               max_abs_err=err.max().item(), mean_abs_err=err.mean().item(),
               max_err_over_scale=err.max().item() / scale, mean_err_over_scale=err.mean().item() / scale,
               flips_total=int(flips.sum()), exact_ties_in_oracle=int((margin == 0).sum()))
    for band in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1):
        rec[f'flips_margin_gt_{band:g}'] = int((flips & (margin > band)).sum())
        rec[f'pixels_margin_le_{band:g}'] = int((margin <= band).sum())
    rec['flips_margin_gt_2x_max_err'] = int((flips & (margin > 2 * err.max())).sum())
    for frac in (0.01, 0.05, 0.1):
        rec[f'disagree_frac_margin_gt_{frac:g}_scale'] = float((flips & (margin > frac * scale)).float().mean())
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--batch', type=int, default=2)
    args = ap.parse_args()
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        for seed in (1, 2, 3):
            out.append(measure(dtype, args.size, args.batch, seed))
            print(json.dumps(out[-1]), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
