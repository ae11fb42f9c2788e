#!/usr/bin/env python3
"""Model complexity with the command line and output of the reference's tools/analysis_tools/get_flops.py
(:14-131: CONFIG, --shape H [W] default 1280 720, SyncBN reverted, one `mode='tensor'` forward through the data
preprocessor, "Flops" = multiply-accumulates as mmengine's counter reports them for convolutions and matrix
products, "Params").  The counter here is the library's own launch instrumentation (led_net_amd.ops.start_timing:
every C-ABI call reports the algorithmic flops of its kernel), so what is counted is what runs -- the published
figures it is compared with: 9.206 G / 1.661 M (supplementary PDF p.17, Table 8).

    python tools/analysis_tools/get_flops.py CONFIG [--shape 1280 720] [--cfg-options backbone.cespb_depth=(2,3)]
"""
import argparse
import ast
import os.path as osp
import sys

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import led_net_amd as L  # noqa: E402
from led_net_amd import ops  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='Get the FLOPs of a segmentor')
    p.add_argument('config', help='train config file path')
    p.add_argument('--shape', type=int, nargs='+', default=[1280, 720], help='input image size')
    p.add_argument('--cfg-options', nargs='+', default=[], help='model.* overrides, e.g. backbone.cespb_depth=(2,3)')
    return p.parse_args()


def _fmt(v, units):
    for scale, suffix in units:
        if v >= scale:
            return f'{v / scale:.3f}{suffix}'
    return str(v)


def complexity(cfg, shape, device):
    """-> dict(macs, params, per_kernel={kernel: macs}) of one eval forward (mode='tensor') at 3 x shape"""
    mcfg = dict(cfg['model'])
    mcfg['train_cfg'] = None
    model = L.MODELS.build(mcfg).to(device).eval()          # BN in eval == SyncBN reverted (get_flops.py:81)
    h, w = shape
    img = torch.rand(3, h, w) * 255.0                         # get_flops.py:84-90: one random image through the
    batch = model.data_preprocessor(dict(inputs=[img.to(device)]))['inputs']        # data preprocessor
    with torch.no_grad():
        model(batch.to(torch.uint8), None, mode='tensor')     # warm-up: weight packs / BN folds are not the model
        torch.cuda.synchronize()
        ops.start_timing()
        try:
            model(batch.to(torch.uint8), None, mode='tensor')
            torch.cuda.synchronize()
        finally:
            rec = ops.stop_timing()
    per = {}
    for r in rec:
        if r['flops'] and ('conv' in r['kernel'] or 'attn' in r['kernel']):     # contractions only, as mmengine counts
            per[r['kernel']] = per.get(r['kernel'], 0) + r['flops'] // 2
    params = sum(p.numel() for p in model.parameters())
    return dict(macs=sum(per.values()), params=params, per_kernel=per)


def main():
    args = parse_args()
    if len(args.shape) == 1:
        shape = (args.shape[0], args.shape[0])
    elif len(args.shape) == 2:
        shape = tuple(args.shape)
    else:
        raise ValueError('invalid input shape')
    cfg = L.load_config(args.config)
    for kv in args.cfg_options:
        key, val = kv.split('=', 1)
        node = cfg['model']
        parts = key.split('.')
        for part in parts[:-1]:
            node = node[part]
        try:
            val = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            pass
        node[parts[-1]] = val
    res = complexity(cfg, shape, torch.device('cuda:0'))
    split_line = '=' * 30
    print(f'{split_line}\nCompute type: direct: randomly generate a picture\nInput shape: {shape}\n'
          f'Flops: {_fmt(res["macs"], [(1e9, "G"), (1e6, "M"), (1e3, "K")])}\n'
          f'Params: {_fmt(res["params"], [(1e6, "M"), (1e3, "K")])}\n{split_line}')
    for k, v in sorted(res['per_kernel'].items(), key=lambda kv: -kv[1]):
        print(f'  {k:28s} {v / 1e9:8.4f} GMAC')
    print(f'published (PDF p.17): Flops 9.206G  Params 1.661M  ->  {res["macs"] / 9.206e9 * 100:.1f} % / '
          f'{res["params"] / 1.661e6 * 100:.1f} %')
    print('!!!Please be cautious if you use the results in papers. You may need to check if all ops are supported '
          'and verify that the flops computation is correct.')


if __name__ == '__main__':
    main()
