"""Micro-benchmark of the HBM-bound streaming entries (BatchNorm fwd/bwd, depthwise, SESP pyramid)
on the shapes of the bs16 1024x1024 train step.

    python tools/stream_bench.py [--iters 20] [--only substr]

One line per op: HIP-event time per call (includes the second-stage reduction kernels) and
algorithmic GB/s (every tensor read / written once).  GPU only.
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module('led_net_amd')
from led_net_amd import ops, ops_train  # noqa: E402


def bench(name, fn, nbytes, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f'{name:44s} {us:8.1f} us  {nbytes / us * 1e-3:7.0f} GB/s', flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    bf = torch.bfloat16

    def rnd(*shape, dtype=bf):
        return torch.randn(shape, device=dev).to(dtype)

    cases = []
    for C, N, H in ((64, 16, 128), (32, 16, 256), (32, 16, 512), (128, 16, 128), (128, 16, 64)):
        x, dy = rnd(N, H, H, C), rnd(N, H, H, C)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        mean, invstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        nb = x.numel() * 2
        tag = f'C{C} {N}x{H}x{H}'
        cases.append((f'affine_act relu {tag}', lambda x=x, sc=sc, sh=sh: ops.affine_act(x, sc, sh, act=ops.ACT_RELU), 2 * nb))
        cases.append((f'bn_act_bwd relu {tag}', lambda x=x, dy=dy, sc=sc, sh=sh, mean=mean, invstd=invstd:
                      ops_train.bn_act_bwd(x, dy, scale=sc, shift=sh, mean=mean, invstd=invstd, act=ops.ACT_RELU), 5 * nb))
        cases.append((f'channel_stats {tag}', lambda x=x: ops.channel_stats(x), nb))
    for C, N, H in ((64, 16, 128), (128, 16, 128), (256, 16, 64)):
        x, dz = rnd(N, H, H, C), rnd(N, H, H, C)
        w = torch.randn(3, 3, C, device=dev) * 0.3
        st = (torch.zeros(C, device=dev), torch.zeros(C, device=dev))
        dil, gs = (2, 3, 4, 5), C // 4
        nb = x.numel() * 2
        tag = f'C{C} {N}x{H}x{H}'
        cases.append((f'dwconv3x3 +stats {tag}', lambda x=x, w=w, st=st, dil=dil, gs=gs:
                      ops.dwconv2d(x, w, dil=dil, group_size=gs, stats=st), 2 * nb))
        cases.append((f'dwconv3x3 bwd(data+weight) {tag}', lambda x=x, dz=dz, w=w, dil=dil, gs=gs:
                      ops_train.dwconv2d_bwd(x, dz, w, dil=dil, group_size=gs), 4 * nb))
    for n, N, H in ((16, 16, 128), (32, 16, 128), (64, 16, 64)):
        x, dy = rnd(N, H, H, n), rnd(N, H, H, 4 * n)
        w = torch.randn(4, 3, 3, n, device=dev) * 0.3
        nb = x.numel() * 2
        tag = f'n{n} {N}x{H}x{H}'
        cases.append((f'sesp_pyramid {tag}', lambda x=x, w=w: ops.sesp_pyramid(x, w, (1, 2, 3, 4), 1), 5 * nb))
        cases.append((f'sesp_pyramid bwd {tag}', lambda x=x, dy=dy, w=w:
                      ops_train.sesp_pyramid_bwd(x, dy, w, (1, 2, 3, 4), 1), 2 * nb + 2 * 4 * nb + 4 * nb))
    for name, fn, nbytes in cases:
        if args.only and args.only not in name:
            continue
        bench(name, fn, nbytes, args.iters)


if __name__ == '__main__':
    main()
