"""Micro-benchmark of the conv entries on the shapes of the bs16 1024x1024 train step.

    python tools/conv_bench.py [--iters 20]

Prints one line per shape: HIP-event time per launch, algorithmic GB/s (unique activation bytes
in + out) and TFLOP/s.  GPU only (the product path has no CPU fallback).
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ledn = importlib.import_module('led_net_amd')
from led_net_amd import ops  # noqa: E402

# (kind, Cin, Cout, k, stride, groups, N, H, W)   H, W = forward input size
SHAPES = [
    ('fwd', 32, 32, 3, 1, 1, 16, 256, 256),
    ('dgrad', 32, 32, 3, 1, 1, 16, 256, 256),
    ('wgrad', 32, 32, 3, 1, 1, 16, 256, 256),
    ('fwd', 32, 32, 3, 1, 1, 16, 512, 512),          # (not a layer of the network: the 1/4-resolution kernel at 4 x the pixels)
    ('fwd', 64, 64, 3, 1, 1, 16, 128, 128),
    ('dgrad', 64, 64, 3, 1, 1, 16, 128, 128),
    ('wgrad', 64, 64, 3, 1, 1, 16, 128, 128),
    ('fwd', 32, 32, 1, 1, 1, 16, 512, 512),
    ('wgrad', 32, 32, 1, 1, 1, 16, 512, 512),
    ('fwd', 32, 32, 3, 2, 1, 16, 512, 512),
    ('dgrad', 32, 32, 3, 2, 1, 16, 512, 512),
    ('wgrad', 32, 32, 3, 2, 1, 16, 512, 512),
    ('fwd', 32, 2, 3, 1, 1, 16, 512, 512),
    ('fwdraw', 32, 2, 3, 1, 1, 16, 512, 512),        # no statistics, no prologue
    ('fwdpro', 32, 2, 3, 1, 1, 16, 512, 512),        # training: BatchNorm + ReLU prologue, bias, raw bf16 output, no statistics
    ('dgrad', 32, 2, 3, 1, 1, 16, 512, 512),
    ('wgrad', 32, 2, 3, 1, 1, 16, 512, 512),
    ('wgradpro', 32, 2, 3, 1, 1, 16, 512, 512),      # training: BatchNorm + ReLU prologue, bias gradient
    ('fwd', 64, 64, 1, 1, 4, 16, 128, 128),
    ('dgrad', 64, 64, 1, 1, 4, 16, 128, 128),
    ('wgrad', 64, 64, 1, 1, 4, 16, 128, 128),
    ('fwd', 64, 16, 1, 1, 4, 16, 128, 128),
    ('dgrad', 64, 16, 1, 1, 4, 16, 128, 128),
    ('fwd', 16, 64, 1, 1, 1, 16, 128, 128),
    ('fwd', 128, 128, 1, 1, 4, 16, 128, 128),
    ('dgrad', 128, 128, 1, 1, 4, 16, 128, 128),
    ('fwd', 128, 64, 1, 1, 1, 16, 64, 64),
    ('fwd', 128, 32, 1, 1, 1, 16, 64, 64),
    ('fwd', 128, 64, 3, 1, 1, 16, 128, 128),
    ('fwd', 128, 512, 1, 1, 1, 16, 64, 64),
    ('fwd', 512, 128, 1, 1, 1, 16, 64, 64),
    ('wgrad', 128, 512, 1, 1, 1, 16, 64, 64),
    ('wgrad', 64, 2, 1, 1, 1, 16, 128, 128),
    ('wgrad', 64, 16, 1, 1, 1, 16, 16, 16),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--conv-wgs', type=int, default=0, help='LEDN_OPT_CONV_WORKGROUPS (0 = default)')
    ap.add_argument('--wgrad-wgs', type=int, default=0, help='LEDN_OPT_WGRAD_WORKGROUPS (0 = default)')
    ap.add_argument('--only', default='', help='substring filter on the kind (fwd/dgrad/wgrad)')
    ap.add_argument('--k', type=int, default=0, help='only this kernel size')
    ap.add_argument('--ab', default='', help='comma list of LEDN_OPT_STREAM_FAST values to time each shape under')
    args = ap.parse_args()
    from led_net_amd import _lib
    _lib.get_lib().set_option(0, args.conv_wgs)
    _lib.get_lib().set_option(1, args.wgrad_wgs)
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(1)
    for kind, ci, co, k, s, grp, N, H, W in SHAPES:
        if (args.only and args.only not in kind) or (args.k and args.k != k):
            continue
        pad = k // 2
        Ho, Wo = ops.conv_out_size(H, k, s, pad, 1), ops.conv_out_size(W, k, s, pad, 1)
        x = torch.randn((N, H, W, ci), generator=g).to(dev, torch.bfloat16)
        dz = torch.randn((N, Ho, Wo, co), generator=g).to(dev, torch.bfloat16)
        w = (torch.randn((co, ci // grp, k, k), generator=g) * 0.1).to(dev)
        stats = (torch.zeros(co, device=dev), torch.zeros(co, device=dev))
        if kind == 'fwd':
            wp = ops.pack_conv_weights(w, 0, grp) if ops.mfma_weight_ok(w, grp) else None
            fn = lambda: ops.conv2d(x, w, stride=s, pad=pad, groups=grp, stats=stats, w_bf16=wp)
            nbytes = x.numel() * 2 + dz.numel() * 2
        elif kind == 'fwdraw':
            wp = ops.pack_conv_weights(w, 0, grp)
            fn = lambda: ops.conv2d(x, w, stride=s, pad=pad, groups=grp, w_bf16=wp)
            nbytes = x.numel() * 2 + dz.numel() * 2
        elif kind == 'fwdpro':
            wp = ops.pack_conv_weights(w, 0, grp)
            isc, ish, bias = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1, torch.randn(co, device=dev)
            fn = lambda: ops.conv2d(x, w, stride=s, pad=pad, groups=grp, in_scale=isc, in_shift=ish, in_act=ops.ACT_RELU,
                                    out_shift=bias, w_bf16=wp)
            nbytes = x.numel() * 2 + dz.numel() * 2
        elif kind == 'wgradpro':
            isc, ish = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
            fn = lambda: ops.conv2d_wgrad(x, dz, tuple(w.shape), stride=s, pad=pad, groups=grp, in_scale=isc, in_shift=ish,
                                          in_act=ops.ACT_RELU, bias=True)
            nbytes = x.numel() * 2 + dz.numel() * 2
        elif kind == 'dgrad':
            wp = ops.pack_conv_weights(w, 1, grp) if ops.mfma_weight_ok(w, grp) else None
            fn = lambda: ops.conv2d(dz, w, stride=s, pad=pad, groups=grp, transposed=True, out_hw=(H, W), w_bf16=wp)
            nbytes = x.numel() * 2 + dz.numel() * 2
        else:
            fn = lambda: ops.conv2d_wgrad(x, dz, tuple(w.shape), stride=s, pad=pad, groups=grp)
            nbytes = x.numel() * 2 + dz.numel() * 2
        flops = 2.0 * N * Ho * Wo * co * (ci // grp) * k * k
        line = f'{kind:5s} {k}x{k} {ci:3d}->{co:3d} g{grp} s{s} {N}x{H}x{W}:'
        for opt in ([int(v) for v in args.ab.split(',')] if args.ab else [None]):
            if opt is not None:
                _lib.get_lib().set_option(2, opt)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            line += (f' [{opt}]' if opt is not None else '') + (f'{us:8.1f} us  {nbytes / us * 1e-3:7.0f} GB/s  '
                                                                  f'{flops / us * 1e-6:6.1f} TFLOP/s')
        print(line, flush=True)


if __name__ == '__main__':
    main()
