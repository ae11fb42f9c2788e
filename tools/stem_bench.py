#!/usr/bin/env python3
"""First stem convolution at the bench sizes: ledn_im2col_stem_planar + ledn_conv2d (two kernels, the patch matrix in
HBM) against ledn_stem_conv in its LDS-window form (LEDN_OPT_STREAM_FAST 27) and its register-direct form (91)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import led_net_amd as L  # noqa: E402,F401
from led_net_amd import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
lib = _lib.get_lib()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, mode in ((16, 'train'), (8, 'infer')):
    x = torch.randint(0, 256, (N, 3, 1024, 1024), dtype=torch.uint8, generator=g).to(dev)
    w = (0.2 * torch.randn(32, 3, 3, 3, generator=g)).to(dev)
    sc, sh = torch.rand(3, device=dev) * 0.02, torch.randn(3, device=dev)
    cmap = torch.tensor([2, 1, 0], dtype=torch.int32, device=dev)
    w1 = ops.stem_weight_as_1x1(w)
    wp = ops.pack_conv_weights(w1, 0)
    st = (torch.zeros(32, device=dev), torch.zeros(32, device=dev))
    osc, osh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    if mode == 'train':
        two = lambda: ops.conv2d(ops.im2col_stem_planar(x, sc, sh, cmap), w1, stats=st, w_bf16=wp)
        one = lambda: ops.stem_conv(x, wp, sc, sh, cmap, stats=st)
    else:
        two = lambda: ops.conv2d(ops.im2col_stem_planar(x, sc, sh, cmap), w1, out_scale=osc, out_shift=osh, act=ops.ACT_RELU, w_bf16=wp)
        one = lambda: ops.stem_conv(x, wp, sc, sh, cmap, out_scale=osc, out_shift=osh, act=ops.ACT_RELU)
    t2 = timeit(two)
    lib.set_option(2, 27)
    t1a = timeit(one)
    lib.set_option(2, 91)
    t1b = timeit(one)
    lib.set_option(2, -1)
    if mode == 'train':
        dz = torch.randn((N, 512, 512, 32), generator=g).to(dev, torch.bfloat16)
        dw = torch.zeros(32, 3, 3, 3, device=dev)
        pat = ops.im2col_stem_planar(x, sc, sh, cmap)
        tw2 = timeit(lambda: ops.conv2d_wgrad(pat, dz, (32, 32, 1, 1)))
        tw1 = timeit(lambda: ops.stem_conv_wgrad(x, dz, dw, sc, sh, cmap))
        print(f'stem wgrad {N}x3x1024x1024: 1x1 weight gradient on the patch matrix {tw2:.1f} us | from the planar batch {tw1:.1f} us', flush=True)
    print(f'stem {mode} {N}x3x1024x1024: im2col + GEMM {t2:.1f} us | stem_conv LDS window {t1a:.1f} us | register-direct {t1b:.1f} us', flush=True)
