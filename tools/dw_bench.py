#!/usr/bin/env python3
"""Depthwise 3x3 weight gradient at the step's shapes (the second SESP stage: dilations 2..5, 16 x 128 x 128 x C)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import led_net_amd as L  # noqa: E402,F401
from led_net_amd import ops_train as T  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(2)
for C, dil in ((64, (2, 2, 2, 2)), (64, (2, 3, 4, 5)), (128, (2, 3, 4, 5))):
    x = torch.randn((16, 128, 128, C), generator=g).to(dev, torch.bfloat16)
    dz = torch.randn((16, 128, 128, C), generator=g).to(dev, torch.bfloat16)
    w = torch.randn((3, 3, C), generator=g).to(dev)
    dw = torch.zeros((3, 3, C), device=dev)
    fn = lambda: T.dwconv2d_bwd(x, dz, w, dil=dil, group_size=C // 4, need_dx=False, dw_out=dw)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    # replayed from a graph: eager calls through the Python wrapper cost ~30 us of host time each and hide any kernel below that
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(30):
                fn()
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f'dw3x3 wgrad C{C} dil{dil} 16x128x128: {us:.1f} us  {2 * x.numel() * 2 / us * 1e-3:.0f} GB/s', flush=True)
