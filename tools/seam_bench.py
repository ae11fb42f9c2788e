#!/usr/bin/env python3
"""Timing split of ledn_seam_edge at the training size: percentile rule (radix select) vs fixed threshold."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import led_net_amd as L  # noqa: E402
from led_net_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
for shape in ((16, 128, 128, 1), (1, 128, 128, 1), (16, 64, 64, 1)):
    seg = torch.randn(shape, device=dev, generator=g)
    for pct in (0.8, None):
        for _ in range(5):
            ops.seam_edge(seg, pct)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.seam_edge(seg, pct)
        e1.record()
        torch.cuda.synchronize()
        print(f'seam_edge {shape} percentile={pct}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us')
