"""Micro-benchmark of the depthwise / SESP-pyramid kernels at the training step's shapes (bf16, batch 16): forward, data
gradient, weight gradient, each launch timed with HIP events (ops.start_timing) over rotating buffer sets whose total
exceeds the 256 MB Infinity Cache, so that every call streams from HBM as it does inside the step.
    python tools/stencil_bench.py [--iters 24]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=24)
    args = ap.parse_args()
    import importlib
    importlib.import_module('led_net_amd')
    from led_net_amd import ops, ops_train as T
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(1)
    bf = torch.bfloat16

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(bf).to(dev)
    cases = [('pyr', 16, 128, 128, 16, [1, 1, 1, 1], 1), ('pyr', 16, 128, 128, 32, [1, 1, 1, 1], 1),
             ('pyr', 16, 128, 128, 32, [1, 2, 3, 4], 1), ('pyr', 16, 64, 64, 64, [1, 2, 3, 4], 1),
             ('dw', 16, 128, 128, 64, [2, 2, 2, 2], 16), ('dw', 16, 128, 128, 128, [2, 2, 2, 2], 32),
             ('dw', 16, 128, 128, 128, [2, 3, 4, 5], 32), ('dw', 16, 64, 64, 256, [2, 3, 4, 5], 64)]
    for kind, N, H, W, n, dil, gs in cases:
        nset = max(2, int(400e6 // (N * H * W * (n * (10 if kind == 'pyr' else 4)))) + 1)
        sets = []
        for _ in range(nset):
            if kind == 'pyr':
                sets.append((rnd(N, H, W, n), rnd(N, H, W, 4 * n), (0.3 * torch.randn(4, 3, 3, n, generator=g)).to(dev)))
            else:
                sets.append((rnd(N, H, W, n), rnd(N, H, W, n), (0.3 * torch.randn(3, 3, n, generator=g)).to(dev)))
        torch.cuda.synchronize()
        ops.start_timing()
        for it in range(args.iters):
            x, dy, w = sets[it % nset]
            if kind == 'pyr':
                ops.sesp_pyramid(x, w, dil, 1)
                T.sesp_pyramid_bwd(x, dy, w, dil, 1)
            else:
                ops.dwconv2d(x, w, dil=dil, group_size=gs)
                T.dwconv2d_bwd(x, dy, w, dil=dil, group_size=gs)
        torch.cuda.synchronize()
        rec = ops.stop_timing()
        agg = {}
        for r in rec:
            agg.setdefault(r['entry'], []).append(r['ms'] * 1e3)
        line = []
        for e, ts in agg.items():
            ts = sorted(ts[len(ts) // 4:])          # the first quarter warms up
            line.append(f'{e[5:]} {ts[len(ts) // 2]:6.1f}')
        print(f'{kind} {N}x{H}x{W} n{n} dil{dil}: ' + ' | '.join(line) + ' us', flush=True)


if __name__ == '__main__':
    main()
