#!/usr/bin/env python3
"""Inference-speed benchmark with the command line and protocol of the reference's
tools/analysis_tools/benchmark.py (:22-118): batch size 1, `mode='predict'`, the first 5 iterations skipped,
200 iterations, a device synchronisation around every call, "Overall fps", results dumped as
work_dir/fps_<timestamp>.json with the same keys.  Datasets are out of scope here:
the images are synthetic uint8 batches of the requested size.

    python tools/analysis_tools/benchmark.py CONFIG [CHECKPOINT] [--repeat-times N] [--height 1024 --width 1024]
"""
import argparse
import json
import os
import os.path as osp
import sys
import time

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import led_net_amd as L  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='LED-Net (HIP) benchmark a model')
    p.add_argument('config', help='test config file path')
    p.add_argument('checkpoint', nargs='?', default=None, help='checkpoint file (optional)')
    p.add_argument('--log-interval', type=int, default=50, help='interval of logging')
    p.add_argument('--work-dir', help='if specified, the results will be dumped into the directory as json')
    p.add_argument('--repeat-times', type=int, default=1)
    p.add_argument('--height', type=int, default=1024)
    p.add_argument('--width', type=int, default=1024)
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    p.add_argument('--graph', action='store_true',
                   help='replay the predict pass (backbone + head + fused resize/argmax) as one hipGraph per image')
    return p.parse_args()


def main():
    args = parse_args()
    work_dir = args.work_dir or osp.join('./work_dirs', osp.splitext(osp.basename(args.config))[0])
    os.makedirs(osp.abspath(work_dir), exist_ok=True)
    json_file = osp.join(work_dir, f'fps_{time.strftime("%Y%m%d_%H%M%S", time.localtime())}.json')
    result = dict(config=args.config, unit='img / s')
    fps_list = []
    dev = torch.device('cuda:0')
    for run in range(args.repeat_times):
        print(f'Run {run + 1}:')
        ckpt = args.checkpoint if args.checkpoint and osp.exists(args.checkpoint) else None
        model = L.init_model(args.config, ckpt, device=dev)          # eval mode, SyncBN == BN in eval
        model.set_act_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
        g = torch.Generator().manual_seed(304 + run)
        num_warmup, total_iters, pure = 5, 200, 0.0
        graph = static = None
        if args.graph:
            static = torch.zeros((1, 3, args.height, args.width), dtype=torch.uint8, device=dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(3):
                    model.decode_head.predict_with_mask(model.extract_feat(static))
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                static_out = model.decode_head.predict_with_mask(model.extract_feat(static))   # noqa: F841
        for i in range(total_iters):
            img = torch.randint(0, 256, (1, 3, args.height, args.width), dtype=torch.uint8, generator=g).to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if graph is not None:
                static.copy_(img)
                graph.replay()
            else:
                with torch.no_grad():
                    model(img, None, mode='predict')
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if i >= num_warmup:
                pure += dt
                if (i + 1) % args.log_interval == 0:
                    print(f'Done image [{i + 1:<3}/ {total_iters}], fps: {(i + 1 - num_warmup) / pure:.2f} img / s')
        fps = (total_iters - num_warmup) / pure
        print(f'Overall fps: {fps:.2f} img / s\n')
        result[f'overall_fps_{run + 1}'] = round(fps, 2)
        fps_list.append(fps)
    result['average_fps'] = round(float(np.mean(fps_list)), 2)
    result['fps_variance'] = round(float(np.var(fps_list)), 4)
    print(f'Average fps of {args.repeat_times} evaluations: {result["average_fps"]}')
    print(f'The variance of {args.repeat_times} evaluations: {result["fps_variance"]}')
    with open(json_file, 'w') as fh:
        json.dump(result, fh, indent=4)


if __name__ == '__main__':
    main()
