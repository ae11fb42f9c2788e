#!/usr/bin/env python3
"""Evaluation entry point with the reference's command line (tools/test.py: CONFIG CHECKPOINT [--work-dir])
on the built-in loop: `mode='predict'` -> uint8 argmax masks (fused into the last resize) -> IoUMetric
histograms on the device (ledn_iou_hist) -> the aAcc / mIoU / mAcc summary of the reference's IoUMetric.
Datasets are out of scope: images and labels are synthetic.

    python tools/test.py CONFIG CHECKPOINT [--num-images 32] [--height 1024 --width 1024]
"""
import argparse
import os.path as osp
import sys

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import led_net_amd as L  # noqa: E402


def main():
    p = argparse.ArgumentParser(description='Test (and eval) LED-Net (HIP)')
    p.add_argument('config')
    p.add_argument('checkpoint')
    p.add_argument('--work-dir')
    p.add_argument('--num-images', type=int, default=32)
    p.add_argument('--batch-size', type=int, default=8)
    p.add_argument('--height', type=int, default=1024)
    p.add_argument('--width', type=int, default=1024)
    args = p.parse_args()
    dev = torch.device('cuda:0')
    model = L.init_model(args.config, args.checkpoint if osp.exists(args.checkpoint) else None, device=dev)
    model.set_act_dtype(torch.bfloat16)
    ncls = model.decode_head.num_classes
    metric = L.IoUMetric(ncls, 255, ['mIoU'])
    g = torch.Generator().manual_seed(304)
    for i in range(0, args.num_images, args.batch_size):
        n = min(args.batch_size, args.num_images - i)
        img = torch.randint(0, 256, (n, 3, args.height, args.width), dtype=torch.uint8, generator=g).to(dev)
        lab = torch.randint(0, ncls, (n, args.height, args.width), dtype=torch.int64, generator=g).to(dev)
        with torch.no_grad():
            out = model(img, None, mode='predict')
        metric.process([o.pred_sem_seg.data for o in out], [lab[j] for j in range(n)])
    summary, per_class = metric.compute_metrics()
    classes = getattr(model, 'dataset_meta', {}).get('classes') or [str(c) for c in range(ncls)]
    print('per class results:')
    for c in range(ncls):
        print(f'  {classes[c]:>12s}  IoU {per_class["IoU"][c] * 100:6.2f}  Acc {per_class["Acc"][c] * 100:6.2f}')
    print('  '.join(f'{k}: {v:.2f}' for k, v in summary.items()))


if __name__ == '__main__':
    main()
