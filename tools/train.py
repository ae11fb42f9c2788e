#!/usr/bin/env python3
"""Training entry point with the reference's command line (tools/train.py:15-60: CONFIG, --work-dir,
--resume, --amp, --cfg-options, --launcher, --local_rank) on the built-in loop: config's SGD + PolyLR,
OhemCrossEntropy x2, SyncBN when launched with --launcher pytorch (one process per GPU, RCCL), loss line
every 50 iterations, checkpoints in mmengine layout (iter_N.pth).  Datasets/augmentation are out of scope
(SURVEY.md section 2): batches are synthetic Cityscapes-shaped uint8 images with a 16-px ignore border.

    python tools/train.py CONFIG [--work-dir DIR] [--max-iters N] [--batch-size B] [--resume]
    python -m torch.distributed.run --nproc-per-node 8 tools/train.py CONFIG --launcher pytorch
"""
import argparse
import ast
import glob
import os
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import led_net_amd as L  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='Train LED-Net (HIP)')
    p.add_argument('config')
    p.add_argument('--work-dir')
    p.add_argument('--resume', action='store_true', help='resume from the latest iter_*.pth in the work dir')
    p.add_argument('--amp', action='store_true', help='bf16 activations (the default here; kept for CLI parity)')
    p.add_argument('--cfg-options', nargs='+', default=[], help='key=value overrides (model.backbone.channels=32 ...)')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    p.add_argument('--local_rank', '--local-rank', type=int, default=0)
    p.add_argument('--max-iters', type=int, default=None, help='default: the schedule length of the config')
    p.add_argument('--batch-size', type=int, default=16, help='images per GPU')
    p.add_argument('--height', type=int, default=1024)
    p.add_argument('--width', type=int, default=1024)
    p.add_argument('--save-interval', type=int, default=0)
    p.add_argument('--f32', action='store_true')
    p.add_argument('--augment', action='store_true',
                   help="run the config's train_pipeline (RandomResize / RandomCrop / RandomFlip / PhotoMetricDistortion) on the "
                        'GPU over synthetic decoded images (H x 2W uint8 BGR) instead of feeding ready-made crops')
    return p.parse_args()


def main():
    args = parse_args()
    world, rank = 1, 0
    if args.launcher == 'pytorch':
        import torch.distributed as dist
        dist.init_process_group('nccl')
        world, rank = dist.get_world_size(), dist.get_rank()
        args.local_rank = int(os.environ.get('LOCAL_RANK', args.local_rank))
    torch.cuda.set_device(args.local_rank)
    dev = torch.device('cuda', args.local_rank)
    cfg = L.load_config(args.config)
    for kv in args.cfg_options:
        key, val = kv.split('=', 1)
        node = cfg
        parts = key.split('.')
        for part in parts[:-1]:
            node = node[int(part)] if isinstance(node, list) else node[part]
        try:
            val = ast.literal_eval(val)   # numbers / tuples / lists / None / booleans, as mmengine's DictAction
        except (ValueError, SyntaxError):  # plain string
            pass
        node[parts[-1]] = val
    work_dir = args.work_dir or osp.join('./work_dirs', osp.splitext(osp.basename(args.config))[0])
    os.makedirs(work_dir, exist_ok=True)
    torch.manual_seed(304)
    model = L.MODELS.build(cfg['model'])
    model.set_act_dtype(torch.float32 if args.f32 else torch.bfloat16)
    start, ckpt = 0, None
    if args.resume:
        cks = sorted(glob.glob(osp.join(work_dir, 'iter_*.pth')), key=lambda f: int(osp.basename(f)[5:-4]))
        if cks:
            ckpt = L.load_checkpoint(model, cks[-1])
            start = ckpt['meta'].get('iter', 0)
            print(f'resumed from {cks[-1]} (iter {start})')
    model.to(dev)
    trainer = L.Trainer(model, cfg, world_size=world, max_iters=None)
    if ckpt is not None:
        L.resume(trainer, ckpt)           # momentum buffers + PolyLR position (mmengine 'optimizer' / 'param_schedulers')
    trainer.iter = start
    max_iters = args.max_iters or trainer.max_iters
    g = torch.Generator().manual_seed(304 + rank)
    bs = args.batch_size
    pipe = None
    if args.augment:
        import numpy as np
        from led_net_amd import transforms as T
        if 'train_pipeline' not in cfg:
            raise SystemExit(f'--augment: {args.config} defines no train_pipeline')
        pipe_cfg = [dict(c) for c in cfg['train_pipeline']]
        for c in pipe_cfg:                  # the CLI's crop size wins over the dataset config's
            if c['type'] == 'RandomCrop':
                c['crop_size'] = (args.height, args.width)
            elif c['type'] == 'RandomResize':
                c['scale'] = (2 * args.width, args.height)
        pre = cfg['model'].get('data_preprocessor') or {}
        pipe = T.Compose(pipe_cfg, pad_val=pre.get('pad_val', 0), seg_pad_val=pre.get('seg_pad_val', 255))
        np.random.seed(304 + rank)
    t0, tlog = time.perf_counter(), time.perf_counter()
    for it in range(start, max_iters):
        if pipe is not None:                # "decoded" H x 2W images + label maps -> augmented crops, one launch
            raw = torch.randint(0, 256, (bs, args.height, 2 * args.width, 3), dtype=torch.uint8, generator=g).to(dev)
            seg = torch.randint(0, 2, (bs, args.height // 8, args.width // 4), dtype=torch.uint8, generator=g).to(dev)
            seg = seg.repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous()
            aug = pipe.batch([dict(img=raw[i], gt_seg_map=seg[i]) for i in range(bs)], out_hw=(args.height, args.width))
            img, lab = aug['batch'], aug['labels']
        else:
            img = torch.randint(0, 256, (bs, 3, args.height, args.width), dtype=torch.uint8, generator=g).to(dev)
            lab = torch.randint(0, 2, (bs, 1, args.height, args.width), dtype=torch.int64, generator=g)
            lab[:, :, :16], lab[:, :, -16:], lab[..., :16], lab[..., -16:] = 255, 255, 255, 255
            lab = lab.to(dev)
        out = trainer.train_step(img, [L.SegDataSample(gt=lab[i]) for i in range(bs)])
        if rank == 0 and ((it + 1) % 50 == 0 or it + 1 == max_iters):
            vals = {k: float(v.float().reshape(-1)[0]) for k, v in out.items()}
            dt = (time.perf_counter() - tlog) / min(50, it + 1 - start)
            tlog = time.perf_counter()
            print(f'Iter(train) [{it + 1:6d}/{max_iters}]  lr: {trainer.lr():.4e}  time: {dt:.4f}  '
                  + '  '.join(f'{k}: {v:.4f}' for k, v in vals.items()), flush=True)
        if rank == 0 and ((args.save_interval and (it + 1) % args.save_interval == 0) or it + 1 == max_iters):
            L.save_checkpoint(model, osp.join(work_dir, f'iter_{it + 1}.pth'), trainer=trainer,
                              meta=dict(iter=it + 1, dataset_meta=dict(classes=('background', 'foreground'), palette=None)))
    if rank == 0:
        n = max_iters - start
        print(f'{n} iterations, {bs * world * n / (time.perf_counter() - t0):.1f} images/s (incl. synthetic data generation)')
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
