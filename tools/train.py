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
    p.add_argument('--eager', action='store_true', help='per-kernel launches instead of the captured hipGraph of the step')
    p.add_argument('--capture-warmup', type=int, default=3,
                   help='eager warm-up steps before the graph capture (rolled back: they do not count as iterations)')
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
    # Batches are made ON THE DEVICE (torch's device generator / the augmentation kernel), on a side stream, one
    # iteration ahead of the step that consumes them; each is a function of (seed, rank, iteration) only, so a resumed
    # run sees the batches the uninterrupted run would have seen.  The step itself is ONE hipGraph replay
    # (Trainer.capture / replay: what bench.py times); --eager keeps the per-kernel launches.
    gdev = torch.Generator(device=dev)
    side = torch.cuda.Stream(device=dev)
    H, W = args.height, args.width

    def make_batch(it):
        gdev.manual_seed(304 + 7919 * rank + 1000003 * it)
        if pipe is not None:                # "decoded" H x 2W images + label maps -> augmented crops, one launch
            np.random.seed((304 + 7919 * rank + 1000003 * it) % (2 ** 32))
            raw = torch.randint(0, 256, (bs, H, 2 * W, 3), dtype=torch.uint8, device=dev, generator=gdev)
            seg = torch.randint(0, 2, (bs, H // 8, W // 4), dtype=torch.uint8, device=dev, generator=gdev)
            seg = seg.repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous()
            aug = pipe.batch([dict(img=raw[i], gt_seg_map=seg[i]) for i in range(bs)], out_hw=(H, W))
            # padded_samples carry img_shape / pad_shape / padding_size: the stem then writes pad_val in the
            # NORMALISED domain over the padded area, as SegDataPreProcessor.forward(training=True) + stack_batch do
            return aug['batch'], aug['padded_samples']
        img = torch.randint(0, 256, (bs, 3, H, W), dtype=torch.uint8, device=dev, generator=gdev)
        lab = torch.randint(0, 2, (bs, 1, H, W), dtype=torch.int64, device=dev, generator=gdev)
        lab[:, :, :16], lab[:, :, -16:], lab[..., :16], lab[..., -16:] = 255, 255, 255, 255
        return img, [L.SegDataSample(gt=lab[i]) for i in range(bs)]

    def prefetch(it):
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            b = make_batch(it)
        ev = torch.cuda.Event()
        ev.record(side)
        return b, ev

    def save(it):
        L.save_checkpoint(model, osp.join(work_dir, f'iter_{it}.pth'), trainer=trainer,
                          meta=dict(iter=it, dataset_meta=dict(classes=('background', 'foreground'), palette=None)))

    use_graph = not args.eager
    if use_graph and trainer.dist is not None and trainer.comm is None:
        # 'auto' collectives fell back to torch.distributed: its all-reduces cannot be recorded into a hipGraph
        print('[tools/train.py] direct RCCL communicator unavailable: training with eager launches (as --eager)',
              file=sys.stderr, flush=True)
        use_graph = False
    if start < max_iters and use_graph:
        # capture on the first batch, free of side effects (restore=True: weights, momentum, running statistics and
        # the iteration counter are put back after the warm-up steps): the loop below replays every iteration of
        # the schedule exactly once, and a resumed run continues as the uninterrupted one would have
        (img, samples), ev = prefetch(start)
        ev.wait()
        trainer.capture(img, samples, warmup=args.capture_warmup, restore=True)
    t0, tlog = time.perf_counter(), time.perf_counter()
    first = trainer.iter
    nxt = prefetch(first) if first < max_iters else None
    for it in range(first, max_iters):
        (img, samples), ev = nxt
        ev.wait()                           # the launch stream waits for the batch (device-side dependency)
        nxt = prefetch(it + 1) if it + 1 < max_iters else None
        out = trainer.replay(img, samples) if use_graph else trainer.train_step(img, samples)
        if rank == 0 and ((it + 1) % 50 == 0 or it + 1 == max_iters):
            vals = {k: float(v.float().reshape(-1)[0]) for k, v in out.items()}
            dt = (time.perf_counter() - tlog) / max(1, min(50, it + 1 - first))
            tlog = time.perf_counter()
            print(f'Iter(train) [{it + 1:6d}/{max_iters}]  lr: {trainer.lr():.4e}  time: {dt:.4f}  '
                  + '  '.join(f'{k}: {v:.4f}' for k, v in vals.items())
                  + f'  data: {int(img.sum(dtype=torch.int64))}', flush=True)     # fingerprint of this iteration's batch
        if rank == 0 and args.save_interval and (it + 1) % args.save_interval == 0 and it + 1 < max_iters:
            save(it + 1)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0           # (the final checkpoint below is not part of the training throughput)
    if rank == 0 and max_iters > first:
        save(max_iters)
    if rank == 0:
        n = max_iters - first
        print(f'{n} iterations, {bs * world * n / max(el, 1e-9):.1f} images/s '
              f'({"hipGraph replay" if use_graph else "eager launches"}, incl. on-device batch generation)', flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
