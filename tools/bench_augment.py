"""Throughput of the GPU augmentation pipeline at the Cityscapes shape of the LED-Net config:
16 decoded 1024 x 2048 images -> RandomResize(0.5-2.0) -> RandomCrop 1024^2 (cat_max_ratio 0.75) -> flip ->
PhotoMetricDistortion -> one uint8 batch; next to the numpy restatement of the reference pipeline on the host.
usage: python tools/bench_augment.py [--batch 16] [--iters 20] [--cpu-samples 4]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--cpu-samples', type=int, default=4)
    a = ap.parse_args()
    import led_net_amd  # noqa: F401
    from led_net_amd import transforms as T
    pipe_cfg = [dict(type='RandomResize', scale=(2048, 1024), ratio_range=(0.5, 2.0), keep_ratio=True),
                dict(type='RandomCrop', crop_size=(1024, 1024), cat_max_ratio=0.75),
                dict(type='RandomFlip', prob=0.5), dict(type='PhotoMetricDistortion'), dict(type='PackSegInputs')]
    pipe = T.Compose(pipe_cfg)
    g = np.random.RandomState(0)
    host = []
    for _ in range(a.batch):
        img = g.randint(0, 256, (1024, 2048, 3)).astype(np.uint8)
        seg = g.randint(0, 2, (128, 256)).astype(np.uint8).repeat(8, 0).repeat(8, 1)
        host.append((img, np.ascontiguousarray(seg)))
    dev = torch.device('cuda:0')
    recs = [dict(img=torch.from_numpy(i).to(dev), gt_seg_map=torch.from_numpy(s).to(dev)) for i, s in host]
    np.random.seed(1)
    for _ in range(3):
        pipe.batch([dict(r) for r in recs], out_hw=(1024, 1024))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = pipe.batch([dict(r) for r in recs], out_hw=(1024, 1024))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    # the pixel kernel alone (parameters fixed)
    from led_net_amd import _lib
    from led_net_amd.ops import _run
    rs = [dict(r) for r in recs]
    out = pipe.batch(rs, out_hw=(1024, 1024))
    tab = T._table([T._entry(r) for r in rs], dev)
    lib = _lib.get_lib()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        _run(lib, 'ledn_augment_batch', out['batch'], tab.data_ptr(), a.batch, out['batch'].data_ptr(),
             out['labels'].data_ptr(), 1024, 1024, 0, 255)
    e1.record()
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / a.iters
    from oracle import augment as OA
    np.random.seed(1)
    t0 = time.perf_counter()
    for i in range(a.cpu_samples):
        OA.train_pipeline(host[i][0], host[i][1], (2048, 1024), (0.5, 2.0), (1024, 1024), 0.75, 0.5)
    cpu = (time.perf_counter() - t0) / a.cpu_samples
    out_bytes = a.batch * 1024 * 1024 * (3 + 8)
    print(json.dumps({'metric': 'augmented images/s (1024x2048 source -> 1024x1024 crop)', 'batch': a.batch,
                      'gpu_pipeline_ms_per_batch': round(dt * 1e3, 3), 'gpu_images_per_s': round(a.batch / dt, 1),
                      'kernel_ms_per_batch': round(kern_ms, 3),
                      'kernel_output_GBps': round(out_bytes / kern_ms / 1e6, 1),
                      'cpu_numpy_restatement_s_per_image': round(cpu, 3),
                      'cpu_images_per_s_one_core': round(1.0 / cpu, 2)}))


if __name__ == '__main__':
    main()
