#!/usr/bin/env python3
"""PCIe-inclusive rate of the training step (DESIGN section 6 note; never bench.py's `value`): the uint8 batch
(16 x 3 x 1024 x 1024 = 50 MB) and the int64 labels (16 x 1 x 1024 x 1024 = 134 MB) start in PINNED HOST memory every step.
  serial:     copy on the launch stream, then the captured step
  overlapped: the copy of batch i+1 on a side stream while step i runs (two device buffers)
  uint8 labels: the same two forms with the labels as uint8 on the wire (17 MB instead of 134 MB; Cityscapes / VOC label
                maps ARE uint8 files, int64 is what the reference's PackSegInputs converts them to on the host) -- widened to
                the resident int64 label batch by Trainer.replay's device-side copy
    python tools/pcie_rate.py [--steps 30]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
L = importlib.import_module('led_net_amd')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--batch', type=int, default=16)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    torch.manual_seed(304)
    model = L.MODELS.build(cfg['model']).to(dev)
    model.set_act_dtype(torch.bfloat16)
    B = args.batch
    g = torch.Generator().manual_seed(304)
    host = [(torch.randint(0, 256, (B, 3, 1024, 1024), dtype=torch.uint8, generator=g).pin_memory(),
             torch.randint(0, 2, (B, 1, 1024, 1024), dtype=torch.int64, generator=g).pin_memory()) for _ in range(2)]
    devb = [(torch.empty_like(h[0], device=dev), torch.empty_like(h[1], device=dev)) for h in host]

    def samples(lab):
        return [L.SegDataSample(gt=lab[i]) for i in range(B)]
    for k in range(2):
        devb[k][0].copy_(host[k][0]); devb[k][1].copy_(host[k][1])
    tr = L.Trainer(model, cfg, max_iters=100000)
    tr.capture(devb[0][0], samples(devb[0][1]), warmup=2)
    torch.cuda.synchronize()

    def timed(fn, n):
        for _ in range(3):
            fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    t_res = timed(lambda i: tr.replay(), args.steps)

    def serial(i):
        k = i & 1
        devb[k][0].copy_(host[k][0], non_blocking=True)
        devb[k][1].copy_(host[k][1], non_blocking=True)
        tr.replay(devb[k][0], samples(devb[k][1]))
    t_ser = timed(serial, args.steps)

    side = torch.cuda.Stream(device=dev)
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    done = [torch.cuda.Event(), torch.cuda.Event()]

    def overlapped(i):
        k = i & 1
        cur = torch.cuda.current_stream(dev)
        # batch i was uploaded during step i-1 (first call: upload now)
        if i == 0:
            with torch.cuda.stream(side):
                devb[k][0].copy_(host[k][0], non_blocking=True); devb[k][1].copy_(host[k][1], non_blocking=True)
                ev[k].record(side)
        cur.wait_event(ev[k])
        tr.replay(devb[k][0], samples(devb[k][1]))
        done[k].record(cur)
        n = k ^ 1
        with torch.cuda.stream(side):
            side.wait_event(done[n]) if i > 0 else None
            devb[n][0].copy_(host[n][0], non_blocking=True); devb[n][1].copy_(host[n][1], non_blocking=True)
            ev[n].record(side)
    t_ovl = timed(overlapped, args.steps)
    # ---- uint8 labels on the wire
    host8 = [(h[0], h[1].to(torch.uint8).pin_memory()) for h in host]
    dev8 = [(d[0], torch.empty_like(h[1], device=dev)) for d, h in zip(devb, host8)]

    def serial8(i):
        k = i & 1
        dev8[k][0].copy_(host8[k][0], non_blocking=True)
        dev8[k][1].copy_(host8[k][1], non_blocking=True)
        tr.replay(dev8[k][0], samples(dev8[k][1]))
    t_ser8 = timed(serial8, args.steps)

    def overlapped8(i):
        k = i & 1
        cur = torch.cuda.current_stream(dev)
        if i == 0:
            with torch.cuda.stream(side):
                dev8[k][0].copy_(host8[k][0], non_blocking=True); dev8[k][1].copy_(host8[k][1], non_blocking=True)
                ev[k].record(side)
        cur.wait_event(ev[k])
        tr.replay(dev8[k][0], samples(dev8[k][1]))
        done[k].record(cur)
        n = k ^ 1
        with torch.cuda.stream(side):
            side.wait_event(done[n]) if i > 0 else None
            dev8[n][0].copy_(host8[n][0], non_blocking=True); dev8[n][1].copy_(host8[n][1], non_blocking=True)
            ev[n].record(side)
    torch.cuda.synchronize()
    t_ovl8 = timed(overlapped8, args.steps)
    mb8 = (host8[0][0].numel() + host8[0][1].numel()) / 1e6
    mb = (host[0][0].numel() + host[0][1].numel() * 8) / 1e6
    print(f'resident inputs      : {t_res * 1e3:7.3f} ms/step  {B / t_res:8.1f} img/s')
    print(f'PCIe, serial copy    : {t_ser * 1e3:7.3f} ms/step  {B / t_ser:8.1f} img/s   ({mb:.0f} MB per step from pinned host memory)')
    print(f'PCIe, overlapped copy: {t_ovl * 1e3:7.3f} ms/step  {B / t_ovl:8.1f} img/s')
    print(f'uint8 labels, serial : {t_ser8 * 1e3:7.3f} ms/step  {B / t_ser8:8.1f} img/s   ({mb8:.0f} MB per step)')
    print(f'uint8 labels, overlap: {t_ovl8 * 1e3:7.3f} ms/step  {B / t_ovl8:8.1f} img/s')


if __name__ == '__main__':
    main()
