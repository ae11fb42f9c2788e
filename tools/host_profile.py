#!/usr/bin/env python3
"""cProfile of the eager (no hipGraph) train step on the GPU box: where the host time goes
(the eager step is host-bound; N > 1 runs eager)."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import led_net_amd as L
dev = torch.device('cuda:0')
model, cfg = bench.build_model(dev, 'bf16', True)
img, lab = bench.synthetic_batch(16, 1024, 1024, dev)
tr = L.Trainer(model, cfg)
samples = [L.SegDataSample(gt=lab[i]) for i in range(16)]
for _ in range(3):
    tr.train_step(img, samples)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.train_step(img, samples)
pr.disable()
torch.cuda.synchronize()
print('ms/step (profiled)', (time.perf_counter() - t0) / 5 * 1e3)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
