#!/usr/bin/env python3
"""Which parameters differ between the gradient-sink path and autograd accumulation (GPU)?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import led_net_amd as L
dev = torch.device('cuda:0')
torch.manual_seed(304)
cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
for c in cfg['model']['decode_head']['loss_decode']:
    c['min_kept'] = 5000
model = L.MODELS.build(cfg['model'])
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.dim() > 1:
            p.copy_(torch.randn(p.shape, generator=g) * (1.0 / max(1, p[0].numel())) ** 0.5)
model.to(dev)
g = torch.Generator().manual_seed(12)
img = torch.randint(0, 256, (8, 3, 320, 320), dtype=torch.uint8, generator=g).to(dev)
lab = torch.randint(0, 2, (8, 1, 320, 320), dtype=torch.int64, generator=g)
lab[:, :, :4, :] = 255
lab = lab.to(dev)
samples = [L.SegDataSample(gt=lab[i]) for i in range(8)]
tr = L.Trainer(model, cfg, max_iters=100)
tr.train_step(img, samples)
state = {k: v.clone() for k, v in model.state_dict().items()}
sinks = tr._sink_map
names = {id(p): n for n, p in model.named_parameters()}
def grads(use):
    model.load_state_dict(state)
    tr.flat_grad.zero_()
    tr._sink_map = sinks if use else {}
    tr.forward_backward(img, samples)
    return tr.flat_grad.detach().cpu().clone()
for rep in range(4):
    a, b, c = grads(True), grads(False), grads(False)
    off = 0
    bad = []
    gmax = max(b[o:o + p.numel()].norm().item() for o, p in [(sum(q.numel() for q in tr.params[:i]), p) for i, p in enumerate(tr.params)])
    for p in tr.params:
        n = p.numel()
        if any(p is q for q in tr.live):
            den = max(b[off:off + n].norm().item(), 1e-4 * gmax) + 1e-12
            r = (a[off:off + n] - b[off:off + n]).norm().item() / den
            f = (c[off:off + n] - b[off:off + n]).norm().item() / den
            if r > 0.3 or f > 0.3:
                bad.append((names[id(p)], round(r, 3), round(f, 3), b[off:off + n].norm().item() / gmax))
        off += n
    print('rep', rep, 'outliers (name, sink-vs-ref, ref-vs-ref, |g|/gmax):')
    for x in bad[:14]:
        print('   ', x)
