import os, sys, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import led_net_amd as L
variant = sys.argv[1]
dev = torch.device('cuda:0')
torch.manual_seed(5)
cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
for c in cfg['model']['decode_head']['loss_decode']:
    c['min_kept'] = 5000
size = 1024 if 'big' in variant else 320
n = 4 if 'big' in variant else 2
g = torch.Generator().manual_seed(3)
img = torch.randint(0, 256, (n, 3, size, size), dtype=torch.uint8, generator=g).to(dev)
lab = torch.randint(0, 2, (n, 1, size, size), dtype=torch.int64, generator=g).to(dev)
samples = [L.SegDataSample(gt=lab[i]) for i in range(n)]
if variant == 'exact':
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import faulthandler; faulthandler.enable()
    import test_distributed as t
    t.test_rccl_in_graph_single_rank()
    print('exact OK', flush=True)
    sys.exit(0)
if 'plainfirst' in variant:
    m0 = L.MODELS.build(cfg['model']).to(dev)
    t0 = L.Trainer(m0, cfg, max_iters=100)
    t0.train_step(img, samples)
    if 'gc' in variant:
        del t0, m0
        import gc; gc.collect()
model = L.MODELS.build(cfg['model']).to(dev)
if 'bf16' in variant:
    model.set_act_dtype(torch.bfloat16)
tr = L.Trainer(model, cfg, max_iters=100, collectives='rccl')
if 'prestep' in variant:
    tr.train_step(img, samples)
torch.cuda.synchronize()
tr.capture(img, samples, warmup=3 if 'w3' in variant else 1)
out = tr.replay()
torch.cuda.synchronize()
print(variant, 'OK', {k: float(v.reshape(-1)[0]) for k, v in out.items()}, flush=True)
