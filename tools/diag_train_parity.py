"""Per-parameter relative error of one train step (product vs CPU oracle).
usage: python tools/diag_train_parity.py [emu|hip]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import led_net_amd as L
from oracle import spec
from test_train import _randomize
mode = sys.argv[1] if len(sys.argv) > 1 else 'emu'
dev = torch.device('cuda:0' if mode == 'hip' else 'cpu')
def run():
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']: c['min_kept'] = 20000
    model = L.MODELS.build(cfg['model']); _randomize(model, 3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(dev)
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (2, 3, 320, 320), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (2, 1, 320, 320), dtype=torch.int64, generator=g)
    lab[:, :, :6, :] = 255; lab[:, :, :, -5:] = 255
    init = {k: v.clone() for k, v in sd.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k}
    want = spec.loss(spec.preprocess(img), lab, sd, loss_cfg=((0.9, 20000, 1.0), (0.9, 20000, 0.4)))
    (want['decode.loss_context'] + want['decode.loss_spatial']).backward()
    tr = L.Trainer(model, cfg, max_iters=80000, lr=1.0, momentum=0.0, weight_decay=0.0)   # update == -grad
    got = tr.train_step(img.to(dev), [L.SegDataSample(gt=lab[i].to(dev)) for i in range(2)])
    print({k: float(v.reshape(-1)[0]) for k, v in got.items()}, {k: float(v.reshape(-1)[0]) for k, v in want.items()})
    new = model.state_dict(); rows = []
    for k, v in leaves.items():
        if v.grad is None: continue
        gp = (init[k] - new[k].cpu())      # lr=1, no momentum/wd -> gradient
        rel = ((gp - v.grad).norm() / (v.grad.norm() + 1e-12)).item()
        rows.append((rel, k, v.grad.norm().item()))
    rows.sort(reverse=True)
    for r in rows[:25]: print(f'{r[0]:.3e}  |g|={r[2]:.3e}  {r[1]}')
    print('median rel', sorted(r[0] for r in rows)[len(rows) // 2])
if mode == 'emu':
    from conftest import bind_emu
    with bind_emu(): run()
else:
    run()
