"""Data-parallel path with world_size 2 over gloo on CPU (kernels on the emulator):
rank-0 broadcast, SyncBN statistics all-reduce, flat-gradient all-reduce with the mean
folded into the SGD kernel.  The 8-GPU RCCL run is the driver's; this covers the logic."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import led_net_amd as L
    from conftest import bind_emu
    torch.manual_seed(100 + rank)            # different init per rank: the broadcast must fix it
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = 5000
    with bind_emu():
        model = L.MODELS.build(cfg['model'])          # norm_cfg = SyncBN in the config
        assert model.backbone.sync_bn and model.decode_head.sync_bn
        tr = L.Trainer(model, cfg, world_size=world)
        g = torch.Generator().manual_seed(7 + rank)   # each rank its own shard of the batch
        img = torch.randint(0, 256, (2, 3, 320, 320), dtype=torch.uint8, generator=g)
        lab = torch.randint(0, 2, (2, 1, 320, 320), dtype=torch.int64, generator=g)
        out = tr.train_step(img, [L.SegDataSample(gt=lab[i]) for i in range(2)])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.save({'sd': sd, 'loss': {k: v.detach().clone() for k, v in out.items()}, 'img': img},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_ddp_world2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    # (1) identical replicas after broadcast + all-reduced update (bitwise: same reduced bytes)
    for k in r0['sd']:
        assert torch.equal(r0['sd'][k], r1['sd'][k]), k
    # (2) each rank computed its own loss on its own shard
    assert all(torch.isfinite(v).all() for v in r0['loss'].values())
    assert float(r0['loss']['decode.loss_context']) != float(r1['loss']['decode.loss_context'])
    # (3) SyncBN: the stem BN's running mean is the statistic of the GLOBAL batch
    from oracle import spec
    x = spec.preprocess(torch.cat([r0['img'], r1['img']]))
    # rank 0's initial weights were broadcast; the stem conv weight changed by one SGD step only,
    # so recompute the batch mean with the UPDATED weight's pre-update value is not available:
    # check instead the invariant that does not depend on the weights: both ranks hold the same
    # running stats and they differ from the init (0 / 1).
    rm = r0['sd']['backbone.stem.0.bn.running_mean']
    assert rm.abs().max() > 0 and torch.equal(rm, r1['sd']['backbone.stem.0.bn.running_mean'])
    assert x.shape[0] == 4
