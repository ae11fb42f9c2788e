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


def _cmp_cfg():
    import led_net_amd as L
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']:
        # every valid pixel selected (p < thres always): the per-rank OHEM means then average to the mean over
        # the concatenated batch (equal pixel counts per rank), so DDP's gradient mean IS the big-batch gradient
        c['thres'], c['min_kept'] = 1.5, 1
    return cfg


def _cmp_batch():
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (4, 3, 320, 320), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (4, 1, 320, 320), dtype=torch.int64, generator=g)
    return img, lab


def _two_steps(tr, img, samples):
    """step 1 with learning rate 0 (gradients through autograd's AccumulateGrad; parameters stay identical, the
    momentum buffer takes the gradient), step 2 with the real rate: gradient sinks, grouped SyncBN collectives
    and the exchange started from inside the backward -- from IDENTICAL parameters, so the comparison is not
    swamped by the divergence of an earlier step."""
    lr = tr.base_lr
    tr.base_lr = 0.0
    tr.train_step(img, samples)
    tr.base_lr = lr
    tr.train_step(img, samples)


def _cmp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LEDN_EXPERIMENTAL'] = '1'
    os.environ['LEDN_MULTI_COMM'] = '1'      # the opt-in form: exchange of the non-stem gradients from inside the backward
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import led_net_amd as L
    from conftest import bind_emu
    cfg = _cmp_cfg()
    img, lab = _cmp_batch()
    per = img.shape[0] // world
    img, lab = img[rank * per:(rank + 1) * per], lab[rank * per:(rank + 1) * per]
    with bind_emu():
        model = L.MODELS.build(cfg['model'])
        model.load_state_dict(torch.load(os.path.join(out_dir, 'init.pt')))
        tr = L.Trainer(model, cfg, world_size=world)
        _two_steps(tr, img, [L.SegDataSample(gt=lab[i]) for i in range(per)])
        assert tr._early_done, 'the non-stem gradients were not exchanged from inside the backward'
    if rank == 0:
        torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, 'world2.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_ddp_world2_equals_world1_on_concatenated_batch(tmp_path):
    """SyncBN + DDP semantics, end to end: two ranks with two images each must make the same SGD step as ONE
    process on the four images (BatchNorm statistics over the global batch; BatchNorm gamma/beta gradients =
    LOCAL sums averaged by the exchange -- not the all-reduced sums, which would train every BatchNorm affine
    parameter at world x the learning rate; the non-stem gradients exchanged from inside the backward).

    Tolerance: this random-init network on noise images is chaotic in f32 (ReLU / SEAM-percentile / 4-sample
    BatchNorm flips): the SAME single-process step with the batch merely permuted differs by median 0.8 %,
    worst 1.5 % of the update norm (measured).  A wrong 1/world or a doubled SyncBN sum is a factor 2."""
    sys.path.insert(0, ROOT)
    import led_net_amd as L
    from conftest import bind_emu
    cfg = _cmp_cfg()
    torch.manual_seed(100)
    img, lab = _cmp_batch()
    with bind_emu():
        model = L.MODELS.build(cfg['model'])
        init = {k: v.detach().clone() for k, v in model.state_dict().items()}
        torch.save(init, tmp_path / 'init.pt')
        port = 31500 + (os.getpid() % 2000)
        ctx = mp.spawn(_cmp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=False)
        tr = L.Trainer(model, cfg, world_size=1)       # meanwhile: the single-process reference
        _two_steps(tr, img, [L.SegDataSample(gt=lab[i]) for i in range(4)])
    while not ctx.join():
        pass
    one = model.state_dict()
    two = torch.load(tmp_path / 'world2.pt')
    rels, ratios = [], {'bn': [], 'other': []}
    for k, v in one.items():
        if not v.is_floating_point():
            continue
        if 'running_mean' in k:
            assert (two[k] - v).norm().item() <= 2e-2 * v.norm().item() + 1e-5, k     # global-batch statistics
            continue
        upd = (v - init[k]).norm().item()
        if 'running_' in k or upd < 1e-6:     # (a bias in front of a BatchNorm has zero gradient: noise only)
            continue
        rels.append(((two[k] - v).norm().item() / upd, k))
        bn = v.dim() == 1 and (k.endswith('.weight') or k.endswith('.bias')) and ('.bn.' in k or 'norm' in k or '_att.' in k
                                                                                   or '.context' in k or '.proj.1.' in k)
        ratios['bn' if bn else 'other'].append((two[k] - init[k]).norm().item() / upd)
    rels.sort(reverse=True)
    for kind, r in ratios.items():
        r.sort()
        assert len(r) > 50 and abs(r[len(r) // 2] - 1.0) < 2e-2, (kind, len(r), r[len(r) // 2])
        assert 0.8 < r[0] and r[-1] < 1.25, (kind, r[0], r[-1])
    assert rels[len(rels) // 2][0] < 3e-2 and rels[0][0] < 0.25, (rels[len(rels) // 2], rels[:5])


def _order_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.pop('LEDN_MULTI_COMM', None)
    os.environ.pop('LEDN_EXPERIMENTAL', None)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import led_net_amd as L
    from led_net_amd import train as TR
    from conftest import bind_emu
    torch.manual_seed(100 + rank)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = 3000
    logs = []
    with bind_emu():
        model = L.MODELS.build(cfg['model'])
        tr = L.Trainer(model, cfg, world_size=world)
        assert tr._single_stream and not tr.overlap_exchange          # the default for N > 1
        g = torch.Generator().manual_seed(7 + rank)
        img = torch.randint(0, 256, (1, 3, 320, 320), dtype=torch.uint8, generator=g)
        lab = torch.randint(0, 2, (1, 1, 320, 320), dtype=torch.int64, generator=g)
        for _ in range(2):        # step 1: gradients through autograd; step 2: gradient sinks + table launches
            TR._Collective.log = []
            tr.train_step(img, [L.SegDataSample(gt=lab[0])])
            logs.append(TR._Collective.log)
            TR._Collective.log = None
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.save({'logs': logs, 'sd': sd}, os.path.join(out_dir, f'order{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_collective_order_identical_on_four_ranks(tmp_path):
    """The deadlock condition of the data-parallel step, checked without hardware: every rank must issue the SAME
    ordered list of (communicator slot, operation, element count) -- per slot and overall -- in every step; a rank
    that issues one all-reduce more, fewer, or in another order than its peers hangs the job.  World 4 over gloo
    (one image per rank), the default N > 1 path (one launch stream); also: four identical replicas afterwards."""
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_order_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    res = [torch.load(tmp_path / f'order{r}.pt') for r in range(4)]
    ref = res[0]['logs']
    assert len(ref[0]) > 100, len(ref[0])                       # ~170 SyncBN all-reduces + the gradient buckets
    for r in range(1, 4):
        for step in range(2):
            assert res[r]['logs'][step] == ref[step], (r, step)
        for k in res[0]['sd']:
            assert torch.equal(res[r]['sd'][k], res[0]['sd'][k]), (r, k)
    # same sequence with and without the gradient sinks (step 1 vs step 2): what a captured replay re-issues
    assert ref[0] == ref[1]
    slots = {c[0] for c in ref[0]}
    assert slots == {0, 'grad'}, slots                           # one stream: SyncBN on slot 0, then the exchange
    first_grad = next(i for i, c in enumerate(ref[0]) if c[0] == 'grad')
    assert all(c[0] == 'grad' for c in ref[0][first_grad:])      # the exchange comes after the last SyncBN collective
    nbytes = 4 * sum(c[2] for c in ref[0] if c[0] == 'grad' and c[1] == 'all_reduce')
    assert 5.5e6 < nbytes < 7e6, nbytes                          # the flat f32 gradient buffer, once


@pytest.mark.gpu
@pytest.mark.parametrize('multi', ['0', '1'])
def test_rccl_in_graph_single_rank(multi, monkeypatch):
    """The N > 1 step keeps its collectives inside the hipGraph by issuing ncclAllReduce on the launch
    stream (led_net_amd/rccl.py).  One GPU can check everything but the wire: a one-rank communicator,
    SyncBN statistics and the flat gradient all-reduced through it (identity), the whole step captured
    and replayed -- same parameters as the plain single-GPU trainer after three steps."""
    import copy
    sys.path.insert(0, ROOT)
    monkeypatch.setenv('LEDN_EXPERIMENTAL', '1')
    monkeypatch.setenv('LEDN_MULTI_COMM', multi)     # '0': the default N > 1 path (one launch stream); '1': opt-in
    import led_net_amd as L
    assert torch.cuda.is_available()
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = 5000
    base = L.MODELS.build(cfg['model'])
    g = torch.Generator().manual_seed(3)
    # (4 images: MFAF's global-pool BatchNorm then sees 4 values per channel; with 2 it is xhat = +-1 and its
    #  gradient is pure rounding noise that the rest of this chaotic random-init step amplifies)
    img = torch.randint(0, 256, (4, 3, 320, 320), dtype=torch.uint8, generator=g).to(dev)
    lab = torch.randint(0, 2, (4, 1, 320, 320), dtype=torch.int64, generator=g).to(dev)
    samples = [L.SegDataSample(gt=lab[i]) for i in range(4)]
    init = {k: v.detach().float().clone() for k, v in base.state_dict().items()}
    res = []
    for mode in (None, 'rccl'):
        model = copy.deepcopy(base).to(dev)
        tr = L.Trainer(model, cfg, max_iters=100, collectives=mode)
        assert (tr.comm is not None) == (mode == 'rccl')
        if mode == 'rccl':
            assert tr.comm.nranks == 1 and set(tr.comm.nranks_all.values()) == {1}
            assert set(tr.comm.comms) == ({0, 'grad'} if multi == '0' else {0, 1, 'grad'})
            assert tr._single_stream == (multi == '0')
        out = tr.train_step(img, samples)           # one eager step from identical weights
        torch.cuda.synchronize()
        res.append(({k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()},
                    {k: float(v.reshape(-1)[0]) for k, v in out.items()}))
        if mode == 'rccl':                           # ... then the same step captured with its collectives
            tr.capture(img, samples, warmup=1)
            rep = tr.replay()
            torch.cuda.synchronize()
            assert tr._graph is not None
            assert all(torch.isfinite(v.float()).all() for v in rep.values())
            # (the communicator is left to process exit: the captured graph still references it)
    (sd0, o0), (sd1, o1) = res
    for k in o0:      # same forward from the same weights: summation-order noise only
        assert abs(o1[k] - o0[k]) <= 2e-3 * abs(o0[k]) + 1e-4, (k, o0[k], o1[k])
    rels = []
    for k, v in sd0.items():
        if v.is_floating_point() and 'running_' not in k and 'num_batches' not in k:
            upd = (v - init[k]).norm().item()
            if upd > 1e-5:
                rels.append((sd1[k] - v).norm().item() / upd)
    rels.sort()
    # statistical bound as in test_whole_train_step_vs_oracle: a dropped or doubled all-reduce
    # (SyncBN sums entering the gradient buffer twice) is rel ~ 1 on the affected parameters
    # (run-to-run spread of this chaotic random-init step on the GPU: median up to ~0.1)
    assert rels[len(rels) // 2] < 0.15 and rels[int(len(rels) * 0.9)] < 0.4, (rels[len(rels) // 2], rels[int(len(rels) * 0.9)], rels[-1])
