#!/usr/bin/env python3
"""LED-Net hot-path benchmark (driver contract: see the round prompt).

  python bench.py --gpus N --steps K --warmup W [--mode train|infer] [--dtype bf16|f32]

One "step" = one pass of the hot path over one synthetic batch already resident
in HBM: mode=train -> forward + OHEM-CE loss + backward + SGD on 16 x 3 x 1024 x 1024
per GPU (BASELINE.json configs[2]/[3]); mode=infer -> forward + multi-scale logit
fusion + argmax on 8 x 3 x 1024 x 1024 per GPU (configs[1]).  N>1 is launched by
torch.distributed.run, one rank per GPU (RCCL); images are sharded across ranks
(weak scaling), value = images of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     dominant kernel's achieved rate vs the gfx950 peak, timed with
                  HIP events on the launch stream inside the timed region;
  "cpu_baseline": the CPU oracle (oracle/spec.py, stock PyTorch fp32 ops) timed on
                  this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}   # dense peaks, same guide


def synthetic_batch(n, h, w, dev, seed=304):
    """uint8 BGR images, labels in {0,1} with a 16-px ignore border."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (n, 3, h, w), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (n, 1, h, w), dtype=torch.int64, generator=g)
    lab[:, :, :16, :] = 255
    lab[:, :, -16:, :] = 255
    lab[:, :, :, :16] = 255
    lab[:, :, :, -16:] = 255
    return img.to(dev), lab.to(dev)


def build_model(dev, dtype, train, backbone_flags=None):
    import led_net_amd as L
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    cfg['model']['backbone'].update(backbone_flags or {})
    model = L.MODELS.build(cfg['model'])
    model.set_act_dtype(torch.bfloat16 if dtype == 'bf16' else torch.float32)
    model.to(dev)
    model.train(train)
    return model, cfg


def pmc_record(kernel, mode, dtype):
    """the committed PMC record of `kernel` (profiles/pmc_traffic_<mode>_<dtype>.json) or {}"""
    path = os.path.join(ROOT, 'profiles', f'pmc_traffic_{mode}_{dtype}.json')
    try:
        with open(path) as fh:
            return json.load(fh)['kernels'].get(kernel) or {}
    except (OSError, ValueError, KeyError):
        return {}


def pmc_traffic(kernel, mode, dtype):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of this same command
    (tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 corrections of
    MI355X_MICROARCH.md applied), committed as profiles/pmc_traffic_<mode>_<dtype>.json; None when
    no such collection exists."""
    path = os.path.join(ROOT, 'profiles', f'pmc_traffic_{mode}_{dtype}.json')
    try:
        with open(path) as fh:
            rec = json.load(fh)['kernels'].get(kernel)
    except (OSError, ValueError, KeyError):
        return None
    return None if rec is None else rec['hbm_bytes_per_launch']


def host_cores():
    """Threads the CPU leg may use: the cgroup CPU share / affinity of this process, never
    os.cpu_count() (the GPU box reports 256 logical CPUs for a 16-CPU share; 256 torch threads on
    16 CPUs ran the oracle 50x slower).  LEDN_CPU_THREADS overrides."""
    if os.environ.get('LEDN_CPU_THREADS'):
        return max(1, int(os.environ['LEDN_CPU_THREADS']))
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            pr = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // pr))
        except (OSError, ValueError):
            pass
    return min(n, 16)    # one GPU's share of the box's host cores


def cpu_baseline(mode, h, w, budget_s=20.0):
    """Time the CPU oracle on a bounded sample (batch 1-2 of the bench image size, a few iterations)."""
    from oracle import spec
    import led_net_amd as L
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    model = L.MODELS.build(cfg['model'])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    bs = 2 if mode == 'train' else 1   # train-mode BN over the global-pool branch needs > 1 value per channel
    if mode == 'train':
        for k, v in sd.items():
            if v.is_floating_point() and 'running_' not in k:
                v.requires_grad_(True)

    def step(x, lab):
        if mode == 'train':
            out = spec.loss(x, lab, sd)
            (out['decode.loss_context'] + out['decode.loss_spatial']).backward()
            with torch.no_grad():
                for v in sd.values():
                    if v.grad is not None:
                        v -= 0.01 * v.grad
                        v.grad = None
        else:
            with torch.no_grad():
                spec.predict(x, sd)
    img, lab = synthetic_batch(bs, 512, 512, 'cpu')
    step(spec.preprocess(img), lab)          # warm-up (thread pool, allocator) on a small image
    img, lab = synthetic_batch(bs, h, w, 'cpu')
    x = spec.preprocess(img)
    print(f'[bench] cpu_baseline: oracle {mode} at {bs}x{h}x{w} on {cores} threads ...', file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    n = 0
    while True:
        step(x, lab)
        n += 1
        dt = time.perf_counter() - t0
        print(f'[bench] cpu_baseline: {n} iterations, {dt:.1f} s', file=sys.stderr, flush=True)
        if dt > budget_s or n >= 20:
            break
    return {'value': round(bs * n / dt, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} iterations of oracle.spec.{"loss+backward+SGD" if mode == "train" else "predict"} '
                      f'at batch {bs}, {h}x{w}, fp32, torch CPU ops, {cores} threads'}


def bf16_parity(h, w, dev):
    """The north_star's "argmax masks bit-exact" holds for f32 activations (tests/test_parity_argmax.py); for the
    benched dtype the flip rate is MEASURED here, at the bench image size: one synthetic h x w image cut into its four
    (h/2) x (w/2) quadrants (the oracle needs ~3 s per quadrant on the host cores), the product's bf16 predict pass
    against the f32 CPU oracle on the same quadrants.  Checker side only: the oracle never computes a benched value."""
    from oracle import spec
    import led_net_amd as L
    torch.manual_seed(304)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    model = L.MODELS.build(cfg['model']).eval()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():            # non-trivial running statistics / affine parameters, as a trained network has
        for n, b in model.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif n.endswith('running_var'):
                b.copy_(0.6 + 0.8 * torch.rand(b.shape, generator=g))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.set_act_dtype(torch.bfloat16)
    model.to(dev)
    img, _ = synthetic_batch(1, h, w, 'cpu')
    hh, hw = h // 2, w // 2
    crops = torch.cat([img[:, :, i:i + hh, j:j + hw] for i in (0, hh) for j in (0, hw)]).contiguous()
    with torch.no_grad():
        want, want_mask = spec.predict(spec.preprocess(crops), sd)
        out = model(crops.to(dev), mode='predict')
    logits = torch.stack([o.seg_logits.data for o in out]).float().cpu()
    mask = torch.cat([o.pred_sem_seg.data for o in out]).long().cpu()
    err = (logits - want).abs()
    scale = want.abs().max().item()
    margin = (want[:, 0] - want[:, 1]).abs()
    flips = mask != want_mask
    return {'what': f'bf16 predict vs f32 CPU oracle, argmax over the four {hh}x{hw} quadrants of one {h}x{w} image',
            'pixels': int(mask.numel()), 'flips': int(flips.sum()), 'bf16_flip_frac': float(flips.float().mean()),
            'flip_frac_margin_gt_1pct_of_scale': float((flips & (margin > 0.01 * scale)).float().mean()),
            'flip_frac_margin_gt_5pct_of_scale': float((flips & (margin > 0.05 * scale)).float().mean()),
            'max_logit_err_over_scale': round(err.max().item() / scale, 5),
            'mean_logit_err_over_scale': round(err.mean().item() / scale, 6),
            'f32_flip_frac': 0.0, 'f32_source': 'tests/test_parity_argmax.py (asserted per run on the GPU)'}


def launch_ranks(n):
    """python bench.py --gpus N without a launcher: start `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 bench.py <same arguments>` as a child process, pass its output
    through and return its exit code.  Fails loudly when the node has fewer than N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f'[bench] --gpus {n} requested but this node exposes {have} GPU(s): refusing to run '
              f'(a 1-rank number must never be reported as an N-rank one)', file=sys.stderr)
        return 2
    port = os.environ.get('MASTER_PORT')
    if not port:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] launching', ' '.join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def _coll_launches(log):
    """launches the logged collectives amount to: an ncclGroup is one launch, every other all-reduce one"""
    n, depth = 0, 0
    for _slot, op, _numel in log:
        if op == 'group_begin':
            depth += 1
            n += 1
        elif op == 'group_end':
            depth -= 1
        elif depth == 0:
            n += 1
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None, help='ranks (one per GPU); default: WORLD_SIZE when launched by torch.distributed.run, else 1')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--mode', default=None, choices=['train', 'infer'])
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=1024)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-seconds', type=float, default=20.0, help='budget of the bounded CPU-oracle sample')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--collectives', default=None, choices=['auto', 'rccl', 'torch'],
                    help='N>1: rccl = ncclAllReduce on the launch stream (graph-capturable), torch = torch.distributed (eager); '
                         'with --gpus 1, rccl runs the one-rank self-test of that path (SyncBN + gradient all-reduce in the graph)')
    ap.add_argument('--local-bn', action='store_true',
                    help='N>1: per-rank BatchNorm statistics instead of the config\'s SyncBN (one all-reduce per BN and direction)')
    ap.add_argument('--backbone', default='', help="reconstruction flags of LEDNet as key=value,... (e.g. cespb_depth=(2,3),context_tail='pappm'); default: the survey's contract")
    ap.add_argument('--deterministic', action='store_true',
                    help='run the whole bench in deterministic mode (LEDN_OPT_DETERMINISTIC: fixed-order reductions, no f32 atomics)')
    ap.add_argument('--no-det-probe', action='store_true',
                    help='skip the deterministic-mode cost measurement after the timed region (profiling runs: keeps its kernels out of the rocprofv3 statistics)')
    ap.add_argument('--trace-only', action='store_true',
                    help='stop after the timed region (for rocprofv3 timeline traces: no instrumented eager pass, no JSON)')
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get('WORLD_SIZE', 1))

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # launched bare (python bench.py --gpus N): become the launcher.  The N ranks are CHILD processes of
        # torch.distributed.run (the shape of the reference's tools/dist_train.sh:9-18); nothing in this
        # process has touched the GPU yet (device_count() does not initialise it), and it never will.
        sys.exit(launch_ranks(args.gpus))
    # stdout carries exactly ONE line (the JSON, rank 0).  Libraries write there too (librccl prints a version banner
    # to stdout on its first communicator): for the whole run fd 1 points at stderr, the JSON goes to the saved fd.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit(f'[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                 f'torch.distributed.run --nproc-per-node {args.gpus} (or run bare: bench.py spawns the ranks itself)')
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('nccl')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    import led_net_amd as L
    from led_net_amd import ops
    if args.deterministic:
        L.set_deterministic(True)
    has_train = hasattr(L, 'Trainer')
    mode = args.mode or ('train' if has_train else 'infer')
    bs = args.batch or (16 if mode == 'train' else 8)
    H, W = args.height, args.width
    import ast
    flags = {}
    for kv in filter(None, __import__('re').split(r',(?![^(]*\))', args.backbone)):
        k, v = kv.split('=', 1)
        flags[k.strip()] = ast.literal_eval(v.strip())
    model, cfg = build_model(dev, args.dtype, mode == 'train', flags)
    if args.local_bn:
        model.backbone.sync_bn = model.decode_head.sync_bn = False
    img, lab = synthetic_batch(bs, H, W, dev, seed=304 + rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    class Watchdog:
        """N > 1 only: a rank that makes no progress (a collective some peer never joins blocks inside the HIP runtime
        and cannot be interrupted) ends the job loudly instead of hanging until the driver's limit."""

        def __init__(self, phase, seconds):
            import threading
            self.phase, self.seconds, self.done = phase, seconds, threading.Event()
            self.thread = threading.Thread(target=self._run, daemon=True)

        def _run(self):
            if not self.done.wait(self.seconds):
                print(f'[bench] rank {rank}: no progress for {self.seconds} s in phase "{self.phase}" '
                      f'(collectives={args.collectives or "default"}); aborting. --collectives torch selects the '
                      f'torch.distributed path (eager launches).', file=sys.stderr, flush=True)
                os._exit(86)

        def __enter__(self):
            if world > 1:
                self.thread.start()
            return self

        def __exit__(self, *exc):
            self.done.set()
            return False

    graphed = False
    if mode == 'train':
        trainer = L.Trainer(model, cfg, world_size=world, collectives=args.collectives)
        samples = [L.SegDataSample(gt=lab[i]) for i in range(bs)]

        def eager_step():
            return trainer.train_step(img, samples)
        step = eager_step
        if (world == 1 or trainer.comm is not None) and not args.no_graph:
            try:        # whole step (fwd + loss + bwd [+ in-stream RCCL all-reduces] + SGD) as one hipGraph
                with Watchdog('first steps + graph capture', 420):
                    trainer.capture(img, samples)
                step, graphed = (lambda: trainer.replay()), True
            except Exception as e:   # noqa: BLE001 -- report and fall back to eager launches
                print(f'[bench] graph capture failed, running eager: {e!r}', file=sys.stderr)
    else:
        def eager_step():
            with torch.no_grad():
                return model.decode_head.predict_with_mask(model.extract_feat(img))
        step = eager_step
        if not args.no_graph:
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(3):
                        eager_step()
                torch.cuda.current_stream(dev).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = eager_step()
                step, graphed = (lambda: graph.replay()), True
            except Exception as e:   # noqa: BLE001
                print(f'[bench] graph capture failed, running eager: {e!r}', file=sys.stderr)

    with Watchdog('warm-up + timed steps', 420):
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
    if args.trace_only:
        if rank == 0:
            print(f'[bench] trace-only: {bs * world * args.steps / dt:.1f} images/s', file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        return
    # per-kernel HIP-event timing needs individual launches and costs two event records per launch:
    # an instrumented eager pass of the same step right after the timed region (every rank runs it,
    # the collectives stay matched; rank 0 reports)
    k_steps = min(args.steps, 5)
    with Watchdog('instrumented eager pass', 420):
        eager_step()
        barrier()
        ops.start_timing()
        coll_log = None
        if mode == 'train':
            from led_net_amd import train as _TR
            coll_log = _TR._Collective.log = []
        try:
            for _ in range(k_steps):
                eager_step()
            barrier()
        finally:
            if mode == 'train':
                _TR._Collective.log = None
        launches = ops.stop_timing()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        # ---- dominant kernel FUNCTION (as rocprofv3 --stats groups them, template instances merged)
        # by summed HIP-event time; its roofline = the larger of its two time floors
        # (algorithmic bytes / HBM peak, algorithmic flops / dense MFMA peak)
        agg, fam = {}, {}
        for r in launches:
            k = (r['entry'], r['sig'])
            a = agg.setdefault(k, dict(ms=0.0, n=0, bytes=r['bytes'], flops=r['flops'], kernel=r['kernel']))
            a['ms'] += r['ms']
            a['n'] += 1
            f = fam.setdefault(r['kernel'], dict(ms=0.0, n=0, bytes=0, flops=0))
            f['ms'] += r['ms']
            f['n'] += 1
            f['bytes'] += r['bytes']
            f['flops'] += r['flops']
        top = sorted(agg.items(), key=lambda kv: -kv[1]['ms'])
        kname, f = max(fam.items(), key=lambda kv: kv[1]['ms'])
        avg_ms = f['ms'] / f['n']
        gbs = f['bytes'] / (f['ms'] * 1e-3) / 1e9
        tfl = f['flops'] / (f['ms'] * 1e-3) / 1e12
        peak_t = MFMA_PEAK_TFLOPS[args.dtype]
        mfma_bound = f['flops'] / (peak_t * 1e12) > f['bytes'] / (HBM_PEAK_GBS * 1e9)
        if mfma_bound:
            roof = dict(bound='mfma', achieved=round(tfl, 3), peak=peak_t, unit='TFLOP/s', frac=round(tfl / peak_t, 5))
        else:
            roof = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(gbs / HBM_PEAK_GBS, 5))
        roof['traffic'] = pmc_traffic(kname, mode, args.dtype)
        roof['mfma_busy'] = pmc_record(kname, mode, args.dtype).get('mfma_busy_frac')   # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), same PMC collection
        roof.update(kernel=kname, avg_us=round(avg_ms * 1e3, 2), launches_per_step=f['n'] // k_steps,
                    alg_bytes_per_launch=f['bytes'] // f['n'], alg_flops_per_launch=f['flops'] // f['n'],
                    alg_gbs=round(gbs, 1), alg_tflops=round(tfl, 3), mfma_frac=round(tfl / peak_t, 5),
                    arithmetic_intensity=round(f['flops'] / max(1, f['bytes']), 1),
                    share_of_gpu_time=round(f['ms'] / max(1e-9, sum(v['ms'] for v in fam.values())), 4))
        # the other kernel functions by time, each against the same two roofs (the convolution work is spread over five
        # matrix-core kernels since round 3: the dominant one alone no longer describes it)
        roof['by_kernel'] = []
        for kn, ff in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])[:8]:
            g_, t_ = ff['bytes'] / (ff['ms'] * 1e-3) / 1e9, ff['flops'] / (ff['ms'] * 1e-3) / 1e12
            roof['by_kernel'].append(dict(kernel=kn, ms_per_step=round(ff['ms'] / k_steps, 3), launches_per_step=ff['n'] // k_steps,
                                          avg_us=round(ff['ms'] / ff['n'] * 1e3, 2), alg_gbs=round(g_, 1), hbm_frac=round(g_ / HBM_PEAK_GBS, 4),
                                          alg_tflops=round(t_, 2), mfma_frac=round(t_ / peak_t, 4)))
        total_flops = sum(v['flops'] * v['n'] for k, v in agg.items() if k[0] in ('ledn_conv2d', 'ledn_conv2d_wgrad', 'ledn_window_attn'))
        total_bytes = sum(v['bytes'] * v['n'] for v in agg.values())
        gpu_ms = sum(v['ms'] for v in agg.values())
        out = {
            'metric': 'images/sec at 1024x1024, LED-Net fwd+bwd' if mode == 'train'
                      else 'images/sec at 1024x1024, LED-Net fwd (inference)',
            'value': round(bs * world * args.steps / dt, 3), 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'backbone_flags': flags or None, 'workload': (f'LED-Net {H}x{W} train_step (fwd+OHEM-CE+bwd+SGD) batch {bs}/GPU' if mode == 'train'
                                    else f'LED-Net {H}x{W} inference (fwd+fusion+argmax) batch {bs}/GPU'),
                       'global_batch': bs * world, 'parallelism': f'dp{world}',
                       'ranks': world, 'images_per_rank_per_step': bs,
                       'rccl_nranks': (trainer.comm.nranks if mode == 'train' and trainer.comm is not None
                                       else (dist.get_world_size() if world > 1 else None)),
                       'batchnorm': ('SyncBN (config): one RCCL all-reduce of [2,C] per BN (or group of BNs ready together) and direction' if mode == 'train' and not args.local_bn and trainer._all_reduce is not None
                                     else 'per-rank statistics'),
                       'collectives': (None if mode != 'train' or trainer._all_reduce is None else
                                       (('ncclAllReduce on the launch stream (in the graph)' if trainer._single_stream else
                                         'ncclAllReduce on the launch streams, one communicator per stream (in the graph); '
                                         'non-stem gradients exchanged during the stem backward') if trainer.comm is not None
                                        else 'torch.distributed (eager)')),
                       'rccl_nranks_all': (trainer.comm.nranks_all if mode == 'train' and trainer.comm is not None else None),
                       'collective_path': (None if mode != 'train' or trainer._all_reduce is None else
                                           ('one launch stream, one ordered collective sequence per rank (default for N > 1)'
                                            if trainer._single_stream else 'LEDN_EXPERIMENTAL=1 LEDN_MULTI_COMM=1: one communicator per branch stream')),
                       'collectives_per_step': (None if not coll_log else
                                                sum(1 for c in coll_log if c[1] == 'all_reduce') // k_steps),
                       'collective_launches_per_step': (None if not coll_log else _coll_launches(coll_log) // k_steps),
                       'collective_bytes_per_step': (None if not coll_log else
                                                     4 * sum(c[2] for c in coll_log if c[1] == 'all_reduce') // k_steps),
                       'kernel_launches_per_step': len(launches) // k_steps,
                       'submission': 'hipGraph replay' if graphed else 'eager launches',
                       'kernel_timing': 'HIP events on the launch streams, instrumented eager pass of the same step right after the timed region',
                       'conv_gemm_tflops_whole_step': round(total_flops / (gpu_ms * 1e-3) / 1e12, 3),
                       'hbm_alg_gbs_whole_step': round(total_bytes / (gpu_ms * 1e-3) / 1e9, 1),
                       'gpu_busy_frac': round(gpu_ms / k_steps / (dt / args.steps * 1e3), 3)},
            'roofline': roof,
        }
        out['config']['deterministic'] = bool(L.is_deterministic())
        if mode == 'train' and world == 1 and graphed and not L.is_deterministic() and not args.no_det_probe:
            # the price of deterministic mode on this box, measured after the timed region: the same step re-captured
            # with every reduction in a fixed order (partial rows + ordered summing launches instead of f32 atomics)
            try:
                L.set_deterministic(True)
                trainer.capture(img, samples, warmup=2)
                for _ in range(3):
                    trainer.replay()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(10):
                    trainer.replay()
                torch.cuda.synchronize(dev)
                out['deterministic_mode'] = {'ms_per_step': round((time.perf_counter() - t1) / 10 * 1e3, 3), 'steps': 10,
                                             'default_ms_per_step': out['ms_per_step']}
            except Exception as e:   # noqa: BLE001 -- a note, never the bench line
                out['deterministic_mode'] = {'error': repr(e)}
            finally:
                L.set_deterministic(False)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(mode, H, W, args.cpu_baseline_seconds)
            try:
                out['parity'] = bf16_parity(H, W, dev)
            except Exception as e:   # noqa: BLE001 -- the parity note must never cost the bench line
                out['parity'] = {'error': repr(e)}
        if os.environ.get('LEDN_BENCH_VERBOSE'):
            for e, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms']):
                print(f"{v['ms'] / k_steps:9.3f} ms/step  x{v['n'] // k_steps:4d}  {v['bytes'] / max(1e-9, v['ms']) / 1e6:8.1f} GB/s "
                      f"{v['flops'] / max(1e-9, v['ms']) / 1e9:8.2f} TF/s  [kernel] {e}", file=sys.stderr)
            verbose = os.environ['LEDN_BENCH_VERBOSE']
            n_sig = int(verbose) if verbose.isdigit() and int(verbose) > 1 else 25     # signatures listed
            for (e, sg), v in top[:n_sig]:
                print(f'{v["ms"] / k_steps:9.3f} ms/step  x{v["n"] // k_steps:3d}  {v["bytes"] * v["n"] / max(1e-9, v["ms"]) / 1e6:8.1f} GB/s '
                      f'{v["flops"] * v["n"] / max(1e-9, v["ms"]) / 1e9:8.2f} TF/s  {e} [{sg}] <{v["kernel"]}>', file=sys.stderr)
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
