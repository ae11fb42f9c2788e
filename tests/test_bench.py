"""bench.py: the driver contract (one JSON line on stdout, the keys the driver reads, `roofline` / `cpu_baseline`
objects) and the multi-rank launcher's refusals."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=e,
                          capture_output=True, text=True, timeout=timeout)


def test_bare_multi_gpu_launch_refuses_without_enough_gpus():
    """`python bench.py --gpus N` without WORLD_SIZE becomes the launcher (torch.distributed.run children, as
    tools/dist_train.sh:9-18); on a box with fewer than N GPUs it must fail loudly, never run N = 1 silently"""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip('box has 8 GPUs')
    r = _run(['--gpus', '8', '--steps', '1', '--warmup', '0'], timeout=120)
    assert r.returncode != 0
    assert r.stdout.strip() == '' and 'GPU' in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'},
             timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr and r.stdout.strip() == ''


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['train', 'infer'])
def test_bench_prints_one_json_line_with_the_contract_keys(mode):
    r = _run(['--steps', '3', '--warmup', '1', '--mode', mode, '--cpu-baseline-seconds', '2'], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:1000]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['value'] > 0 and d['ms_per_step'] > 0
    assert d['unit'] == 'images/s' and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and d['dtype'] == 'bf16' and d['data'] == 'synthetic'
    assert '1024x1024' in d['metric'] and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - d['config']['global_batch'] / (d['ms_per_step'] * 1e-3)) < 0.02 * d['value']
    rf = d['roofline']
    assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] in ('GB/s', 'TFLOP/s') and 0 < rf['frac'] < 1
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3 and 'traffic' in rf and rf['kernel']
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1 and cb['sample']
