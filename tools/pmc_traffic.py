#!/usr/bin/env python3
"""HBM traffic per kernel launch from the rocprofv3 PMC counters (run on the GPU box).

  python tools/pmc_traffic.py --mode train --dtype bf16 [--out profiles/pmc_traffic_train_bf16.json]

Two separate passes of the bench command (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2:
they do not fit one pass -- MI355X_MICROARCH.md "rocprofv3 PMC slots"), each with
--kernel-trace only.  Corrections of the same guide ("HBM [CDNA4]"):
  * both counters are in KiB;
  * on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) streaming reads at
    64 bytes: it is doubled here (every hot kernel reads 16 B per lane);
  * WRITE_SIZE is exact for 16 B/lane stores and float atomics.
Kernel names are reduced to the function name (template arguments and the ledn:: namespace
dropped), the way bench.py names them.  A third pass collects SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (N_SIMD x kernel cycles), a fraction in [0, 1] of the matrix-core
  issue capacity: the counter advances 32 cycles per v_mfma_f32_32x32x16_bf16 on the SIMD that issues it and is
  summed over the chip's 256 CUs x 4 SIMDs (MI355X_MICROARCH.md, cycle-constants table); kernel cycles =
  GRBM_GUI_ACTIVE / 8 XCDs when that yields a plausible shader clock (1.0-2.6 GHz against the dispatch's
  timestamps), else duration x 2.4 GHz.  (Round 1 divided by SQ_BUSY_CYCLES, which is counted per shader
  engine, and reported "fractions" of 4.8.)

This process never touches the GPU itself: it starts rocprofv3 (with the program directly after
`--`) as a child and parses the CSV it leaves.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, N_XCD, NOMINAL_GHZ = 256 * 4, 8, 2.4      # MI355X: 256 CUs x 4 SIMDs in 8 XCDs


def base_name(k):
    k = re.sub(r'\[clone.*$', '', k).strip()
    k = re.sub(r'^void\s+', '', k)
    depth, out = 0, []
    for ch in k:                      # drop <...> template arguments and the (...) parameter list
        if ch in '<(':
            depth += 1
        elif ch in '>)':
            depth -= 1
        elif depth == 0:
            out.append(ch)
    return ''.join(out).strip().split('::')[-1]


def run_pass(counters, outdir, bench_args):
    os.makedirs(outdir, exist_ok=True)
    cmd = ['rocprofv3', '--pmc', *counters, '--kernel-trace', '--output-format', 'csv', '-d', outdir, '--',
           sys.executable, os.path.join(ROOT, 'bench.py'), *bench_args]
    print('[pmc] ' + ' '.join(cmd), flush=True)
    env = dict(os.environ, TMPDIR='/tmp')
    with open(os.path.join(outdir, 'run.log'), 'w') as log:
        rc = subprocess.run(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, cwd='/tmp').returncode
    print(f'[pmc] rc={rc}', flush=True)
    files = glob.glob(os.path.join(outdir, '**', '*counter_collection.csv'), recursive=True)
    if rc != 0 or not files:
        raise SystemExit(f'PMC pass {counters} failed (rc={rc}); see {outdir}/run.log')
    acc = {}
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                k = base_name(row['Kernel_Name'])
                a = acc.setdefault(k, {})
                c = a.setdefault(row['Counter_Name'], [0.0, set()])
                c[0] += float(row['Counter_Value'])
                did = row.get('Dispatch_Id') or row.get('Correlation_Id')
                c[1].add(did)
                if row.get('Start_Timestamp') and row.get('End_Timestamp'):
                    d = a.setdefault('__ns__', [0.0, set()])
                    if did not in d[1]:
                        d[0] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
                        d[1].add(did)
    return {k: {c: (v[0], len(v[1])) for c, v in a.items()} for k, a in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='train')
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--out', default=None)
    ap.add_argument('--scratch', default=os.path.join(ROOT, 'gpurun_out', 'pmc'))
    ap.add_argument('--no-mfma-pass', action='store_true')
    args = ap.parse_args()
    bench_args = ['--mode', args.mode, '--dtype', args.dtype, '--steps', '1', '--warmup', '1',
                  '--no-cpu-baseline', '--no-graph']
    tag = f'{args.mode}_{args.dtype}'
    fetch = run_pass(['FETCH_SIZE'], os.path.join(args.scratch, tag, 'fetch'), bench_args)
    write = run_pass(['WRITE_SIZE'], os.path.join(args.scratch, tag, 'write'), bench_args)
    busy = {}
    if not args.no_mfma_pass:
        try:
            busy = run_pass(['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'],
                            os.path.join(args.scratch, tag, 'mfma'), bench_args)
        except SystemExit as e:      # optional evidence; the traffic passes are the required ones
            print(f'[pmc] MFMA-busy pass skipped: {e}', flush=True)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fkib, fn = fetch.get(k, {}).get('FETCH_SIZE', (0.0, 0))
        wkib, wn = write.get(k, {}).get('WRITE_SIZE', (0.0, 0))
        n = max(fn, wn, 1)
        rd = 2.0 * fkib * 1024.0 / max(fn, 1)
        wr = wkib * 1024.0 / max(wn, 1)
        rec = dict(launches=n, fetch_size_kib_raw_per_launch=round(fkib / max(fn, 1), 2),
                   write_size_kib_per_launch=round(wkib / max(wn, 1), 2),
                   hbm_read_bytes_per_launch=int(rd), hbm_write_bytes_per_launch=int(wr),
                   hbm_bytes_per_launch=int(rd + wr))
        b = busy.get(k)
        if b and 'SQ_VALU_MFMA_BUSY_CYCLES' in b:
            mf = b['SQ_VALU_MFMA_BUSY_CYCLES'][0]
            rec['sq_valu_mfma_busy_cycles_per_launch'] = round(mf / max(1, b['SQ_VALU_MFMA_BUSY_CYCLES'][1]), 1)
            ns = b.get('__ns__', (0.0, 0))[0]
            gui = b.get('GRBM_GUI_ACTIVE', (0.0, 0))[0]
            cycles, how = None, None
            if gui > 0 and ns > 0 and 1.0 <= gui / N_XCD / ns <= 2.6:
                cycles, how = gui / N_XCD, 'GRBM_GUI_ACTIVE / 8 XCDs'
            elif ns > 0:
                cycles, how = ns * NOMINAL_GHZ, f'dispatch duration x {NOMINAL_GHZ} GHz'
            elif gui > 0:
                cycles, how = gui / N_XCD, 'GRBM_GUI_ACTIVE / 8 XCDs (no timestamps)'
            if cycles:
                rec['kernel_cycles_per_launch'] = round(cycles / max(1, b['SQ_VALU_MFMA_BUSY_CYCLES'][1]), 1)
                rec['mfma_busy_frac'] = round(min(1.0, mf / (N_SIMD * cycles)), 5)
                rec['mfma_busy_cycles_from'] = how
        kernels[k] = rec
    out = dict(command='bench.py ' + ' '.join(bench_args),
               corrections='FETCH_SIZE (KiB) x1024 x2 [gfx950 wide-read undercount]; WRITE_SIZE (KiB) x1024',
               kernels=kernels)
    path = args.out or os.path.join(ROOT, 'gpurun_out', f'pmc_traffic_{tag}.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    top = sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]
    for k, r in top:
        print(f"{k:40s} x{r['launches']:4d}  rd {r['hbm_read_bytes_per_launch'] / 1e6:9.2f} MB  "
              f"wr {r['hbm_write_bytes_per_launch'] / 1e6:9.2f} MB  "
              f"mfma busy {r.get('mfma_busy_frac', float('nan')):.4f}", flush=True)
    print(f'[pmc] wrote {path}', flush=True)


if __name__ == '__main__':
    main()
