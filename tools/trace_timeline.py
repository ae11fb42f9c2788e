#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py (graph replay): for the last
timed steps, wall time per step, time with 0 / 1 / >=2 kernels in flight, and the kernels that
own the most EXCLUSIVE time (nothing else running): what to optimise once streams overlap.
usage: python tools/trace_timeline.py <kernel_trace.csv> [launches_per_step]"""
import csv
import re
import sys
from collections import defaultdict


def base(k):
    k = re.sub(r'^void\s+', '', k)
    depth, out = 0, []
    for ch in k:
        if ch in '<(':
            depth += 1
        elif ch in '>)':
            depth -= 1
        elif depth == 0:
            out.append(ch)
    return ''.join(out).strip().split('::')[-1]


rows = []
with open(sys.argv[1], newline='') as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), base(r['Kernel_Name']),
                     (int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0), int(r.get('Grid_Size_Y') or 1),
                      int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or 0))))
rows.sort()
# steps are delimited by the single sgd_step_kernel launch of each step
ends = [i for i, r in enumerate(rows) if r[2] == 'sgd_kernel']
if len(ends) < 4:      # inference: one planar im2col (or layout kernel) per step
    # (round 3 moved the inference stem onto stem_conv_reg_kernel: without it in this list the r03 inference timeline came out empty)
    first = next((k for k in ('stem_conv_reg_kernel', 'im2col_stem_planar_kernel', 'nchw_to_nhwc_kernel') if any(r[2] == k for r in rows)), None)
    ends = [i - 1 for i, r in enumerate(rows) if r[2] == first]
assert len(ends) >= 4, 'run bench.py --trace-only (steps are delimited by sgd_kernel / the first kernel of a pass)'
lo, hi = ends[-4] + 1, ends[-1] + 1          # the last three steps (graph replays of the timed region)
seg = rows[lo:hi]
nsteps = 3
t0, t1 = seg[0][0], max(r[1] for r in seg)
ev = []
for s, e, k, _g in seg:
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
active = defaultdict(int)
n = 0
last = t0
idle = one = multi = 0
excl = defaultdict(float)
gap_after = defaultdict(float)
prev_end_kernel = None
for t, d, k in ev:
    dt = t - last
    if n == 0:
        idle += dt
        if prev_end_kernel:
            gap_after[prev_end_kernel] += dt
    elif n == 1:
        one += dt
        excl[[kk for kk, c in active.items() if c > 0][0]] += dt
    else:
        multi += dt
    last = t
    n += d
    active[k] += d
    if d < 0:
        prev_end_kernel = k
wall = (t1 - t0) / nsteps / 1e6
print(f'steps analysed: {nsteps}, kernels/step: {len(seg) / nsteps:.0f}, wall {wall:.3f} ms/step')
print(f'idle {idle / nsteps / 1e6:.3f} ms  one-kernel {one / nsteps / 1e6:.3f} ms  overlapped {multi / nsteps / 1e6:.3f} ms')
tot = defaultdict(float)
cnt = defaultdict(int)
for s, e, k, _g in seg:
    tot[k] += e - s
    cnt[k] += 1
print('--- exclusive time (ms/step), total duration, launches/step')
for k, v in sorted(excl.items(), key=lambda kv: -kv[1])[:60]:
    print(f'{v / nsteps / 1e6:8.3f}  {tot[k] / nsteps / 1e6:8.3f}  x{cnt[k] / nsteps:6.1f}  {k}')
print('--- idle gaps following a kernel (ms/step)')
for k, v in sorted(gap_after.items(), key=lambda kv: -kv[1])[:10]:
    print(f'{v / nsteps / 1e6:8.3f}  x{cnt[k] / nsteps:6.1f}  {k}')

# ---- per launch shape (kernel, grid): average duration and average start-to-next-start interval (the cost on a
# serial chain: duration + the gap before the next kernel starts)
shape_t, shape_i, shape_n = defaultdict(float), defaultdict(float), defaultdict(int)
for i, (s0, e0, k, g) in enumerate(seg):
    key = (k, g)
    shape_t[key] += e0 - s0
    shape_n[key] += 1
    if i + 1 < len(seg):
        shape_i[key] += max(0, seg[i + 1][0] - s0)
print('--- by launch shape: ms/step, launches/step, avg us, avg start-to-next-start us, kernel (grid x, grid y, wg)')
for key, v in sorted(shape_t.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 70]:
    n_ = shape_n[key]
    print(f'{v / nsteps / 1e6:8.3f}  x{n_ / nsteps:6.1f}  {v / n_ / 1e3:8.1f}  {shape_i[key] / n_ / 1e3:8.1f}  {key[0]} {key[1]}')

# ---- what follows the kernels with the largest trailing gaps (gap = next start - this end, >= 5 us)
follow = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for i, (s0, e0, k, g) in enumerate(seg[:-1]):
    gap = seg[i + 1][0] - e0
    if gap >= 5000:
        f = follow[(k, g)][seg[i + 1][2]]
        f[0] += 1
        f[1] += gap
print('--- trailing gaps >= 5 us: kernel (grid) -> next kernel: count/step, mean gap us')
for key, nxt in sorted(follow.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:12]:
    for nk, (c, tot_gap) in sorted(nxt.items(), key=lambda kv: -kv[1][1])[:3]:
        print(f'  {key[0]} {key[1]} -> {nk}: {c / nsteps:.1f}/step, {tot_gap / c / 1e3:.1f} us')
# the largest idle gaps of the LAST step with the kernels around them (what was the device waiting for?)
last_step = rows[ends[-2] + 1:ends[-1] + 1]
gaps = []
run_end = last_step[0][1]
for i in range(1, len(last_step)):
    s = last_step[i][0]
    if s > run_end:
        gaps.append((s - run_end, i))
    run_end = max(run_end, last_step[i][1])
print('--- largest idle gaps of the last step: gap us | two kernels before -> two kernels after (name grid dur us)')
for g, i in sorted(gaps, reverse=True)[:24]:
    def fmt(r):
        return f'{r[2]}{r[3][:2]} {(r[1] - r[0]) / 1e3:.1f}'
    before = ' ; '.join(fmt(r) for r in last_step[max(0, i - 2):i])
    after = ' ; '.join(fmt(r) for r in last_step[i:i + 2])
    print(f'{g / 1e3:7.1f} | {before}  ->  {after}')
