"""Environment switches of the package, in two classes.

SUPPORTED (documented in INTEGRATION.md, honoured always):
    LEDN_DETERMINISTIC   1 = fixed-order reductions, no f32 atomics (led_net_amd.set_deterministic)
    LEDN_COLLECTIVES     auto | rccl | torch: which all-reduce the N > 1 step uses (Trainer(collectives=))
    LEDN_CAPTURE_MODE    hipGraph capture error mode override (global | thread_local | relaxed)
    LEDN_CPU_THREADS     threads of bench.py's CPU-oracle baseline
    LEDN_BENCH_VERBOSE   bench.py prints its per-kernel table

EXPERIMENTAL (A/B measurement knobs and built-but-slower paths kept for the record: DESIGN.md / EXPERIMENTS.md say what
each measured): read ONLY when LEDN_EXPERIMENTAL=1 is set; otherwise the tuned default applies and the variable is
ignored with a one-time warning, so an old shell export cannot silently change the numerics or the speed of a run.
"""
import os
import sys

EXPERIMENTAL = bool(int(os.environ.get('LEDN_EXPERIMENTAL', '0')))
_warned = set()


def knob(name, default):
    """value of the experimental environment knob `name` (str) or `default` when unset / not in experimental mode"""
    v = os.environ.get(name)
    if v is None:
        return default
    if not (EXPERIMENTAL or bool(int(os.environ.get('LEDN_EXPERIMENTAL', '0')))):
        if name not in _warned:
            _warned.add(name)
            print(f'[led_net_amd] {name}={v} ignored: experimental knobs need LEDN_EXPERIMENTAL=1', file=sys.stderr)
        return default
    return v


def knob_int(name, default):
    return int(knob(name, default))
