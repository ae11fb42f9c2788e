import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


class Fixture:
    """One tests/golden/*.npz file split into its sections (torch tensors)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.sd, self.ins, self.outs, self.gin, self.gp = {}, {}, {}, {}, {}
        sec = {'sd': self.sd, 'in': self.ins, 'out': self.outs, 'gin': self.gin, 'gp': self.gp}
        for k in z.files:
            if k == 'meta':
                self.meta = json.loads(str(z[k]))
                continue
            s, rest = k.split('/', 1)
            sec[s][rest] = torch.from_numpy(np.array(z[k]))
        self.name = name

    def sd_clone(self, requires_grad=False):
        out = {}
        for k, v in self.sd.items():
            v = v.clone()
            if requires_grad and v.is_floating_point() and 'running_' not in k:
                v.requires_grad_(True)
            out[k] = v
        return out


@pytest.fixture
def golden():
    return Fixture


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.startswith(prefix) and f.endswith('.npz'))


# --------------------------------------------------------------------------- #
# CPU emulation build of the kernel sources (tests only; see csrc/emu.h)
# --------------------------------------------------------------------------- #
EMU_SO = os.path.join(ROOT, 'tests', 'emu', 'libledn_emu.so')
_emu_lib = None


def _build_emu():
    import subprocess
    subprocess.run(['make', '-s', '-j8', '-C', os.path.join(ROOT, 'led-net_amd', 'csrc'), 'emu'],
                   check=True)


import contextlib


@contextlib.contextmanager
def bind_emu():
    global _emu_lib
    import led_net_amd  # noqa: F401
    from led_net_amd import _lib
    if _emu_lib is None:
        _build_emu()
        _emu_lib = _lib.Library(EMU_SO, is_hip=False)
    with _lib.use_library(_emu_lib):
        yield _emu_lib


@pytest.fixture
def emu():
    """Binds the CPU-emulated kernels for the duration of a test."""
    with bind_emu() as lib:
        yield lib


class Backend:
    def __init__(self, dev):
        self.dev = torch.device(dev)

    def __call__(self, t):
        return t.to(self.dev) if torch.is_tensor(t) else t


@pytest.fixture(params=['emu', pytest.param('hip', marks=pytest.mark.gpu)])
def be(request):
    """Backend under test: 'emu' = kernel sources on the CPU emulator (no GPU
    needed), 'hip' = libledn_hip.so on cuda:0 through the same C ABI."""
    if request.param == 'emu':
        with bind_emu():
            yield Backend('cpu')
    else:
        assert torch.cuda.is_available(), 'gpu-marked test needs a HIP device'
        yield Backend('cuda:0')


def slow_on_emu(dev):
    """whole-network tests that take 2-5 minutes each on the CPU emulator AND have a -m gpu twin (same test body on
    libledn_hip.so) run on the emulator only with LEDN_EMU_SLOW=1: the default CPU suite keeps one whole-step test per
    kind (f32 step vs oracle, world-2 = world-1, collective order, resume, bf16 fan-in chains) plus every block-level
    test of the same kernels."""
    import os
    import pytest as _pytest
    if getattr(dev, 'type', str(dev)) == 'cpu' and not int(os.environ.get('LEDN_EMU_SLOW', '0')):
        _pytest.skip('whole-network test on the emulator: set LEDN_EMU_SLOW=1 (the GPU twin always runs)')
