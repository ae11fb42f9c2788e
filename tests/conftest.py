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
