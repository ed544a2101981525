import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_pkg():
    """The product package directory is named `kgc-gcn_amd` (not a Python identifier)."""
    return importlib.import_module('kgc-gcn_amd')


@pytest.fixture(scope='session')
def pkg():
    return load_pkg()


@pytest.fixture(scope='session')
def oracle():
    return importlib.import_module('oracle.mgcn_oracle')


class Golden(object):
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'))
        import json
        self.hp = json.loads(str(self.z['params_json']))

    def __getitem__(self, k):
        return self.z[k]

    def t(self, k):
        return torch.from_numpy(np.ascontiguousarray(self.z[k]))

    def has(self, k):
        return k in self.z.files

    def state_dict(self):
        return {k[3:]: self.t(k) for k in self.z.files if k.startswith('sd_')}

    def grads(self):
        return {k[5:]: self.t(k) for k in self.z.files if k.startswith('grad_')}

    @property
    def data_dir(self):
        ds = 'Toy' if self.name.startswith('toy') else self.name
        return os.path.join(GOLDEN, 'data', ds)


_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


FULL_CASES = ['toy_small', 'syn_a', 'syn_b']
ENCODER_CASES = ['toy_d100', 'syn_c']
ALL_CASES = FULL_CASES + ENCODER_CASES
