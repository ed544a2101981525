"""Small helpers with the names the reference's loops import from its utils.py (utils.py:10-155).
Only `get_param` matters to the hot path (parameter init order = seeding parity, utils.py:113-118);
the rest is host plumbing kept so the reference's main.py can run against this package unchanged."""
import json
import logging
import os
import shutil

import torch
import torch.nn as nn


def get_param(shape):
    p = nn.Parameter(torch.empty(*shape))
    nn.init.xavier_uniform_(p.data)
    return p


class Params(object):
    """Hyper-parameters as attributes, read from / written to a json file (utils.py:10-42)."""

    def __init__(self, json_path):
        self.update(json_path)

    def update(self, json_path):
        with open(json_path) as handle:
            vars(self).update(json.load(handle))

    def save(self, json_path):
        save_json(vars(self), json_path)

    @property
    def dict(self):
        return vars(self)


class RunningAverage(object):
    """Mean of the values fed to update() (utils.py:45-66)."""

    def __init__(self):
        self._sum = 0.0
        self._count = 0

    def update(self, val):
        self._sum, self._count = self._sum + val, self._count + 1

    def __call__(self):
        return self._sum / float(self._count)

    @property
    def steps(self):
        return self._count

    @property
    def total(self):
        return self._sum


def save_json(obj, json_file):
    with open(json_file, 'w') as handle:
        json.dump(obj, handle, indent=4)


def set_logger(log_path):
    """Root logger at INFO with one file and one console handler, installed once (utils.py:69-95)."""
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    if root.handlers:
        return
    layout = logging.Formatter('%(asctime)s [%(levelname)s] %(message)s')
    for handler in (logging.FileHandler(log_path), logging.StreamHandler()):
        handler.setFormatter(layout)
        root.addHandler(handler)


def save_checkpoint(state, is_best, checkpoint_dir):
    """utils.py:121-139. `measure` comes from np.round(...) in evaluate (main.py:92,158-162): stored as a Python float
    so that the file holds tensors and plain numbers only."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, 'last.ckpt')
    if isinstance(state, dict) and 'measure' in state and state['measure'] is not None:
        state = dict(state, measure=float(state['measure']))
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(checkpoint_dir, 'best.ckpt'))


def _numpy_scalar_globals():
    """What a reference-written checkpoint pickles besides tensors: its `measure` is an np.float64 (main.py:158-162).
    Allow-listing exactly these keeps torch.load on the no-code (weights_only) path."""
    import numpy as np
    core = getattr(np, '_core', None) or np.core
    scalar = core.multiarray.scalar
    # the constructor is pickled under the module path of the numpy that WROTE the file: numpy.core (1.x) or numpy._core (2.x)
    return [(scalar, 'numpy.core.multiarray.scalar'), (scalar, 'numpy._core.multiarray.scalar'), np.dtype,
            type(np.dtype('float64'))]


def load_checkpoint(checkpoint, model, optimizer=None):
    if not os.path.exists(checkpoint):
        raise FileNotFoundError("File doesn't exist {}".format(checkpoint))   # the reference raises a str (Q11)
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        ckpt = torch.load(checkpoint, map_location='cpu', weights_only=True)
    model.load_state_dict(ckpt['state_dict'])
    if optimizer is not None:
        # 'optim_dict' is in reference edge order: the model lays it out with its tables and keeps the two tied
        loader = getattr(model, 'load_optimizer_state_dict', None)
        if loader is not None:
            loader(optimizer, ckpt['optim_dict'])
        else:
            optimizer.load_state_dict(ckpt['optim_dict'])
    measure = ckpt.get('measure', None)
    return float(measure) if measure is not None else None
