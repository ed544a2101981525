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
    """Attribute bag backed by a json file."""

    def __init__(self, json_path):
        self.update(json_path)

    def update(self, json_path):
        with open(json_path) as f:
            self.__dict__.update(json.load(f))

    def save(self, json_path):
        with open(json_path, 'w') as f:
            json.dump(self.__dict__, f, indent=4)

    @property
    def dict(self):
        return self.__dict__


class RunningAverage(object):
    def __init__(self):
        self.total, self.steps = 0.0, 0

    def update(self, val):
        self.total += val
        self.steps += 1

    def __call__(self):
        return self.total / float(self.steps)


def save_json(obj, json_file):
    with open(json_file, 'w') as f:
        json.dump(obj, f, indent=4)


def set_logger(log_path):
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    if not logger.handlers:
        fmt = logging.Formatter('%(asctime)s [%(levelname)s] %(message)s')
        for h in (logging.FileHandler(log_path), logging.StreamHandler()):
            h.setFormatter(fmt)
            logger.addHandler(h)


def save_checkpoint(state, is_best, checkpoint_dir):
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, 'last.ckpt')
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(checkpoint_dir, 'best.ckpt'))


def load_checkpoint(checkpoint, model, optimizer=None):
    if not os.path.exists(checkpoint):
        raise FileNotFoundError("File doesn't exist {}".format(checkpoint))   # the reference raises a str (Q11)
    ckpt = torch.load(checkpoint, map_location='cpu', weights_only=True)
    model.load_state_dict(ckpt['state_dict'])
    if optimizer is not None:
        optimizer.load_state_dict(ckpt['optim_dict'])
    return ckpt.get('measure', None)
