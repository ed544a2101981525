"""Make this package answer to the module names the reference's main.py imports.

    import importlib; importlib.import_module('kgc-gcn_amd').dropin.install()
    import main        # the reference's own main.py: `from model import MGCN`, `from data_loader import DataLoader`

After install(), `model`, `data_loader` and `utils` resolve to this package's modules.
"""
import sys


def install():
    from . import data_loader, model, utils
    sys.modules['model'] = model
    sys.modules['data_loader'] = data_loader
    sys.modules['utils'] = utils
    return model, data_loader, utils
