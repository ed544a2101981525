"""kgc-gcn_amd — the MI355X (gfx950) implementation of M-GCN's hot path: relational neighbour
aggregation, the dense layer epilogue, full-graph scoring and filtered ranking, behind the reference's
model.py / data_loader.py surface. The directory name is not a Python identifier: import it with
importlib.import_module('kgc-gcn_amd')."""
from . import _native, data_loader, dist, dropin, graph, harness, model, utils  # noqa: F401
from .data_loader import DataLoader, KBDataset  # noqa: F401
from .graph import Graph, GraphCSR  # noqa: F401
from .model import MGCN, ConvE, MGCNConv  # noqa: F401
