"""Dataset reader, query datasets and the graph feeder — the reference's data_loader.py surface
(DataLoader, KBDataset, get_data_loaders) so main.py's loops are drop-in callers.

Host-side, integer work. What differs from the reference: the graph is our own `Graph` (attribute
names unchanged) and carries the device CSR the HIP kernels walk (built by libmgcn_hip's host feeder
on first use). Semantics kept on purpose, quirks included:
  * ids in first-seen order over train/valid/test, lower-cased on build, raw on lookup (data_loader.py:64-86);
  * bi-directional edge list, edge k<E = s->o type r id k, edge E+k = o->s type r+R (data_loader.py:143-149);
  * label smoothing (1-eps)*y + 1/N for training rows only (data_loader.py:41-43);
  * five shuffling loaders; train triples carry tail -1 (data_loader.py:100-102,180-192).
"""
import logging
import os

import numpy as np
import torch
from torch.utils import data

from .graph import Graph


class KBDataset(data.Dataset):
    def __init__(self, triplets, num_entity, params, training=False):
        self.triplets = triplets
        self.num_entity = num_entity
        self.params = params
        self.training = training

    def __len__(self):
        return len(self.triplets)

    def get_label(self, label):
        y = np.zeros(self.num_entity, dtype=np.float32)
        y[np.asarray(label, dtype=np.int64)] = 1.0
        return torch.from_numpy(y)

    def __getitem__(self, idx):
        q = self.triplets[idx]
        y = self.get_label(q['label'])
        eps = self.params.lbl_smooth
        if self.training is True and eps != 0.0:
            y = (1.0 - eps) * y + (1.0 / self.num_entity)
        return torch.tensor(q['triple'], dtype=torch.long), y

    def collate_fn(self, batch):
        return torch.stack([b[0] for b in batch], dim=0), torch.stack([b[1] for b in batch], dim=0)


class DataLoader(object):
    def __init__(self, dataset, params):
        self.data_dir = os.path.join('data', dataset)
        # params.ingest_only (extension, default False): keep only what scales to 10^8 triples — ids, the graph, the
        # device filter index and the evaluation query tensor, all from the native reader; no per-query dict lists
        # (`triplets`, get_data_loaders), which are what main.py's loaders need and what costs the Python time.
        self._ingest_only = bool(getattr(params, 'ingest_only', False))
        self.graph = self._load_data()

    # -- reading ---------------------------------------------------------------------------------
    def _read(self, split):
        with open(os.path.join(self.data_dir, split + '.txt'), 'r') as f:
            return [line.strip().split() for line in f]

    def _read_ids_python(self):
        """data_loader.py:61-86 literally: two passes with dict lookups (ids built lower-cased, looked up raw)."""
        raw = {split: self._read(split) for split in ('train', 'valid', 'test')}
        self.entity2id, self.relation2id = {}, {}
        for split in ('train', 'valid', 'test'):
            for h, r, t in raw[split]:
                self.entity2id.setdefault(h.lower(), len(self.entity2id))
                self.relation2id.setdefault(r.lower(), len(self.relation2id))
                self.entity2id.setdefault(t.lower(), len(self.entity2id))
        return {split: [(self.entity2id[h], self.relation2id[r], self.entity2id[t]) for h, r, t in raw[split]]
                for split in ('train', 'valid', 'test')}

    def _read_ids_native(self, as_lists=True):
        """The same ids from the native reader (mgcn_ingest_*): one pass in C++, same insertion order,
        same errors. Returns None for files it declines (non-ASCII names)."""
        from . import _native
        paths = [os.path.join(self.data_dir, split + '.txt') for split in ('train', 'valid', 'test')]
        try:
            ent, rel, ids = _native.ingest(*paths)
        except _native.IngestUnsupported:
            return None
        self.entity2id = {name: i for i, name in enumerate(ent)}
        self.relation2id = {name: i for i, name in enumerate(rel)}
        self._id_triples = ids                                   # [n, 3] int64 per split, for filter_index()
        if not as_lists:
            return ids
        return {split: [tuple(row) for row in t.tolist()] for split, t in ids.items()}

    def _load_data(self):
        self._id_triples = None
        ids = None
        if os.environ.get('MGCN_NATIVE_INGEST', '1') != '0':
            ids = self._read_ids_native(as_lists=not self._ingest_only)
        if ids is None:
            ids = self._read_ids_python()
        nrel = len(self.relation2id)
        self.relation2id.update({name + '_reverse': idx + nrel for name, idx in list(self.relation2id.items())})
        self.num_entity, self.num_relation = len(self.entity2id), nrel
        if self._ingest_only:
            if self._id_triples is None:
                self._id_triples = {k: torch.tensor(v, dtype=torch.int64).reshape(-1, 3) for k, v in ids.items()}
            self.triplets, self._known_all = None, None
            self.num_edge = int(self._id_triples['train'].size(0))
            return self._build_graph(np.arange(self.num_entity, dtype=np.int64), self._id_triples['train'].numpy(),
                                     bi_direction=True)

        known, known_train = {}, None
        for split in ('train', 'valid', 'test'):
            for s, p, o in ids[split]:
                known.setdefault((s, p), set()).add(o)
                known.setdefault((o, p + nrel), set()).add(s)
            if split == 'train':
                known_train = {k: list(v) for k, v in known.items()}
        known_all = {k: list(v) for k, v in known.items()}
        self._known_all = known_all
        self.num_edge = len(ids['train'])

        self.triplets = {'train': [{'triple': (s, p, -1), 'label': objs, 'sub_samp': 1}
                                   for (s, p), objs in known_train.items()]}
        for split in ('valid', 'test'):
            tails, heads = [], []
            for s, p, o in ids[split]:
                tails.append({'triple': (s, p, o), 'label': known_all[(s, p)]})
                heads.append({'triple': (o, p + nrel, s), 'label': known_all[(o, p + nrel)]})
            self.triplets[split + '_tail'], self.triplets[split + '_head'] = tails, heads

        graph = self._build_graph(np.arange(self.num_entity, dtype=np.int64),
                                  np.array(ids['train'], dtype=np.int64).reshape(-1, 3), bi_direction=True)
        logging.info('entity={}, relation={}, train_triplets={}, valid_triplets={}, test_triplets={}'.format(
            self.num_entity, self.num_relation, len(ids['train']), len(ids['valid']), len(ids['test'])))
        return graph

    # -- feeder ----------------------------------------------------------------------------------
    def _edge_normal(self, edge_type, edge_index, num_entity):
        """1 / in-degree(dst) with inf -> 0. Stored on the graph and never read by the layer (Q1)."""
        dst = torch.from_numpy(edge_index[1]).long()
        deg = torch.bincount(dst, minlength=num_entity).to(torch.float32)
        norm = 1.0 / deg[dst]
        norm[torch.isinf(norm)] = 0
        return norm

    def _build_graph(self, graph_nodes, triplets, bi_direction=True):
        src, rel, dst = triplets[:, 0], triplets[:, 1], triplets[:, 2]
        if bi_direction is True:
            src, dst = np.concatenate((src, dst)), np.concatenate((dst, src))
            rel = np.concatenate((rel, rel + self.num_relation))
        edge_index = np.stack((src, dst))
        edge_attr = np.stack((rel, np.arange(edge_index.shape[1], dtype=np.int64)))
        graph = Graph(edge_index=torch.from_numpy(edge_index), edge_attr=torch.from_numpy(edge_attr))
        graph.entity = torch.from_numpy(graph_nodes)
        graph.num_nodes = len(graph_nodes)
        graph.edge_norm = self._edge_normal(rel, edge_index, len(graph_nodes))
        return graph

    def filter_index(self):
        """Device-side form of the evaluation filter (every known tail of every (subject, relation), i.e. what the
        dense label rows of the *_head / *_tail datasets mark): see dist.FilterIndex."""
        from .dist import FilterIndex
        if self._id_triples is not None:                         # sort-based build in C++ (mgcn_filter_index_build)
            from . import _native
            triples = torch.cat([self._id_triples[s] for s in ('train', 'valid', 'test')], dim=0)
            keys, ptr, tails = _native.filter_index_build(triples, self.num_relation)
            return FilterIndex(keys, ptr, tails, 2 * self.num_relation)
        return FilterIndex.from_known(self._known_all, 2 * self.num_relation)

    def train_index(self):
        """Known tails of every (subject, relation) over the TRAIN split, both directions — what the train dataset's
        label rows mark (data_loader.py:80-83,100-102) — as a dist.FilterIndex for mgcn_label_rows."""
        from .dist import FilterIndex
        if self._id_triples is not None:
            from . import _native
            keys, ptr, tails = _native.filter_index_build(self._id_triples['train'], self.num_relation)
            return FilterIndex(keys, ptr, tails, 2 * self.num_relation)
        known = {q['triple'][:2]: q['label'] for q in self.triplets['train']}
        return FilterIndex.from_known(known, 2 * self.num_relation)

    def train_queries(self):
        """[Q, 2] int64 (subject, relation id): the distinct training queries, in the train dataset's order."""
        if self.triplets is not None:
            return torch.tensor([q['triple'][:2] for q in self.triplets['train']], dtype=torch.int64).reshape(-1, 2)
        t = self._id_triples['train']
        both = torch.cat([t[:, :2], torch.stack([t[:, 2], t[:, 1] + self.num_relation], dim=1)], dim=0)
        return torch.unique(both, dim=0)

    def eval_queries(self, split):
        """[Q, 3] int64 (subject, relation id, object): the split's tail queries followed by its head queries."""
        if self._id_triples is not None:
            t = self._id_triples[split]
            heads = torch.stack([t[:, 2], t[:, 1] + self.num_relation, t[:, 0]], dim=1)
            return torch.cat([t, heads], dim=0).contiguous()
        rows = [q['triple'] for q in self.triplets[split + '_tail']] + [q['triple'] for q in self.triplets[split + '_head']]
        return torch.tensor(rows, dtype=torch.int64).reshape(-1, 3)

    # -- query loaders ---------------------------------------------------------------------------
    def _get_dataset(self, data_type, params):
        if self.triplets is None:
            raise ValueError('this DataLoader was built with ingest_only: no per-query lists (use eval_queries / filter_index)')
        if data_type == 'train':
            return KBDataset(self.triplets['train'], self.num_entity, params, training=True)
        if data_type in ('valid_head', 'valid_tail', 'test_head', 'test_tail'):
            return KBDataset(self.triplets[data_type], self.num_entity, params)
        raise ValueError('Unkown data type')

    def _create_data_loader(self, dataset, batch_size, num_workers, shuffle, drop_last=False):
        return data.DataLoader(dataset, batch_size=batch_size, num_workers=max(0, num_workers), shuffle=shuffle,
                               collate_fn=dataset.collate_fn, drop_last=drop_last)

    def get_data_loaders(self, batch_size, num_workers, params):
        return {mark: self._create_data_loader(self._get_dataset(mark, params), batch_size, num_workers, shuffle=True)
                for mark in ('train', 'valid_head', 'valid_tail', 'test_head', 'test_tail')}
