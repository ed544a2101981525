"""ORACLE — test infrastructure, NOT product code.

A CPU restatement (torch-CPU / numpy, written from scratch) of the reference's M-GCN hot path in the
REFERENCE'S OWN ORDER OF OPERATIONS. Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; nothing under kgc-gcn_amd/ does, and the product path
raises when its HIP library is missing instead of falling back to anything here.

Pinning: every function below is checked in tests/test_oracle_golden.py against the vectors in
tests/golden/*.npz, which were produced by running the reference's unmodified model.py /
data_loader.py / main.py in the build container (tests/golden/gen/make_golden.py). The
gather / scatter-add primitive of the three absent, un-pinned third-party packages
(torch_geometric, torch_scatter, ordered_set; requirements.txt:1-5) is restated from its published
semantics — at that boundary the reference ships no tests, so that part is "parity unpinned".

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import os
from collections import OrderedDict, defaultdict

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# data_loader.py:61-120 — ids, triples, queries
# ------------------------------------------------------------------------------------------------
def load_dataset(data_dir):
    """data_loader.py:61-111. Ids in first-seen order over train, valid, test; tokens lower-cased when
    the id maps are built (67) but looked up raw (85-86) — an upper-case token raises KeyError (Q7)."""
    ent, rel = OrderedDict(), OrderedDict()
    for split in ('train', 'valid', 'test'):
        for line in open(os.path.join(data_dir, split + '.txt'), 'r'):
            s, r, o = [t.lower() for t in line.strip().split()]
            ent.setdefault(s, len(ent))
            rel.setdefault(r, len(rel))
            ent.setdefault(o, len(ent))
    entity2id = dict(ent)
    relation2id = dict(rel)
    nrel = len(relation2id)
    for name, idx in list(relation2id.items()):
        relation2id[name + '_reverse'] = idx + nrel

    data = defaultdict(list)
    sr2o = defaultdict(set)
    sr2o_train = None
    for split in ('train', 'valid', 'test'):
        for line in open(os.path.join(data_dir, split + '.txt'), 'r'):
            h, r, t = line.strip().split()
            s, p, o = entity2id[h], relation2id[r], entity2id[t]
            data[split].append((s, p, o))
            sr2o[(s, p)].add(o)
            sr2o[(o, p + nrel)].add(s)
        if split == 'train':
            sr2o_train = {k: sorted(v) for k, v in sr2o.items()}
    sr2o_all = {k: sorted(v) for k, v in sr2o.items()}

    queries = defaultdict(list)
    for (s, p), objs in sr2o_train.items():                       # data_loader.py:100-102
        queries['train'].append(((s, p, -1), objs))
    for split in ('valid', 'test'):                                # data_loader.py:104-110
        for s, p, o in data[split]:
            queries[split + '_tail'].append(((s, p, o), sr2o_all[(s, p)]))
            queries[split + '_head'].append(((o, p + nrel, s), sr2o_all[(o, p + nrel)]))
    return dict(entity2id=entity2id, relation2id=relation2id, num_entity=len(entity2id),
                num_relation=nrel, num_edge=len(data['train']), data=dict(data), queries=dict(queries))


def build_edge_list(train_triples, num_relation):
    """data_loader.py:132-149. Edge k<E: s_k -> o_k, type r_k, id k; edge E+k: o_k -> s_k, type r_k+R."""
    t = np.asarray(train_triples, dtype=np.int64).reshape(-1, 3)
    src, rel, dst = t[:, 0], t[:, 1], t[:, 2]
    s2 = np.concatenate((src, dst))
    d2 = np.concatenate((dst, src))
    r2 = np.concatenate((rel, rel + num_relation))
    edge_index = np.stack((s2, d2))
    edge_attr = np.stack((r2, np.arange(edge_index.shape[1], dtype=np.int64)))
    return edge_index, edge_attr


def edge_normal(edge_index, num_entity):
    """data_loader.py:122-130 (dead argument, Q1): 1 / in-degree(dst), inf -> 0."""
    dst = torch.from_numpy(edge_index[1])
    deg = torch.zeros(num_entity).index_add_(0, dst, torch.ones(dst.numel()))
    norm = 1.0 / deg[dst]
    norm[torch.isinf(norm)] = 0
    return norm


def label_row(label_idx, num_entity, training=False, lbl_smooth=0.0):
    """data_loader.py:34-51 (Q6: (1-eps)*y + 1/N)."""
    y = torch.zeros(num_entity, dtype=torch.float32)
    y[torch.as_tensor(list(label_idx), dtype=torch.long)] = 1.0
    if training and lbl_smooth != 0.0:
        y = (1.0 - lbl_smooth) * y + (1.0 / num_entity)
    return y


# ------------------------------------------------------------------------------------------------
# model.py:72-118 — the layer, reference order (weight multiply per edge)
# ------------------------------------------------------------------------------------------------
def compute_norm(edge_index_half, num_ent):
    """model.py:72-80. Degree counted by SOURCE (row) for both endpoints (Q2); deg^-1/2, inf -> 0."""
    row, col = edge_index_half[0], edge_index_half[1]
    deg = torch.zeros(num_ent, dtype=torch.float32).index_add_(0, row, torch.ones(row.numel()))
    deg_inv = deg.pow(-0.5)
    deg_inv[deg_inv == float('inf')] = 0
    return deg_inv[row] * deg_inv[col]


def _propagate(x, edge_index, edge_type, edge_norm, edge_embs, rels_embs, weight):
    """model.py:111-118 (message) + the absent MessagePassing.propagate (aggr='add', source->target):
    gather source rows, per-edge product, per-edge weight multiply, scale, scatter-add in edge order."""
    x_j = x.index_select(0, edge_index[0])
    rel_emb = rels_embs.index_select(0, edge_type)
    msg = torch.matmul(x_j * rel_emb * edge_embs, weight)
    if edge_norm is not None:
        msg = msg * edge_norm.view(-1, 1)
    out = torch.zeros(x.size(0), weight.size(1), dtype=x.dtype)
    return out.index_add_(0, edge_index[1], msg)


def layer_forward(sd, prefix, x, edge_index, edge_type, edge_embs, rels_embs, training=False,
                  drop_p=0.0, momentum=0.1, eps=1e-5):
    """model.py:82-109. `sd` maps state-dict names to tensors; `prefix` e.g. 'conv1.'.
    Returns (all_ent, all_rel). Dropout inside the layer (103) only with drop_p > 0 (RNG-dependent).
    In training mode the running statistics in `sd` are updated in place, as nn.BatchNorm1d does."""
    E = edge_type.size(0) // 2
    N = x.size(0)
    rels = torch.cat([rels_embs, sd[prefix + 'loop_rel']], dim=0)
    in_idx, out_idx = edge_index[:, :E], edge_index[:, E:]
    loop_idx = torch.stack([torch.arange(N), torch.arange(N)])
    loop_type = torch.full((N,), rels.size(0) - 1, dtype=torch.long)
    loop_embs = sd[prefix + 'loop_edge'].expand(N, -1)
    in_norm = compute_norm(in_idx, N)
    out_norm = compute_norm(out_idx, N)
    in_res = _propagate(x, in_idx, edge_type[:E], in_norm, edge_embs[:E], rels, sd[prefix + 'in_weight'])
    out_res = _propagate(x, out_idx, edge_type[E:], out_norm, edge_embs[E:], rels, sd[prefix + 'out_weight'])
    loop_res = _propagate(x, loop_idx, loop_type, None, loop_embs, rels, sd[prefix + 'loop_weight'])
    if training and drop_p > 0:
        in_res = F.dropout(in_res, drop_p, True)
        out_res = F.dropout(out_res, drop_p, True)
    out = (in_res + out_res + loop_res) / 3
    if sd.get(prefix + 'bias') is not None:
        out = out + sd[prefix + 'bias']
    rm, rv = sd[prefix + 'ent_bn.running_mean'], sd[prefix + 'ent_bn.running_var']
    out = F.batch_norm(out, rm, rv, sd[prefix + 'ent_bn.weight'], sd[prefix + 'ent_bn.bias'],
                       training, momentum, eps)
    all_ent = torch.tanh(out)
    all_rel = torch.matmul(rels, sd[prefix + 'rels_weight'])[:-1]
    return all_ent, all_rel


def aggregate_then_weight(sd, prefix, x, edge_index, edge_type, edge_embs, rels_embs):
    """The BUILD's order for the same layer: A_mode = sum_e norm_e * x_j*rel*ee, then one
    weight multiply per mode. Returns the pre-BN sum `out` and the three aggregates; used by tests to
    bound the rounding difference between the two orders."""
    E = edge_type.size(0) // 2
    N = x.size(0)
    rels = torch.cat([rels_embs, sd[prefix + 'loop_rel']], dim=0)
    aggs = []
    for lo, hi in ((0, E), (E, 2 * E)):
        idx = edge_index[:, lo:hi]
        nrm = compute_norm(idx, N)
        m = x.index_select(0, idx[0]) * rels.index_select(0, edge_type[lo:hi]) * edge_embs[lo:hi] * nrm.view(-1, 1)
        aggs.append(torch.zeros(N, x.size(1)).index_add_(0, idx[1], m))
    aggs.append(x * rels[-1] * sd[prefix + 'loop_edge'])
    out = (aggs[0] @ sd[prefix + 'in_weight'] + aggs[1] @ sd[prefix + 'out_weight'] + aggs[2] @ sd[prefix + 'loop_weight']) / 3
    return out, aggs


# ------------------------------------------------------------------------------------------------
# model.py:159-181 — ConvE decoder, and model.py:24-40 — the whole forward
# ------------------------------------------------------------------------------------------------
def conve_trunk(sd, hp, src_emb, rel_emb, training=False):
    """model.py:161-175 (dense trunk; dropouts only with p > 0 in training)."""
    O = hp['gcn_out_dim']
    stack = torch.cat([src_emb.view(-1, 1, O), rel_emb.view(-1, 1, O)], dim=1)
    stack = torch.transpose(stack, 2, 1).reshape(-1, 1, 2 * hp['k_w'], hp['k_h'])

    def bn(x, name):
        return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'], sd[name + '.weight'],
                            sd[name + '.bias'], training, 0.1, 1e-5)
    x = bn(stack, 'conv2.bn0')
    x = F.conv2d(x, sd['conv2.conv_e.weight'], sd.get('conv2.conv_e.bias'))
    x = F.relu(bn(x, 'conv2.bn1'))
    x = F.dropout(x, hp.get('feat_drop', 0.0), training)
    x = x.view(x.size(0), -1)
    x = F.linear(x, sd['conv2.fc.weight'], sd['conv2.fc.bias'])
    x = F.dropout(x, hp.get('hidden_drop', 0.0), training)
    x = F.relu(bn(x, 'conv2.bn2'))
    return x


def score_all(x, all_ent, bias):
    """model.py:177-179."""
    s = torch.mm(x, all_ent.transpose(1, 0))
    s = s + bias.expand_as(s)
    return torch.sigmoid(s)


def mgcn_forward(sd, hp, src, rel, edge_index, edge_attr, training=False):
    """model.py:24-40: identity gathers, layer, dropout, pick rows, trunk, score every entity."""
    edge_type, edge_ids = edge_attr[0], edge_attr[1]
    entity = torch.arange(sd['entity_embedding'].size(0))
    ent = torch.index_select(sd['entity_embedding'], 0, entity)
    ee = torch.index_select(sd['edge_embeddings'], 0, edge_ids)
    all_ent, all_rel = layer_forward(sd, 'conv1.', ent, edge_index, edge_type, ee, sd['relation_embedding'],
                                     training=training)
    all_ent = F.dropout(all_ent, hp.get('gcn_drop', 0.0), training)
    x = conve_trunk(sd, hp, all_ent.index_select(0, src), all_rel.index_select(0, rel), training)
    return score_all(x, all_ent, sd['conv2.bias'])


# ------------------------------------------------------------------------------------------------
# main.py:105-135, 80-102 — filtered ranking and metrics
# ------------------------------------------------------------------------------------------------
def filtered_rank(pred, label, obj):
    """main.py:122-126 literally (double argsort, unstable on ties — Q5), plus the decomposition the
    build is held to: gt = #(masked score > target), ties / ties_lower = #(== target) among the other
    entities (all / those with a lower index). On rows with ties == 0: rank == 1 + gt exactly."""
    pred = pred.clone()
    b = torch.arange(pred.size(0))
    target = pred[b, obj]
    pred = torch.where(label.to(torch.uint8).bool(), -torch.ones_like(pred) * 10000000, pred)
    pred[b, obj] = target
    ranks = 1 + torch.argsort(torch.argsort(pred, dim=1, descending=True), dim=1, descending=False)[b, obj]
    gt = (pred > target[:, None]).sum(1)
    eq = pred == target[:, None]
    eq[b, obj] = False
    idx = torch.arange(pred.size(1))[None, :]
    return dict(ranks=ranks, gt=gt, ties=eq.sum(1), ties_lower=(eq & (idx < obj[:, None])).sum(1), target=target)


def accumulate(results, ranks):
    """main.py:128-133."""
    ranks = ranks.float()
    results['count'] = torch.numel(ranks) + results.get('count', 0.0)
    results['mr'] = torch.sum(ranks).item() + results.get('mr', 0.0)
    results['mrr'] = torch.sum(1.0 / ranks).item() + results.get('mrr', 0.0)
    for k in range(10):
        results['hits@%d' % (k + 1)] = torch.numel(ranks[ranks <= (k + 1)]) + results.get('hits@%d' % (k + 1), 0.0)
    return results


def combine(tail, head, hits=(1, 3, 10)):
    """main.py:84-97."""
    count = float(tail['count'])
    res = {'mr': np.round((tail['mr'] + head['mr']) / (2 * count), 5),
           'mrr': np.round((tail['mrr'] + head['mrr']) / (2 * count), 5)}
    for k in hits:
        res['hits@%d' % k] = np.round((tail['hits@%d' % k] + head['hits@%d' % k]) / (2 * count), 5)
    return res


def evaluate(sd, hp, ds, edge_index, edge_attr, split, batch_size=128):
    """main.py:80-135 with the encoder recomputed for every batch (Q4) — this is also what the
    cpu_baseline leg of bench.py times. `ds` is load_dataset()'s dict."""
    N = ds['num_entity']
    out = {}
    for mode in ('tail', 'head'):
        res = {}
        qs = ds['queries']['%s_%s' % (split, mode)]
        for i in range(0, len(qs), batch_size):
            chunk = qs[i:i + batch_size]
            trip = torch.tensor([q[0] for q in chunk], dtype=torch.long)
            label = torch.stack([label_row(q[1], N) for q in chunk])
            pred = mgcn_forward(sd, hp, trip[:, 0], trip[:, 1], edge_index, edge_attr)
            accumulate(res, filtered_rank(pred, label, trip[:, 2])['ranks'])
        out[mode] = res
    return combine(out['tail'], out['head']), out


# ------------------------------------------------------------------------------------------------
# synthetic graphs of the BASELINE.json shapes — shared by tests and bench
# ------------------------------------------------------------------------------------------------
def synthetic_triples(num_entity, num_relation, num_edge, seed=0, zipf=0.0):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, num_entity, size=num_edge)
    r = rng.integers(0, num_relation, size=num_edge)
    if zipf > 0:
        p = 1.0 / np.arange(1, num_entity + 1) ** zipf
        o = rng.choice(num_entity, size=num_edge, p=p / p.sum())
        o = rng.permutation(num_entity)[o]
    else:
        o = rng.integers(0, num_entity, size=num_edge)
    return np.stack((s, r, o), axis=1).astype(np.int64)


def init_layer_state(prefix, d_in, d_out, gen, bias=False):
    """Parameters of one layer, xavier-uniform like utils.get_param (utils.py:113-118), plus non-trivial
    BN statistics so eval-mode BN is exercised."""
    def xavier(*shape):
        bound = float(np.sqrt(6.0 / (shape[0] + shape[1])))
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound
    sd = {prefix + k: xavier(d_in, d_out) for k in ('loop_weight', 'in_weight', 'out_weight', 'rels_weight')}
    sd[prefix + 'loop_rel'] = xavier(1, d_in)
    sd[prefix + 'loop_edge'] = xavier(1, d_in)
    sd[prefix + 'ent_bn.weight'] = torch.rand(d_out, generator=gen) + 0.5
    sd[prefix + 'ent_bn.bias'] = torch.randn(d_out, generator=gen) * 0.1
    sd[prefix + 'ent_bn.running_mean'] = torch.randn(d_out, generator=gen) * 0.05
    sd[prefix + 'ent_bn.running_var'] = torch.rand(d_out, generator=gen) * 0.5 + 0.05
    if bias:
        sd[prefix + 'bias'] = torch.randn(d_out, generator=gen) * 0.1
    return sd
