#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference here.

Build-container only (needs /root/reference, which never travels to the GPU box). It imports the
reference's own, unmodified utils.py / model.py / data_loader.py / main.py from /root/reference.
Three third-party packages those files import are absent from this image and un-pinned in the
reference (requirements.txt:1-5): torch_geometric, torch_scatter, ordered_set. They are supplied by
the stand-ins in ./standins (our own restatement of their published semantics, see each file's
header). Consequently:
  * everything computed by the reference's OWN code (id assignment, edge list, norms, message,
    layer epilogue, ConvE, filtered ranking, metrics, autograd through all of it) is pinned by
    these vectors;
  * the gather / scatter-add primitive inside the absent packages is "parity unpinned" by the
    reference (it ships no tests); it is restated, not executed.

Nothing from /root/reference is copied into the repo: outputs are arrays (inputs + expected
outputs) and the Toy data files (fixture data the reference ships: data/Toy/*.txt).

Usage:  python tests/golden/gen/make_golden.py        (rewrites tests/golden/*.npz, data/*)
"""
import json
import os
import shutil
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
REF = '/root/reference'

sys.path.insert(0, os.path.join(HERE, 'standins'))
sys.path.insert(0, REF)
warnings.filterwarnings('ignore')

import utils as ref_utils            # noqa: E402  (reference, unmodified)
import model as ref_model            # noqa: E402
import data_loader as ref_dl         # noqa: E402
import main as ref_main              # noqa: E402


# ------------------------------------------------------------------------------------------------
# synthetic triple files (our own data; committed next to the vectors)
# ------------------------------------------------------------------------------------------------
def synth_triples(seed, n_ent, n_rel, n_train, n_eval, zipf=0.0, dup_frac=0.0, extra_rel=True):
    """Returns dict split -> list of (h, r, t) token triples. Entities that never occur in train,
    duplicate train triples, self loops and a relation that only occurs in valid/test are all
    produced on purpose (they exist in data/Toy or in real KGs)."""
    rng = np.random.default_rng(seed)

    def draw_nodes(k):
        if zipf > 0:
            p = 1.0 / np.arange(1, n_ent + 1) ** zipf
            p /= p.sum()
            return rng.choice(n_ent, size=k, p=p)
        return rng.integers(0, n_ent, size=k)

    def make(k, rel_hi):
        h = rng.integers(0, n_ent, size=k)
        t = draw_nodes(k)
        r = rng.integers(0, rel_hi, size=k)
        return [(int(a), int(b), int(c)) for a, b, c in zip(h, r, t)]

    train = make(n_train, n_rel)
    n_dup = int(dup_frac * n_train)
    for i in range(n_dup):
        train.append(train[int(rng.integers(0, n_train))])
    rel_hi = n_rel + 1 if extra_rel else n_rel
    valid = make(n_eval, rel_hi)
    test = make(n_eval, rel_hi)
    # a few eval triples repeat train (h, r) pairs so filtered ranking has something to filter
    for i in range(min(n_eval // 2, n_train)):
        h, r, _ = train[int(rng.integers(0, n_train))]
        valid[i] = (h, r, int(rng.integers(0, n_ent)))
        h, r, _ = train[int(rng.integers(0, n_train))]
        test[i] = (h, r, int(rng.integers(0, n_ent)))
    tok = lambda tr: [('n%d' % h, 'p%d' % r, 'n%d' % t) for h, r, t in tr]
    return {'train': tok(train), 'valid': tok(valid), 'test': tok(test)}


def write_dataset(root, name, splits):
    d = os.path.join(root, 'data', name)
    os.makedirs(d, exist_ok=True)
    for split, triples in splits.items():
        with open(os.path.join(d, split + '.txt'), 'w') as f:
            f.write('\n'.join('\t'.join(t) for t in triples))
    return d


# ------------------------------------------------------------------------------------------------
def make_params(**over):
    p = dict(dataset='x', seed=2020, batch_size=16, lbl_smooth=0.1, num_workers=0, bias=False,
             gcn_in_dim=100, gcn_out_dim=200, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3,
             k_w=10, k_h=20, num_filter=200, kernel_size=7, clip_grad=1.0)
    p.update(over)
    ns = types.SimpleNamespace(**p)
    ns.device = torch.device('cpu')
    return ns, p


def randomize_state(model, seed):
    """Make BN running stats / affine and the decoder bias non-trivial, deterministically."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, buf in model.named_buffers():
            if name.endswith('running_mean'):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.05)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand(buf.shape, generator=g) * 0.5 + 0.05)
        for name, p in model.named_parameters():
            if '.bn' in name or 'ent_bn' in name:
                if name.endswith('weight'):
                    p.copy_(torch.rand(p.shape, generator=g) + 0.5)
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif name == 'conv2.bias':
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)


def npify(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def loader_arrays(dl):
    g = dl.graph
    out = {
        'num_entity': dl.num_entity, 'num_relation': dl.num_relation, 'num_edge': dl.num_edge,
        'edge_index': g.edge_index, 'edge_attr': g.edge_attr, 'edge_norm': g.edge_norm,
        'entity': g.entity, 'num_nodes': g.num_nodes,
    }
    ents = sorted(dl.entity2id, key=dl.entity2id.get)
    rels = sorted(dl.relation2id, key=dl.relation2id.get)
    out['entity_names'] = np.array(ents)
    out['relation_names'] = np.array(rels)
    for split, items in dl.triplets.items():
        out['q_%s_triple' % split] = np.array([it['triple'] for it in items], dtype=np.int64).reshape(-1, 3)
        lab = [sorted(it['label']) for it in items]   # python set order is an implementation detail
        out['q_%s_label_ptr' % split] = np.cumsum([0] + [len(l) for l in lab]).astype(np.int64)
        out['q_%s_label_idx' % split] = np.array([e for l in lab for e in l], dtype=np.int64)
    return out


def rank_details(pred, label, obj):
    """Reference predict() body (main.py:122-126) step by step, plus the tie-free decomposition
    gt / ties that our rank definition is checked against."""
    b = torch.arange(pred.size(0))
    target = pred[b, obj]
    masked = torch.where(label.byte().bool(), -torch.ones_like(pred) * 10000000, pred)
    masked[b, obj] = target
    ranks = 1 + torch.argsort(torch.argsort(masked, dim=1, descending=True), dim=1, descending=False)[b, obj]
    gt = (masked > target[:, None]).sum(1)
    eq = (masked == target[:, None])
    eq[b, obj] = False
    ties = eq.sum(1)
    idx = torch.arange(pred.size(1))[None, :]
    ties_lower = (eq & (idx < obj[:, None])).sum(1)
    return dict(ranks=ranks, gt=gt, ties=ties, ties_lower=ties_lower, target=target)


def run_case(root, name, params_over, scale_tables=1.0, full_model=True, seed=2020, state_seed=7):
    os.chdir(root)                                   # data path is relative (data_loader.py:57)
    params, pdict = make_params(**params_over)
    dl = ref_dl.DataLoader(name, params)
    out = {}
    out.update({'dl_' + k: v for k, v in npify(loader_arrays(dl)).items()})

    torch.manual_seed(seed)
    model = ref_model.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
    for k, p in model.named_parameters():            # seeded-init parity (construction order + initialisers)
        out['init_sum_' + k] = p.detach().double().sum()
    randomize_state(model, state_seed)
    if scale_tables != 1.0:                           # bigger margins between scores
        with torch.no_grad():
            model.entity_embedding.mul_(scale_tables)
            model.edge_embeddings.mul_(scale_tables)
            model.relation_embedding.mul_(scale_tables)
    graph = dl.graph

    keep = None if full_model else ('entity_embedding', 'relation_embedding', 'edge_embeddings', 'conv1.')
    for k, v in model.state_dict().items():
        if keep is None or k.startswith(keep):
            out['sd_' + k] = v.clone()

    # ---- encoder, eval mode (model.py:24-34, 72-118) ---------------------------------------
    model.eval()
    with torch.no_grad():
        edge_type, edge_ids = graph.edge_attr
        E = dl.num_edge
        out['norm_in'] = model.conv1.compute_norm(graph.edge_index[:, :E], dl.num_entity)
        out['norm_out'] = model.conv1.compute_norm(graph.edge_index[:, E:], dl.num_entity)
        ent = torch.index_select(model.entity_embedding, 0, graph.entity)
        ee = torch.index_select(model.edge_embeddings, 0, edge_ids)
        all_ent, all_rel = model.conv1(ent, graph.edge_index, edge_type, graph.edge_norm, ee, model.relation_embedding)
        out['eval_all_ent'] = all_ent
        out['eval_all_rel'] = all_rel

    if full_model:
        iters = dl.get_data_loaders(pdict['batch_size'], 0, params)
        # ---- full forward + filtered ranking on every eval query, fixed order ----------------
        model.eval()
        with torch.no_grad():
            for split in ['valid_tail', 'valid_head', 'test_tail', 'test_head']:
                ds = iters[split].dataset
                items = [ds[i] for i in range(len(ds))]
                trip = torch.stack([it[0] for it in items])
                lab = torch.stack([it[1] for it in items])
                pred = model(trip[:, 0], trip[:, 1], graph)
                out['eval_%s_score' % split] = pred.clone()
                for k, v in rank_details(pred.clone(), lab, trip[:, 2]).items():
                    out['eval_%s_%s' % (split, k)] = v
            # ---- the reference's own evaluate() (main.py:80-102) -----------------------------
            torch.manual_seed(123)                    # loaders shuffle (data_loader.py:190)
            for split in ['valid', 'test']:
                res = ref_main.evaluate(model, iters, graph, params, split)
                for k, v in res.items():
                    out['evaluate_%s_%s' % (split, k)] = np.float64(v)
                tail = ref_main.predict(model, iters, graph, split, params.device, mode='tail_batch')
                for k, v in tail.items():
                    out['predict_%s_tail_%s' % (split, k)] = np.float64(v)

        # ---- one training step, dropout = 0, lbl_smooth = 0 (main.py:59-66) ------------------
        params.gcn_drop = params.hidden_drop = params.feat_drop = 0.0
        model.conv1.drop.p = 0.0
        model.conv2.hidden_drop.p = 0.0
        model.conv2.feature_drop.p = 0.0
        params.lbl_smooth = 0.0
        model.train()
        ds = iters['train'].dataset
        nb = min(len(ds), pdict['batch_size'])
        items = [ds[i] for i in range(nb)]
        trip = torch.stack([it[0] for it in items])
        lab = torch.stack([it[1] for it in items])
        model.zero_grad()
        pred = model(trip[:, 0], trip[:, 1], graph)
        loss = model.loss(pred, lab)
        loss.backward()
        out['train_triple'] = trip
        out['train_label'] = lab
        out['train_score'] = pred
        out['train_loss'] = loss
        for k, p in model.named_parameters():
            out['grad_' + k] = p.grad if p.grad is not None else torch.zeros_like(p)
        for k, b in model.named_buffers():
            out['train_after_' + k] = b.clone()
        # smoothed labels as the training dataset produces them (data_loader.py:41-43)
        params.lbl_smooth = 0.1
        out['smooth_label0'] = ds[0][1]
        params.lbl_smooth = 0.0
    else:
        # encoder-only gradient check: loss = sum(all_ent * G) + sum(all_rel * H), train-mode BN
        model.conv1.drop.p = 0.0
        model.train()
        model.zero_grad()
        ent = torch.index_select(model.entity_embedding, 0, graph.entity)
        ee = torch.index_select(model.edge_embeddings, 0, edge_ids)
        all_ent, all_rel = model.conv1(ent, graph.edge_index, edge_type, graph.edge_norm, ee, model.relation_embedding)
        g = torch.Generator().manual_seed(11)
        G = torch.randn(all_ent.shape, generator=g)
        H = torch.randn(all_rel.shape, generator=g)
        ((all_ent * G).sum() + (all_rel * H).sum()).backward()
        out['train_all_ent'] = all_ent
        out['train_all_rel'] = all_rel
        out['train_G'] = G
        out['train_H'] = H
        for k, p in model.named_parameters():
            if k.startswith(keep):
                out['grad_' + k] = p.grad if p.grad is not None else torch.zeros_like(p)
        for k, b in model.named_buffers():
            if k.startswith('conv1.'):
                out['train_after_' + k] = b.clone()

    out['params_json'] = np.array(json.dumps(pdict))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **npify(out))
    print('%-10s N=%d R=%d E=%d keys=%d' % (name, dl.num_entity, dl.num_relation, dl.num_edge, len(out)))
    return out


def main():
    root = tempfile.mkdtemp(prefix='mgcn_golden_')
    data_out = os.path.join(GOLDEN, 'data')
    try:
        # Toy: fixture data shipped by the reference (data/Toy/*.txt) -----------------------------
        os.makedirs(os.path.join(root, 'data'))
        shutil.copytree(os.path.join(REF, 'data', 'Toy'), os.path.join(root, 'data', 'Toy'))
        datasets = {
            'syn_a': synth_triples(1, n_ent=60, n_rel=4, n_train=300, n_eval=24, dup_frac=0.1),
            'syn_b': synth_triples(2, n_ent=500, n_rel=11, n_train=1500, n_eval=40, zipf=1.1),
            'syn_c': synth_triples(3, n_ent=200, n_rel=30, n_train=1200, n_eval=8, zipf=1.3, dup_frac=0.05),
        }
        for name, splits in datasets.items():
            write_dataset(root, name, splits)
        if os.path.isdir(data_out):
            shutil.rmtree(data_out)
        os.makedirs(data_out)
        for name in ['Toy'] + list(datasets):
            os.makedirs(os.path.join(data_out, name))
            for split in ['train', 'valid', 'test']:
                shutil.copyfile(os.path.join(root, 'data', name, split + '.txt'), os.path.join(data_out, name, split + '.txt'))

        small = dict(gcn_in_dim=16, gcn_out_dim=32, k_w=4, k_h=8, num_filter=8, kernel_size=3)
        toy = run_case(root, 'Toy', dict(small, batch_size=4), scale_tables=6.0)
        # the survey's probe values must come out of this run too
        np.testing.assert_allclose(toy['norm_in'].numpy(), [.4082, .2887, .4082, 0, 0, 0, 0, .7071, 0, 0], atol=5e-5)
        np.testing.assert_allclose(toy['norm_out'].numpy(), [0, 0, 0, 0, 0, 0, .4082, .7071, .5774, .7071], atol=5e-5)
        shutil.move(os.path.join(GOLDEN, 'Toy.npz'), os.path.join(GOLDEN, 'toy_small.npz'))
        run_case(root, 'Toy', dict(batch_size=4), full_model=False)
        shutil.move(os.path.join(GOLDEN, 'Toy.npz'), os.path.join(GOLDEN, 'toy_d100.npz'))
        run_case(root, 'syn_a', dict(gcn_in_dim=12, gcn_out_dim=24, k_w=3, k_h=8, num_filter=6, kernel_size=3, bias=True),
                 scale_tables=5.0)
        run_case(root, 'syn_b', dict(gcn_in_dim=20, gcn_out_dim=40, k_w=5, k_h=8, num_filter=8, kernel_size=3),
                 scale_tables=5.0)
        run_case(root, 'syn_c', dict(), full_model=False)

        # untrained Toy, seed 2020, default dims, reference evaluate('test')
        os.chdir(root)
        params, pdict = make_params(batch_size=128)
        dl = ref_dl.DataLoader('Toy', params)
        torch.manual_seed(2020)
        model = ref_model.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
        iters = dl.get_data_loaders(128, 0, params)
        res = ref_main.evaluate(model, iters, dl.graph, params, 'test')
        print('Toy untrained seed 2020 evaluate(test):', res)
        with open(os.path.join(GOLDEN, 'toy_untrained_eval.json'), 'w') as f:
            json.dump({k: float(v) for k, v in res.items()}, f, indent=1)
    finally:
        os.chdir('/')
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
