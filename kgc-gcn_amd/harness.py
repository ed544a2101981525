"""The callers of the hot path: one training epoch, filtered-rank prediction and evaluation — what
main.py:49-135 does, against the same model / loader surface, so the two are interchangeable.

`predict` has two forms of the same computation:
  * fused=True  (default): model.rank_counts — the HIP score+filter+count kernel, no [B, N] score
    matrix, rank = 1 + gt + ties_lower (stable tie rule);
  * fused=False: the reference's sequence on torch GPU ops over model(...)'s scores (main.py:121-126),
    double argsort included — kept for A/B timing and as the literal drop-in behaviour.
Both agree exactly on rows without ties.
"""
import logging

import numpy as np
import torch
import torch.nn as nn

from .utils import RunningAverage


def train(model, data_iter, graph, optimizer, params):
    """One epoch (main.py:49-77). Returns the running loss."""
    model.train()
    loss_avg = RunningAverage()
    for triplets, labels in data_iter:
        optimizer.zero_grad()
        triplets = triplets.to(params.device)
        pred = model(triplets[:, 0], triplets[:, 1], graph)
        loss = model.loss(pred, labels.to(params.device))
        loss.backward()
        nn.utils.clip_grad_norm_(parameters=model.parameters(), max_norm=params.clip_grad)
        optimizer.step()
        loss_avg.update(loss.item())
    return loss_avg()


def train_device_labels(model, queries, index, graph, optimizer, params, batch_size, generator=None, fused_loss=True):
    """One epoch like `train`, but the label rows never exist on the host: `queries` [Q, 2] int64
    (DataLoader.train_queries()), `index` = DataLoader.train_index() on the device; per step only B keys are indexed
    and mgcn_label_rows writes the smoothed [B, N] targets next to the scores. Shuffles like the reference's loader
    (data_loader.py:190) with `generator`. Same loss / clipping / optimizer calls as main.py:56-70."""
    from . import _native
    model.train()
    loss_avg = RunningAverage()
    dev = params.device
    queries = queries.to(dev)
    order = torch.randperm(queries.size(0), generator=generator).to(dev)
    n_ent = model.entity_embedding.size(0)
    for i in range(0, queries.size(0), batch_size):
        q = queries.index_select(0, order[i:i + batch_size])
        optimizer.zero_grad()
        if fused_loss:                                   # scores, targets and loss in one launch
            loss = model.forward_loss(q[:, 0], q[:, 1], graph, index, lbl_smooth=params.lbl_smooth)
        else:
            labels = _native.label_rows(index.query_keys(q[:, 0], q[:, 1]), index.keys, index.ptr, index.tails, n_ent,
                                        lbl_smooth=params.lbl_smooth)
            loss = model.loss(model(q[:, 0], q[:, 1], graph), labels)
        loss.backward()
        nn.utils.clip_grad_norm_(parameters=model.parameters(), max_norm=params.clip_grad)
        optimizer.step()
        loss_avg.update(loss.item())
    return loss_avg()


def ranks_from_scores(pred, label, obj):
    """main.py:122-126 on a materialised score block."""
    rows = torch.arange(pred.size(0), device=pred.device)
    target = pred[rows, obj]
    pred = torch.where(label.to(torch.uint8).bool(), torch.full_like(pred, -10000000.0), pred)
    pred[rows, obj] = target
    order = torch.argsort(torch.argsort(pred, dim=1, descending=True), dim=1, descending=False)
    return 1 + order[rows, obj]


def _accumulate(results, ranks):
    ranks = ranks.float()
    results['count'] = torch.numel(ranks) + results.get('count', 0.0)
    results['mr'] = torch.sum(ranks).item() + results.get('mr', 0.0)
    results['mrr'] = torch.sum(1.0 / ranks).item() + results.get('mrr', 0.0)
    hits = (ranks.view(-1, 1) <= torch.arange(1, 11, device=ranks.device, dtype=ranks.dtype)).sum(0).tolist()
    for k in range(10):
        results['hits@{}'.format(k + 1)] = hits[k] + results.get('hits@{}'.format(k + 1), 0.0)
    return results


def predict(model, data_iters, graph, data_type, device, mode='tail_batch', fused=True):
    """main.py:105-135: sums of rank, 1/rank and hits@1..10 over one side of a split."""
    model.eval()
    results = {}
    with torch.no_grad():
        for triplets, label in data_iters['{}_{}'.format(data_type, mode.split('_')[0])]:
            triplets, label = triplets.to(device), label.to(device)
            sub, rel, obj = triplets[:, 0], triplets[:, 1], triplets[:, 2]
            if fused:
                counts, _ = model.rank_counts(sub, rel, obj.contiguous(), label, graph)
                ranks = 1 + counts[:, 0] + counts[:, 1]
            else:
                ranks = ranks_from_scores(model(sub, rel, graph), label, obj)
            _accumulate(results, ranks)
    return results


def evaluate(model, data_iters, graph, params, data_type, mark='Val', hits=(1, 3, 10), fused=True):
    """main.py:80-102: tail + head sides averaged, rounded to 5 decimals."""
    tail = predict(model, data_iters, graph, data_type, params.device, mode='tail_batch', fused=fused)
    head = predict(model, data_iters, graph, data_type, params.device, mode='head_batch', fused=fused)
    count = float(tail['count'])
    results = {'mr': np.round((tail['mr'] + head['mr']) / (2 * count), 5),
               'mrr': np.round((tail['mrr'] + head['mrr']) / (2 * count), 5)}
    for k in hits:
        results['hits@{}'.format(k)] = np.round((tail['hits@{}'.format(k)] + head['hits@{}'.format(k)]) / (2 * count), 5)
    logging.info('- {} metrics: {}  '.format(mark, '; '.join('{}: {:05.3f}'.format(k, v) for k, v in results.items())))
    return results
