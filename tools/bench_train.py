"""One training step (main.py:56-70: forward, BCE, backward, clip, Adam) at a benchmark shape: wall-clock per step.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split. Synthetic batch: B queries with label rows of
the loader's form (multi-hot, label smoothing applied).

    python tools/bench_train.py [--shape wn18rr] [--layers 1] [--steps 30] [--batch 128]
"""
import argparse
import importlib
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (shapes + graph generator)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='wn18rr', choices=sorted(bench.SHAPES))
    ap.add_argument('--layers', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--batch', type=int, default=128)
    args = ap.parse_args()
    pkg = importlib.import_module('kgc-gcn_amd')
    dev = torch.device('cuda', 0)
    shape = bench.SHAPES[args.shape]
    N, R, E = shape['N'], shape['R'], shape['E']
    D, O = 100, 200
    params = types.SimpleNamespace(gcn_in_dim=D, gcn_out_dim=O, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=10,
                                   k_h=20, num_filter=200, kernel_size=7, bias=False, lbl_smooth=0.1,
                                   gcn_layers=args.layers, clip_grad=1.0, device=dev)
    edge_index, edge_attr = bench.synth_graph(shape, seed=0)
    graph = pkg.Graph(edge_index=edge_index, edge_attr=edge_attr)
    graph.entity, graph.num_nodes, graph.edge_norm = torch.arange(N), N, None
    graph.to(dev)
    torch.manual_seed(0)
    model = pkg.MGCN(N, R, E, params).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(2)
    B = args.batch
    trip = torch.stack([torch.randint(0, N, (B,), generator=g), torch.randint(0, 2 * R, (B,), generator=g)], 1).to(dev)
    label = torch.zeros(B, N)
    label[torch.arange(B).repeat_interleave(4), torch.randint(0, N, (4 * B,), generator=g)] = 1.0
    label = ((1.0 - params.lbl_smooth) * label + 1.0 / N).to(dev)            # data_loader.py:48-49

    def step():
        model.train()
        opt.zero_grad()
        pred = model(trip[:, 0], trip[:, 1], graph)
        loss = model.loss(pred, label)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(parameters=model.parameters(), max_norm=params.clip_grad)
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    marks = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
        return out

    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / args.steps
    for _ in range(args.steps):                                            # same step, synchronised between phases
        model.train()
        opt.zero_grad()
        pred = timed('forward', lambda: model(trip[:, 0], trip[:, 1], graph))
        loss = timed('loss', lambda: model.loss(pred, label))
        timed('backward', loss.backward)
        timed('clip', lambda: torch.nn.utils.clip_grad_norm_(parameters=model.parameters(), max_norm=params.clip_grad))
        timed('adam', opt.step)
    print('%s  layers %d  batch %d: %.3f ms per training step (loss %.5f)' % (args.shape, args.layers, B, total * 1e3, float(loss)))
    # target rows: what the reference's loader does per step (per-sample numpy rows, stack, host-to-device copy of
    # [B, N] f32, data_loader.py:34-51 + main.py:62) against mgcn_label_rows on the device
    import numpy as np
    known = {}
    for s_, r_ in trip.tolist():
        known.setdefault((s_, r_), set()).update(int(v) for v in torch.randint(0, N, (4,), generator=g))
    index = pkg.dist.FilterIndex.from_known(known, 2 * R).to(dev)
    items = [{'triple': (s_, r_, -1), 'label': sorted(known[(s_, r_)])} for s_, r_ in trip.tolist()]
    ds = pkg.data_loader.KBDataset(items, N, params, training=True)
    def host_rows():
        batch = ds.collate_fn([ds[i] for i in range(B)])
        return batch[1].to(dev)
    def device_rows():
        return pkg._native.label_rows(index.query_keys(trip[:, 0], trip[:, 1]), index.keys, index.ptr, index.tails, N,
                                      lbl_smooth=params.lbl_smooth)
    for name, fn in (('host rows + copy', host_rows), ('mgcn_label_rows', device_rows)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = fn()
        torch.cuda.synchronize()
        print('  targets for one step, %-18s %.3f ms' % (name + ':', (time.perf_counter() - t0) / 10 * 1e3))
    assert torch.equal(host_rows(), device_rows())
    # the same step with scores + targets + BCE + d loss / d logits in one launch (model.forward_loss)
    def fused_step():
        model.train()
        opt.zero_grad()
        loss = model.forward_loss(trip[:, 0], trip[:, 1], graph, index, lbl_smooth=params.lbl_smooth)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(parameters=model.parameters(), max_norm=params.clip_grad)
        opt.step()
        return loss
    def two_step():
        model.train()
        opt.zero_grad()
        loss = model.loss(model(trip[:, 0], trip[:, 1], graph), device_rows())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(parameters=model.parameters(), max_norm=params.clip_grad)
        opt.step()
        return loss
    for name, fn in (('label_rows + forward + BCELoss', two_step), ('forward_loss (fused)', fused_step)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            l = fn()
        torch.cuda.synchronize()
        print('  training step, %-32s %.3f ms (loss %.5f)' % (name + ':', (time.perf_counter() - t0) / args.steps * 1e3, float(l)))
    print('  phases (synchronised): ' + '  '.join('%s %.3f ms' % (k, v / args.steps * 1e3) for k, v in marks.items()))


if __name__ == '__main__':
    main()
