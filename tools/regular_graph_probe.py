"""Probe (not product code): the fused layer on a REGULAR graph (every destination has exactly k slots in each half, so the
two lane groups of a gather wave cross their row boundaries at the same instructions) against a random graph with the same
N and E (Poisson run lengths). Tells how much of the gather's time is the divergence of the row flushes.
Usage: python tools/regular_graph_probe.py [k=2]"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    pkg = importlib.import_module('kgc-gcn_amd')
    nat = pkg._native
    dev = torch.device('cuda:0')
    N, R = 40943, 11
    E = k * N
    rng = np.random.default_rng(0)
    graphs = {
        'regular': (np.concatenate([rng.permutation(N) for _ in range(k)]), np.concatenate([rng.permutation(N) for _ in range(k)])),
        'random': (rng.integers(0, N, E), rng.integers(0, N, E)),
    }
    out = {}
    for name, (s, o) in graphs.items():
        r = rng.integers(0, R, E)
        ei = torch.from_numpy(np.stack((np.concatenate((s, o)), np.concatenate((o, s))))).long()
        et = torch.from_numpy(np.concatenate((r, r + R))).long()
        csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, dev)
        res = []
        for D, O in ((100, 200), (200, 200)):
            torch.manual_seed(0)
            layer = pkg.model.MGCNConv(D, O, 2 * R).to(dev).eval()
            x = torch.randn(N, D, device=dev) * 0.3
            rel = torch.randn(2 * R, D, device=dev) * 0.5
            ee = torch.randn(2 * E, D, device=dev) * 0.5
            bn = layer.ent_bn
            _, wpack = layer.derived_weights()
            o_ = torch.empty((N, O), device=dev)
            rel_out = torch.empty((2 * R, O), device=dev)

            def run():
                nat.layer_fwd_fused(csr, x, rel, layer.loop_rel.reshape(-1), ee, True, layer.loop_edge.reshape(-1), wpack, O, layer.bias,
                                    bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, o_, rels_weight=layer.rels_weight.detach(),
                                    rel_out=rel_out)
            with torch.no_grad():
                for _ in range(30):
                    run()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(200):
                    run()
                b.record()
                torch.cuda.synchronize()
            res.append(round(1e3 * a.elapsed_time(b) / 200, 1))
        out[name] = res
    print(json.dumps({'k': k, 'E': E, 'us_per_layer': out}))


if __name__ == '__main__':
    main()
