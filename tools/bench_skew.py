"""Fused layer launch on uniform vs hub-heavy (Zipf destination) graphs of the FB15k-237 shape: how much does degree
skew cost the gather role (lane groups own fixed destination runs)? Prints one line per variant."""
import importlib, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('kgc-gcn_amd'); nat = pkg._native
dev = torch.device('cuda:0')
N, R, E, D, O = 14541, 237, 272115, 100, 200
def graph(zipf, seed=0):
    rng = np.random.default_rng(seed)
    s, r = rng.integers(0, N, E), rng.integers(0, R, E)
    if zipf > 0:
        p = 1.0 / np.arange(1, N + 1) ** zipf
        o = rng.permutation(N)[rng.choice(N, size=E, p=p / p.sum())]
    else:
        o = rng.integers(0, N, E)
    ei = torch.from_numpy(np.stack((np.concatenate((s, o)), np.concatenate((o, s)))))
    et = torch.from_numpy(np.concatenate((r, r + R)))
    return pkg.GraphCSR(N, 2 * R + 1, ei, et, dev, with_backward=True)
torch.manual_seed(0)
conv = pkg.MGCNConv(D, O, 2 * R).to(dev).eval()
x = torch.randn(N, D, device=dev) * 0.1; rel = torch.randn(2 * R, D, device=dev) * 0.3; ee = torch.randn(2 * E, D, device=dev)
_, wpack = conv.derived_weights(); bn = conv.ent_bn
out = torch.empty((N, O), device=dev)
for zipf in (0.0, 0.8, 1.1, 1.4):
    csr = graph(zipf)
    deg = (csr.rowptr[:, 1:] - csr.rowptr[:, :-1]).max().item()
    fn = lambda: nat.layer_fwd_fused(csr, x, rel, conv.loop_rel.reshape(-1), ee, True, conv.loop_edge.reshape(-1), wpack, O,
                                     None, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, out)
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    b.record(); torch.cuda.synchronize()
    agg = torch.empty((N, 3 * D), device=dev)
    fn2 = lambda: nat.aggregate_fwd(csr, x, rel, ee, True, conv.loop_edge.reshape(-1), agg, loop_rel=conv.loop_rel.reshape(-1))
    fn2(); torch.cuda.synchronize()
    c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c.record()
    for _ in range(50): fn2()
    d.record(); torch.cuda.synchronize()
    gA = torch.randn(N, 3 * D, device=dev)
    relfull = torch.cat([rel, conv.loop_rel.detach().reshape(1, -1)])
    fn3 = lambda: nat.aggregate_bwd(csr, x, relfull, ee, gA)
    fn3(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn3()
    e1.record(); torch.cuda.synchronize()
    print('zipf %.1f  max run %5d  hub chunks %6d  fused layer %7.1f us   aggregate-only %7.1f us   backward (gx+gee+grel) %8.1f us'
          % (zipf, deg, csr.num_chunks, a.elapsed_time(b) / 50 * 1e3, c.elapsed_time(d) / 50 * 1e3, e0.elapsed_time(e1) / 20 * 1e3))
