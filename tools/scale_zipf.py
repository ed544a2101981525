"""Hub path at scale (not part of the test suite): a 2M-entity / 20M-triple Zipf(1.1) graph through the feeder, the hub
pre-pass + aggregation forward and the backward (dim 128)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('kgc-gcn_amd'); nat = pkg._native
dev = torch.device('cuda:0')
N, E, R, D = 2_000_000, 20_000_000, 1000, 128
rng = np.random.default_rng(0)
s, r = rng.integers(0, N, E), rng.integers(0, R, E)
w = 1.0 / np.arange(1, N + 1) ** 1.1
o = rng.permutation(N)[rng.choice(N, size=E, p=w / w.sum())]
ei = torch.from_numpy(np.stack((np.concatenate((s, o)), np.concatenate((o, s)))))
et = torch.from_numpy(np.concatenate((r, r + R)))
t0 = time.time()
csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, dev, with_backward=True)
print('feeder %.1f s, hub chunks %d, balanced bounds(8) %s' % (time.time() - t0, csr.num_chunks, csr.balanced_bounds(8)))
x = torch.randn(N, D, device=dev) * 0.1; rel = torch.randn(2 * R, D, device=dev); lr = torch.randn(D, device=dev); le = torch.randn(D, device=dev)
ee = torch.randn(2 * E, D, device=dev)
agg = torch.empty((N, 3 * D), device=dev)
fn = lambda: nat.aggregate_fwd(csr, x, rel, ee, True, le, agg, loop_rel=lr)
fn(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): fn()
torch.cuda.synchronize()
ms = (time.time() - t0) / 5 * 1e3
by = 2 * E * (2 * D * 4 + 16) + N * 4 * D * 4
print('aggregate_fwd with hubs: %.2f ms = %.2f TB/s of compulsory bytes; finite %s' % (ms, by / ms / 1e9, bool(torch.isfinite(agg).all())))
g = torch.randn(N, 3 * D, device=dev)
relf = torch.cat([rel, lr.reshape(1, -1)])
t0 = time.time(); gx, gee, grel = nat.aggregate_bwd(csr, x, relf, ee, g); torch.cuda.synchronize()
print('aggregate_bwd: %.1f ms (first call), finite %s' % ((time.time() - t0) * 1e3, bool(torch.isfinite(gx).all() and torch.isfinite(grel).all())))
