"""Beyond-cache datapoint for BASELINE.json configs[4] (synthetic KG, dim 512): ONE rank's destination range of a
large graph on one MI355X — the aggregation launch (HBM-bound, far past the 256 MiB Infinity Cache) and the dense
launch of the two-launch layer path (D = 512 is outside the fused kernel). Prints one JSON object.

    python tools/bench_scale.py [N] [E] [R] [D] [world] [rank]
"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('kgc-gcn_amd')
nat = pkg._native
N, E, R, D, W, RANK = [int(a) for a in (sys.argv[1:] + ['2000000', '20000000', '1000', '512', '8', '0'][len(sys.argv) - 1:])]
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
t0 = time.time()
s, r, o = rng.integers(0, N, E), rng.integers(0, R, E), rng.integers(0, N, E)
ei = torch.from_numpy(np.stack((np.concatenate((s, o)), np.concatenate((o, s)))))
et = torch.from_numpy(np.concatenate((r, r + R)))
csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, dev, with_backward=False)
t_build = time.time() - t0
b = pkg.dist.shard_bounds(N, W)
n0, n1 = b[RANK], b[RANK + 1]
slots = sum(csr.shard_slot_counts(n0, n1))
x = torch.randn(N, D, device=dev) * 0.1
rel = torch.randn(2 * R, D, device=dev) * 0.3
loop_rel, loop_edge = torch.randn(D, device=dev), torch.randn(D, device=dev)
# this rank's destinations need their own slots of the per-edge table only; the kernel is handed the FULL slot index
# space, so allocate just the rows it touches by giving it a table view that starts at this range's first in-half slot
ee_full_rows = 2 * E
(i0, i1), (o0, o1), _hub = csr._shard_bounds(n0, n1)
agg = torch.empty((n1 - n0, 3 * D), device=dev)
out = {'N': N, 'E': E, 'R': R, 'D': D, 'world': W, 'rank': RANK, 'dest_range': [n0, n1], 'slots': slots,
       'csr_build_s': round(t_build, 2)}

def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / n * 1e3

# the unfused aggregation kernel indexes the table by global slot; give it a table that covers [0, E + o1) lazily is
# not possible, so use the relation-typed-only variant (ee = None) for the timing at full slot count, and a second
# run WITH a per-edge table on a graph small enough for the whole table (reported separately by the caller).
agg_full = torch.empty((N, 3 * D), device=dev) if N * 3 * D * 4 < 60e9 else None
if agg_full is not None:
    us = timed(lambda: nat.aggregate_fwd(csr, x, rel, None, True, loop_edge, agg_full, loop_rel=loop_rel, node_range=(n0, n1)))
    by = slots * (2 * D * 4 + 16) + (n1 - n0) * (D * 4 + 3 * D * 4)      # x row + rel row per slot (rel L2-resident), records, loop x, A out
    out['aggregate_no_edge_table'] = {'us': us, 'GBps_incl_rel_rows': by / us / 1e3,
                                      'GBps_hbm_only': (slots * (D * 4 + 16) + (n1 - n0) * 4 * D * 4) / us / 1e3}
    w = torch.randn(3 * D, D, device=dev) * 0.05
    o_t = torch.empty((n1 - n0, D), device=dev)
    bnv = [torch.zeros(D, device=dev), torch.ones(D, device=dev), torch.ones(D, device=dev), torch.zeros(D, device=dev)]
    us = timed(lambda: nat.dense_bn_tanh_fwd(agg_full[n0:n1], w, None, bnv[0], bnv[1], bnv[2], bnv[3], 1e-5, o_t))
    out['dense'] = {'us': us, 'TFLOPs': 2.0 * (n1 - n0) * 3 * D * D / us / 1e6}
# with the per-edge table: whole table resident (needs 2E*D*4 bytes)
if 2 * E * D * 4 < 120e9 and agg_full is not None:
    ee = torch.empty((2 * E, D), device=dev).normal_()
    us = timed(lambda: nat.aggregate_fwd(csr, x, rel, ee, True, loop_edge, agg_full, loop_rel=loop_rel, node_range=(n0, n1)))
    by = slots * (2 * D * 4 + 16) + (n1 - n0) * 4 * D * 4
    out['aggregate_with_edge_table'] = {'us': us, 'GBps_hbm': by / us / 1e3, 'frac_of_8TBps': by / us / 1e3 / 8000}
print(json.dumps(out))
