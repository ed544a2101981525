"""BASELINE configs[4] — one rank's destination range of the synthetic 10 M-entity / 100 M-triple / 1 k-relation / dim-512 layer
(or a scaled-down graph) on one MI355X through the PRODUCT path of a destination-partitioned rank: the rank's model holds
only its shard of the per-edge table (params.edge_table_rows + dist.shard_model_tables, rows from the chunk-wise xavier
table), the layer runs through dist.encode_layer_rows (one fused launch for O <= 512). Prints one JSON object per output
width and a last line with all of them (not product code; run under rocprofv3 --kernel-trace --stats for profiles/).

    python tools/bench_scale_shard.py [N] [E] [R] [D] [O[,O2,...]] [world] [rank]
"""
import importlib, json, os, resource, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('kgc-gcn_amd')
argv = sys.argv[1:] + ['2000000', '20000000', '1000', '512', '512', '8', '0'][len(sys.argv) - 1:]
N, E, R, D = [int(a) for a in argv[:4]]
OS = [int(v) for v in argv[4].split(',')]
W, RANK = int(argv[5]), int(argv[6])
dev = torch.device('cuda:0')
HBM = 8000.0


def rss_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6     # (Linux: kilobytes)


def say(msg):
    print('[%7.1f s] %s' % (time.time() - T0, msg), file=sys.stderr, flush=True)


T0 = time.time()
rng = np.random.default_rng(0)
s, r, o = rng.integers(0, N, E), rng.integers(0, R, E), rng.integers(0, N, E)
ei = torch.from_numpy(np.stack((np.concatenate((s, o)), np.concatenate((o, s)))))
et = torch.from_numpy(np.concatenate((r, r + R)))
del s, r, o
say('edge list built (2E = %d)' % et.numel())
t0 = time.time()
csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, dev, with_backward=False)      # the host feeder: 2E slots, hubs cut into chunks
t_build = time.time() - t0
del ei, et
say('CSR built in %.1f s (%d hub chunks)' % (t_build, csr.num_chunks))
b = csr.balanced_bounds(W)
n0, n1 = b[RANK], b[RANK + 1]
rows = sum(csr.shard_slot_counts(n0, n1))
O0 = OS[0]
params = types.SimpleNamespace(gcn_in_dim=D, gcn_out_dim=O0, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=8, k_h=O0 // 8,
                               num_filter=4, kernel_size=3, bias=False, lbl_smooth=0.1, gcn_layers=1, edge_table_rows=rows)
torch.cuda.reset_peak_memory_stats()
t0 = time.time()
torch.manual_seed(0)
model = pkg.MGCN(N, R, E, params).to(dev).eval()
say('model on the device (entity table %.1f GB, table shard %.1f GB)' % (N * D * 4 / 1e9, rows * D * 4 / 1e9))
pkg.dist.shard_model_tables(model, csr, n0, n1, lambda li, ids: pkg.dist.xavier_rows(ids, 2 * E, D, 11 + li, dev))
torch.cuda.synchronize()
t_tables = time.time() - t0
say('table shard filled')
table = model.edge_embeddings.detach()
x, rel = model.entity_embedding.detach(), model.relation_embedding.detach()
ee_sub = csr.shard_ee_sub(n0, n1)
nr = n1 - n0
results = []
for O in OS:
    if O == O0:
        layer = model.conv1
    else:
        torch.manual_seed(O)
        layer = pkg.model.MGCNConv(D, O, 2 * R).to(dev).eval()
    out = torch.empty((nr, O), device=dev)

    def run():
        with torch.no_grad():
            pkg.dist.encode_layer_rows(layer, csr, x, rel, table, n0, n1, ee_sub, out=out)

    run(); torch.cuda.synchronize()
    reps = 5 if E <= 30000000 else 3
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        run()
    c.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(c) / reps
    pkg._native.check_fused_status(dev)
    # compulsory bytes of the rank's launch, every distinct byte once — its per-edge rows + source + type indices,
    # its row pointers, the layer input ONCE (the rank's sources are all over [N, D]), relation table, weights, its output rows
    bytes_8d = rows * (4 * D + 8) + 2 * (nr + 1) * 4 + N * 4 * D + (2 * R + 1) * 4 * D + 16 * D * O + nr * 4 * O
    # what the launch must actually move when nothing is cache-resident: a gathered x row per slot (the input is far past
    # every cache), the per-edge row, the 16-byte record; its own x rows (self loop) and its output rows
    bytes_gather = rows * (4 * D + 4 * D + 16) + nr * (4 * D + 4 * O)
    res = {'N': N, 'E': E, 'R': R, 'D': D, 'O': O, 'world': W, 'rank': RANK, 'dest_range': [n0, n1], 'slots': rows,
           'hub_chunks_this_rank': csr.chunk_range(n0, n1)[1] - csr.chunk_range(n0, n1)[0],
           'csr_build_s': round(t_build, 2), 'table_shard_GB': table.numel() * 4 / 1e9, 'whole_table_GB': 2 * E * D * 4 / 1e9,
           'layer_input_GB': N * D * 4 / 1e9, 'table_fill_s': round(t_tables, 1), 'peak_device_GB': torch.cuda.max_memory_allocated() / 1e9,
           'peak_host_GB': round(rss_gb(), 1),
           'kernel_generation': pkg._native.lib().mgcn_fused_kernel_generation(D, O, nr, 1),
           'path': 'fused' if pkg._native.fused_supported(D, O) else 'aggregate + dense', 'layer_ms': ms,
           'edges_per_s_this_rank': (rows + nr) / ms * 1e3,
           'bytes_8d_GB': bytes_8d / 1e9, 'frac_8d_of_8TBps': bytes_8d / ms / 1e6 / HBM,
           'bytes_gathered_GB': bytes_gather / 1e9, 'GBps_of_gathered_bytes': bytes_gather / ms / 1e6,
           'frac_gathered_of_8TBps': bytes_gather / ms / 1e6 / HBM}
    say('O = %d: %.2f ms' % (O, ms))
    print(json.dumps(res), flush=True)
    results.append(res)
    del out
print(json.dumps({'runs': results}))
