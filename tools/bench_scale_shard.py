"""BASELINE configs[4] slice on one MI355X through the PRODUCT path of a destination-partitioned rank: the rank's model
holds only its shard of the per-edge table (params.edge_table_rows + dist.shard_model_tables, rows from the chunk-wise
xavier table), the layer runs through dist.encode_layer_rows (fused launch for O <= 208, aggregation + dense launches
otherwise). Prints one JSON object.

    python tools/bench_scale_shard.py [N] [E] [R] [D] [O] [world] [rank]
"""
import importlib, json, os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('kgc-gcn_amd')
N, E, R, D, O, W, RANK = [int(a) for a in (sys.argv[1:] + ['2000000', '20000000', '1000', '512', '512', '8', '0'][len(sys.argv) - 1:])]
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
t0 = time.time()
s, r, o = rng.integers(0, N, E), rng.integers(0, R, E), rng.integers(0, N, E)
ei = torch.from_numpy(np.stack((np.concatenate((s, o)), np.concatenate((o, s)))))
et = torch.from_numpy(np.concatenate((r, r + R)))
graph = pkg.Graph(edge_index=ei, edge_attr=torch.stack([et, torch.arange(2 * E)]))
graph.entity, graph.num_nodes, graph.edge_norm = torch.arange(N), N, None
csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, dev, with_backward=False)
t_build = time.time() - t0
b = csr.balanced_bounds(W)
n0, n1 = b[RANK], b[RANK + 1]
rows = sum(csr.shard_slot_counts(n0, n1))
params = types.SimpleNamespace(gcn_in_dim=D, gcn_out_dim=O, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=8, k_h=O // 8,
                               num_filter=4, kernel_size=3, bias=False, lbl_smooth=0.1, gcn_layers=1, edge_table_rows=rows)
torch.cuda.reset_peak_memory_stats()
t0 = time.time()
torch.manual_seed(0)
model = pkg.MGCN(N, R, E, params).to(dev).eval()
pkg.dist.shard_model_tables(model, csr, n0, n1, lambda li, ids: pkg.dist.xavier_rows(ids, 2 * E, D, 11 + li, dev))
torch.cuda.synchronize()
t_tables = time.time() - t0
layer, table = model.conv1, model.edge_embeddings.detach()
x, rel = model.entity_embedding.detach(), model.relation_embedding.detach()
out = torch.empty((n1 - n0, O), device=dev)
ee_sub = csr.shard_ee_sub(n0, n1)

def run():
    with torch.no_grad():
        pkg.dist.encode_layer_rows(layer, csr, x, rel, table, n0, n1, ee_sub, out=out)

run(); torch.cuda.synchronize()
a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    run()
c.record(); torch.cuda.synchronize()
ms = a.elapsed_time(c) / 5
slots = rows
bytes_alg = slots * (4 * D + 8) + 2 * (n1 - n0 + 1) * 4 + slots * 4 * D * 0 + (n1 - n0) * 4 * D + 4 * N * D * 0 + 16 * D * O + 4 * (n1 - n0) * O
# compulsory bytes of the rank's share: its per-edge rows + records, the x rows its slots gather (each counted once per
# slot: the 4 GB table is far past every cache, so a gathered row is a DRAM access), its own x rows, its output rows
bytes_gather = slots * (4 * D + 4 * D + 16) + (n1 - n0) * (4 * D + 4 * O)
res = {'N': N, 'E': E, 'R': R, 'D': D, 'O': O, 'world': W, 'rank': RANK, 'dest_range': [n0, n1], 'slots': slots,
       'csr_build_s': round(t_build, 2), 'table_shard_GB': table.numel() * 4 / 1e9, 'whole_table_GB': 2 * E * D * 4 / 1e9,
       'table_fill_s': round(t_tables, 1), 'peak_GB': torch.cuda.max_memory_allocated() / 1e9,
       'path': 'fused' if pkg._native.fused_supported(D, O) else 'aggregate + dense', 'layer_ms': ms,
       'edges_per_s_this_rank': (slots + (n1 - n0)) / ms * 1e3,
       'GBps_of_gathered_bytes': bytes_gather / ms / 1e6, 'frac_of_8TBps': bytes_gather / ms / 1e6 / 8000}
print(json.dumps(res))
