"""Where the filtered-MRR evaluation's wall-clock goes (WN18RR shape, 6268 queries): encoder, ConvE trunk, filter bits,
target scores, score+rank kernel, metrics — each synchronised, for the batched (128) and the one-shot form."""
import importlib, os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module('kgc-gcn_amd'); nat = pkg._native
dev = torch.device('cuda:0')
shape = bench.SHAPES['wn18rr']; N, R, E = shape['N'], shape['R'], shape['E']
params = types.SimpleNamespace(gcn_in_dim=100, gcn_out_dim=200, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=10, k_h=20,
                               num_filter=200, kernel_size=7, bias=False, lbl_smooth=0.1, gcn_layers=2, device=dev)
ei, ea = bench.synth_graph(shape, seed=0)
graph = pkg.Graph(edge_index=ei, edge_attr=ea); graph.entity, graph.num_nodes, graph.edge_norm = torch.arange(N), N, None
graph.to(dev)
torch.manual_seed(0)
model = pkg.MGCN(N, R, E, params).to(dev).eval()
g = torch.Generator().manual_seed(3)
Q = 6268
queries = torch.stack([torch.randint(0, N, (Q,), generator=g), torch.randint(0, 2 * R, (Q,), generator=g),
                       torch.randint(0, N, (Q,), generator=g)], 1)
known = {}
for s, r, o in queries.tolist():
    known.setdefault((s, r), set()).add(o)
filt = pkg.dist.FilterIndex.from_known(known, 2 * R).to(dev)

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out

with torch.no_grad():
    ms_enc, (all_ent, all_rel) = t(lambda: (setattr(model, '_enc_cache', None), model.encode(graph))[1])
    q = queries.to(dev); sub, rel, obj = q[:, 0], q[:, 1], q[:, 2].contiguous()
    ms_trunk, x = t(lambda: torch.cat([model.conv2.trunk(all_ent.index_select(0, sub[i:i + 2048]), all_rel.index_select(0, rel[i:i + 2048]))
                                        for i in range(0, Q, 2048)], dim=0))
    keys = filt.query_keys(sub, rel)
    bias = model.conv2.bias.contiguous()
    print('encoder (2 layers, hipGraph replay) %.3f ms   trunk over %d queries %.3f ms' % (ms_enc, Q, ms_trunk))
    for step in (128, Q):
        def targets():
            return [nat.score_target(x[i:i + step], all_ent, bias, obj[i:i + step]) for i in range(0, Q, step)]
        ms_t, tg = t(targets)
        def masks():
            return [nat.filter_mask(keys[i:i + step], filt.keys, filt.ptr, filt.tails, N) for i in range(0, Q, step)]
        ms_m, mk = t(masks)
        def ranks():
            return [nat.score_rank(x[i:i + step], all_ent, bias, obj[i:i + step], tg[k], mask=mk[k]) for k, i in enumerate(range(0, Q, step))]
        ms_r, _ = t(ranks)
        ms_all, res = t(lambda: pkg.dist.evaluate_sharded(model, graph, queries, filt, batch_size=None if step == Q else step))
        print('block %5d: target %.3f ms  filter bits %.3f ms  score+rank %.3f ms (%.1f TF/s)   evaluate_sharded total %.3f ms (encoder cached)'
              % (step, ms_t, ms_m, ms_r, 2.0 * Q * 200 * N / ms_r / 1e9, ms_all))
