"""Per-kernel timings that bench.py's headline does not cover: the scoring / ranking kernels and one
training step (a7) on the WN18RR-shaped synthetic workload. Prints one JSON object."""
import importlib, json, os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module('kgc-gcn_amd')
nat = pkg._native
dev = torch.device('cuda:0')
shape = bench.SHAPES['wn18rr']
N, R, E = shape['N'], shape['R'], shape['E']
O = 200

def timed(fn, n=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

out = {}
g = torch.Generator().manual_seed(0)
ent = (torch.randn(N, O, generator=g) * 0.3).to(dev); bias = (torch.randn(N, generator=g) * 0.1).to(dev)
for B in (128, 1024, 6268):
    x = torch.randn(B, O, generator=g).to(dev)
    obj = torch.randint(0, N, (B,), generator=g).to(dev)
    mask = torch.zeros((B, (N + 31) // 32), dtype=torch.int32, device=dev)
    label = torch.zeros((B, N), device=dev) if B <= 1024 else None
    target = nat.score_target(x, ent, bias, obj)
    counts = torch.zeros((B, 3), dtype=torch.int64, device=dev)
    t_rank = timed(lambda: nat.score_rank(x, ent, bias, obj, target, mask=mask, counts=counts))
    fl = 2.0 * B * O * N
    by = N * O * 4 + mask.numel() * 4
    rec = {'rank_bits_us': t_rank, 'rank_TFLOPs': fl / t_rank / 1e6, 'rank_GBps_compulsory': by / t_rank / 1e3,
           'target_us': timed(lambda: nat.score_target(x, ent, bias, obj, out=target))}
    if label is not None:
        rec['rank_dense_labels_us'] = timed(lambda: nat.score_rank(x, ent, bias, obj, target, label=label, counts=counts))
        rec['score_fwd_us'] = timed(lambda: nat.score_fwd(x, ent, bias))
    out['B=%d' % B] = rec

# one training step (forward + backward + Adam) through the HIP aggregation forward/backward
params = types.SimpleNamespace(gcn_in_dim=100, gcn_out_dim=O, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=10, k_h=20,
                               num_filter=200, kernel_size=7, bias=False, lbl_smooth=0.1, gcn_layers=1, device=dev)
ei, ea = bench.synth_graph(shape, seed=0)
graph = pkg.Graph(edge_index=ei, edge_attr=ea); graph.entity, graph.num_nodes, graph.edge_norm = torch.arange(N), N, None
graph.to(dev)
torch.manual_seed(0)
model = pkg.MGCN(N, R, E, params).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
trip = torch.stack([torch.randint(0, N, (128,)), torch.randint(0, 2 * R, (128,))], 1).to(dev)
lab = (torch.rand(128, N, device=dev) < 1e-4).float()
def step():
    opt.zero_grad()
    loss = model.loss(model(trip[:, 0], trip[:, 1], graph), lab)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
out['train_step_1layer_B128_ms'] = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps(out))
