#!/usr/bin/env python3
"""bench.py — aggregated edges/sec of the full-graph M-GCN encoder forward (+ filtered-MRR evaluation wall-clock).

Headline workload (BASELINE.json configs[1]): synthetic graph of the public WN18RR shape (N=40 943, R=11, E=86 835
train triples -> 173 670 directed edges + N self loops per layer), 2 layers 100 -> 200 -> 200, eval mode. A "step" is
one encoder forward over the whole graph: one fused launch per layer (aggregation + dense step + BN + tanh + relation
projection), replayed from a hipGraph. Inputs (tables, CSR, weights) are resident in HBM before the timed region.
The same run also reports, in sub-objects of the ONE JSON line printed by rank 0:
  kernels / roofline   per-launch device time (HIP events on the launch stream) priced against BOTH roofs (SURVEY §8d
                       bytes / 8 TB/s, and the dense step's flops / MFMA peak), PMC traffic from profiles/;
  fb15k237             the same step on the FB15k-237 shape (BASELINE configs[2]) with its own kernels and CPU baseline;
  eval                 full filtered-MRR evaluation wall-clock on the WN18RR shape (three forms) and, under
                       eval.fb15k237, BASELINE configs[3]: the FB15k-237-shape evaluation with the entity table sharded
                       over all ranks (RCCL) next to the same evaluation on ONE rank, so the N-GPU gain reads off one line;
  cpu_baseline         the oracle on this box's host cores (N = 1 only).

`python bench.py --gpus N --steps K --warmup W`. For N > 1 either run it under torch.distributed.run (one rank per
GPU) or plainly: without WORLD_SIZE in the environment it starts the N ranks itself (a torch.distributed.run child,
before this process touches the GPU) and relays their output.
"""
import argparse
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPES = {  # public dataset shapes
    'wn18rr': dict(N=40943, R=11, E=86835, n_eval=3134),
    'fb15k237': dict(N=14541, R=237, E=272115, n_eval=20466),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # exact-f32 MFMA: what the algorithmic flops of the dense step cost in f32
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA: what the kernel's six split products are issued on
KERNEL_SOURCES = ('layer_fused2.hip', 'layer_fused3.hip', 'layer_fused.hip', 'aggregate.hip')
TRAFFIC_FILE = 'r04_traffic.json'


def synth_graph(shape, seed=0, zipf=0.0):
    """Random triples of the given shape (numpy default_rng, duplicates kept), plus the bi-directional edge list exactly
    as the feeder expects it (data_loader.py:143-149). Uniform endpoints by default; zipf > 0 draws the tails from a
    Zipf-like law over a random permutation of the entities (the power-law profile that exercises hubs)."""
    rng = np.random.default_rng(seed)
    N, R, E = shape['N'], shape['R'], shape['E']
    s = rng.integers(0, N, size=E)
    r = rng.integers(0, R, size=E)
    if zipf > 0:
        w = 1.0 / np.arange(1, N + 1) ** zipf
        o = rng.permutation(N)[rng.choice(N, size=E, p=w / w.sum())]
    else:
        o = rng.integers(0, N, size=E)
    edge_index = np.stack((np.concatenate((s, o)), np.concatenate((o, s))))
    edge_attr = np.stack((np.concatenate((r, r + R)), np.arange(2 * E, dtype=np.int64)))
    return torch.from_numpy(edge_index), torch.from_numpy(edge_attr)


def layer_bytes(N, E2, R2, D, O):
    """Algorithmic (compulsory) bytes of one layer forward, SURVEY §8(d) / BASELINE.md §3."""
    return E2 * (4 * D + 8) + 2 * (N + 1) * 4 + 4 * N * D + 4 * (R2 + 1) * D + 16 * D * O + 4 * N * O


def agg_kernel_bytes(N, E2, R2, D):
    """Compulsory bytes of the aggregation launch alone: slot records as laid out (16 B), per-edge rows,
    row pointers, layer input, relation table, and the [N, 3D] aggregate it writes."""
    return E2 * (4 * D + 16) + 2 * (N + 1) * 4 + 4 * N * D + 4 * (R2 + 1) * D + 4 * N * 3 * D


def split_mfma_flops(N, D, O):
    """MFMA flops the fused kernel actually issues per layer: six bf16 products over K padded to 32-wide k-blocks, the
    output padded to 13 (O <= 208) or 32 column tiles of 16, rows to 16-row tiles."""
    if D <= 256 and O <= 208:     # layer_fused2.hip: 128-column chunks, 80-row tiles, column tiles by O
        kblocks, left = 0, D
        while left > 0:
            w = min(left, 128)
            kblocks += (w + 31) // 32
            left -= w
        return 6 * 2.0 * ((N + 79) // 80 * 80) * (3 * kblocks * 32) * ((O + 15) // 16 * 16)
    kblocks = (D + 31) // 32        # layer_fused3.hip
    rows = (N + 15) // 16 * 16
    return 6 * 2.0 * rows * (3 * kblocks * 32) * (208 if O <= 208 else 512)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a torch.distributed.run child of THIS process
    (which has not touched the GPU: torch.cuda is never initialised here) and relay its output and exit code."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, cwd=os.getcwd()).returncode


def make_model(pkg, shape, dev, layers, seed, zipf, D=100, O=200):
    N, R, E = shape['N'], shape['R'], shape['E']
    params = types.SimpleNamespace(gcn_in_dim=D, gcn_out_dim=O, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=10,
                                   k_h=20, num_filter=200, kernel_size=7, bias=False, lbl_smooth=0.1,
                                   gcn_layers=layers, cache_encoder=False, device=dev)
    edge_index, edge_attr = synth_graph(shape, seed=seed, zipf=zipf)
    graph = pkg.Graph(edge_index=edge_index, edge_attr=edge_attr)
    graph.entity, graph.num_nodes, graph.edge_norm = torch.arange(N), N, None
    graph.to(dev)
    torch.manual_seed(0)
    model = pkg.MGCN(N, R, E, params)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                                   # non-trivial BN statistics
        for layer in [model.conv1] + list(model.conv1_extra):
            layer.ent_bn.running_mean.copy_(torch.randn(O, generator=g) * 0.05)
            layer.ent_bn.running_var.copy_(torch.rand(O, generator=g) * 0.5 + 0.05)
    model.to(dev).eval()
    return model, graph, params, edge_index, edge_attr


def timed_steps(model, graph, steps, warmup, barrier, ramp_s=0.4):
    """W untimed warm-up steps, then exactly K timed ones between two barriers. Before the warm-up the same step runs
    untimed for `ramp_s` seconds: a fresh process otherwise measures the clock ramp and first-touch page faults of a
    0.2 ms step instead of its steady state (--steps 20 --warmup 5 read 16 % slower than --steps 200)."""
    def step():
        with torch.no_grad():
            return model.encode(graph)

    t_end = time.perf_counter() + ramp_s
    while time.perf_counter() < t_end:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--shape', default='wn18rr', choices=sorted(SHAPES))
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--zipf', type=float, default=0.0, help='tail endpoints ~ Zipf(a) instead of uniform (hub-heavy profile)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-eval', action='store_true')
    ap.add_argument('--no-fb', action='store_true', help='skip the FB15k-237-shape sub-objects (configs[2], configs[3])')
    ap.add_argument('--no-scale', action='store_true', help='skip the "scale" object (partitioned encoder, configs[4] slice)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))                         # nothing above has initialised the GPU in this process
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path has no CPU fallback')
    # one rank per GPU; the modulo only matters when a multi-rank run is REHEARSED on a box with fewer GPUs
    # (MGCN_DIST_BACKEND=gloo, ranks sharing a card) — on the 8-GPU node it is the identity
    dev = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('MGCN_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group(backend, device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pkg = importlib.import_module('kgc-gcn_amd')
    shape = SHAPES[args.shape]
    N, R, E = shape['N'], shape['R'], shape['E']
    D, O = 100, 200
    # every rank works on its own graph of the same shape (seed = rank): per-GPU work fixed -> weak scaling
    model, graph, params, edge_index, edge_attr = make_model(pkg, shape, dev, args.layers, rank, args.zipf)
    elapsed = timed_steps(model, graph, args.steps, args.warmup, barrier)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    edges_per_step = args.layers * (2 * E + N)               # per rank
    value = world * edges_per_step * args.steps / elapsed
    result = {
        'metric': 'aggregated_edges_per_sec', 'value': value, 'unit': 'edges/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak' if world == 1 else 'replicas', 'vs_baseline': None,
        'dtype': 'f32 (dense step as six bf16-split MFMA products, f32 accumulation: f32-faithful)', 'data': 'synthetic',
        'config': {'workload': '%s-shape synthetic graph (N=%d, R=%d, E=%d%s), %d-layer M-GCN encoder %s, full-graph '
                               'forward, eval mode' % (args.shape, N, R, E, ', Zipf(%.2f) tails' % args.zipf if args.zipf else '',
                                                       args.layers, '->'.join(map(str, [D] + [O] * args.layers))),
                   'edges_per_step_per_gpu': edges_per_step,
                   'n_ranks_seen': dist.get_world_size() if dist is not None else 1,
                   'parallelism': 'one graph of this shape per GPU x%d, no data-path collective in the encoder step; '
                                  'the RCCL exchange of the sharded scoring pass is timed in "eval"' % world},
    }

    if rank == 0:
        result.update(kernel_breakdown(pkg, model, graph, args.steps, shape, args.shape, args.layers, D, O))
        result['train'] = guarded(lambda: train_breakdown(pkg, model, graph, shape, D, O))
    if not args.no_eval:                                    # every rank takes part (collectives when W > 1)
        if rank != 0:                                       # the sharded pass needs ONE graph on all ranks: rank 0's
            del model, graph
            model, graph, params, edge_index, edge_attr = make_model(pkg, shape, dev, args.layers, 0, args.zipf)
        ev = eval_wallclock(pkg, model, graph, params, shape, dev, edge_index, edge_attr, world, rank)
        if rank == 0:
            result['eval'] = ev
    scale = {}
    if not args.no_scale:                                   # every rank takes part
        if rank != 0 and args.no_eval:                      # (the partitioned encoder needs ONE graph on all ranks: rank 0's)
            del model, graph
            model, graph, params, edge_index, edge_attr = make_model(pkg, shape, dev, args.layers, 0, args.zipf)
        scale[args.shape] = guarded(lambda: encode_sharded_timing(pkg, model, graph, shape, dev, world, rank, dist, O))
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        result['cpu_baseline'] = cpu_baseline(model, edge_index, edge_attr, args.shape, args.layers, N, R, E, D, O, 20.0)
        result['config']['gpu_over_cpu'] = value / result['cpu_baseline']['value']

    if rank == 0:
        real = real_dataset_sections(pkg, args, dev, D, O)
        if real:
            result['real_datasets'] = real
    if not args.no_fb and args.shape == 'wn18rr':
        del model, graph
        torch.cuda.empty_cache()
        fb = fb_sections(pkg, args, dev, world, rank, dist, barrier, D, O)
        if rank == 0:
            result['fb15k237'] = fb['step']
            if 'eval' in fb:
                result.setdefault('eval', {})['fb15k237'] = fb['eval']
        if 'scale' in fb:
            scale['fb15k237'] = fb['scale']
    if not args.no_scale:
        torch.cuda.empty_cache()
        c5 = guarded(lambda: config5_slice(pkg, dev, world, rank, dist))
        if dist is not None:          # outside the guarded body: every rank reaches this collective whatever happened inside
            tt = torch.tensor([c5.get('layer_ms', -1.0)], dtype=torch.float64, device=dev)
            lo = tt.clone()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            c5['layer_ms_max_over_ranks'] = float(tt.item())
            if float(lo.item()) < 0:
                c5['error_on_some_rank'] = True
        else:
            c5['layer_ms_max_over_ranks'] = c5.get('layer_ms')
        scale['config5_slice'] = c5
        if rank == 0:
            scale['note'] = ('SURVEY 8(e) measured: "encode_sharded_s" = the destination-partitioned 2-layer encoder (each rank its '
                             'work-balanced destination range and table shard, all-gather of every layer output included), same '
                             'graph on all ranks: strong scaling against "one_rank_s" (rank 0 alone, same process, no '
                             'collective); "config5_slice" = one rank\'s 1/8 of a 2M-entity / 20M-triple / dim-512 layer '
                             '(BASELINE configs[4] scaled to what one box builds in seconds), every rank its own slice')
            result['scale'] = scale
    if rank == 0 and world > 1:
        # What `value` is for N > 1: N independent replicas of the step (no data-path collective: linear by construction,
        # "scaling": "replicas"). The quantities that DO exchange data over RCCL are lifted to the top level: strong-scaling
        # speed-ups of the destination-partitioned encoder and of the entity-sharded evaluation (north_star's ">= 6x at 8 GPUs"
        # is the last one) against rank 0 alone in the same run.
        result['partitioned'] = {
            'encoder_speedup_vs_one_rank': {k: v.get('speedup_vs_one_rank') for k, v in scale.items() if isinstance(v, dict) and 'speedup_vs_one_rank' in v},
            'fb15k237_sharded_scoring_speedup_vs_one_rank': result.get('eval', {}).get('fb15k237', {}).get('speedup_vs_one_rank'),
            'note': 'strong scaling, same graph on all ranks, collectives included; the headline "value" is replicas',
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


def real_dataset_sections(pkg, args, dev, D, O):
    """if the public triple files are present on the box (data/<name>/{train,valid,test}.txt relative to
    the working directory, where the reference's loader looks: data_loader.py:57,65-66), run the same encoder step on the
    REAL graph as well. They are not shipped (no network): on a box without them this returns {}."""
    out = {}
    for name in ('WN18RR', 'FB15k-237'):
        d = os.path.join('data', name)
        if not all(os.path.isfile(os.path.join(d, s + '.txt')) for s in ('train', 'valid', 'test')):
            continue
        params = types.SimpleNamespace(gcn_in_dim=D, gcn_out_dim=O, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=10, k_h=20,
                                       num_filter=200, kernel_size=7, bias=False, lbl_smooth=0.1, gcn_layers=args.layers,
                                       cache_encoder=False, device=dev, ingest_only=True)
        dl = pkg.DataLoader(name, params)
        dl.graph.to(dev)
        torch.manual_seed(0)
        model = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params).to(dev).eval()
        steps = max(args.steps, 20)
        elapsed = timed_steps(model, dl.graph, steps, args.warmup, torch.cuda.synchronize, ramp_s=0.2)
        eps = args.layers * (2 * dl.num_edge + dl.num_entity)
        out[name] = {'N': dl.num_entity, 'R': dl.num_relation, 'E': dl.num_edge, 'ms_per_step': 1e3 * elapsed / steps,
                     'value': eps * steps / elapsed, 'unit': 'edges/s'}
        del model, dl
    return out


def guarded(fn):
    """Optional sections must not take the headline down with them: an exception becomes {'error': ...} (all ranks run
    the same code, so a failure is the same on every rank and no collective is left half-entered)."""
    try:
        return fn()
    except Exception as err:                                # noqa: BLE001 (reported in the JSON line)
        print('bench.py: optional section failed: %r' % (err,), file=sys.stderr)
        return {'error': repr(err)[:300]}


def encode_sharded_timing(pkg, model, graph, shape, dev, world, rank, dist, O):
    """SURVEY 8(e) on the wire: dist.encode_sharded — every rank computes the rows of its work-balanced destination range
    for both layers from its shard of the slot-ordered per-edge tables, and every layer output is all-gathered (RCCL) —
    timed on the SAME graph on all ranks (strong scaling), best of 5, max over ranks; rank 0 alone (a one-rank group, no
    collective) right after, in the same process."""
    N, R, E = shape['N'], shape['R'], shape['E']
    csr = graph.csr(2 * R + 1)
    b = csr.balanced_bounds(world)
    rp = csr.rowptr.cpu()
    work = [int((rp[0, b[r + 1]] - rp[0, b[r]]) + (rp[1, b[r + 1]] - rp[1, b[r]])) + (b[r + 1] - b[r]) for r in range(world)]
    out = {'world': world, 'layers': 1 + len(model.conv1_extra), 'rows_per_rank': [b[r + 1] - b[r] for r in range(world)],
           'edges_per_rank_excluding_hub_slots': work, 'allgather_bytes_per_layer': N * O * 4}

    def run(group):
        best = None
        for _ in range(5):
            model._enc_cache = None
            if dist is not None and group is None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pkg.dist.encode_sharded(model, graph, group=group)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
        return best

    t = run(None)
    if dist is not None:
        tt = torch.tensor([t], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    out['encode_sharded_s'] = t
    out['edges_per_s'] = out['layers'] * (2 * E + N) / t
    if world > 1:
        solo = dist.new_group([0])                          # (every rank takes part in creating it)
        if rank == 0:
            out['one_rank_s'] = run(solo)
            out['speedup_vs_one_rank'] = out['one_rank_s'] / t
        dist.barrier()
    else:
        out['one_rank_s'] = t
    return out


def config5_slice(pkg, dev, world, rank, dist, N=2000000, E=20000000, R=1000, D=512, O=512, parts=8):
    """One rank's 1/8 of BASELINE configs[4]'s layer through the PRODUCT path of a destination-partitioned rank (the rank's
    model holds ONLY its shard of the per-edge table: params.edge_table_rows + dist.shard_model_tables, rows from the
    chunk-wise xavier table), scaled to a graph one box builds in seconds: 2M entities, 20M triples, 1k relations,
    dim 512 -> 512. Rank r takes slice r % 8; nothing is cache-resident (10 GB of table rows per rank)."""
    part = rank % parts
    rng = np.random.default_rng(0)
    s, r, o = rng.integers(0, N, E), rng.integers(0, R, E), rng.integers(0, N, E)
    ei = torch.from_numpy(np.stack((np.concatenate((s, o)), np.concatenate((o, s)))))
    et = torch.from_numpy(np.concatenate((r, r + R)))
    del s, r, o
    csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, dev, with_backward=False)
    del ei, et
    b = csr.balanced_bounds(parts)
    n0, n1 = b[part], b[part + 1]
    rows = sum(csr.shard_slot_counts(n0, n1))
    params = types.SimpleNamespace(gcn_in_dim=D, gcn_out_dim=O, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=8, k_h=O // 8,
                                   num_filter=4, kernel_size=3, bias=False, lbl_smooth=0.1, gcn_layers=1, edge_table_rows=rows)
    torch.manual_seed(0)
    model = pkg.MGCN(N, R, E, params).to(dev).eval()
    pkg.dist.shard_model_tables(model, csr, n0, n1, lambda li, ids: pkg.dist.xavier_rows(ids, 2 * E, D, 11 + li, dev))
    layer, table = model.conv1, model.edge_embeddings.detach()
    x, rel = model.entity_embedding.detach(), model.relation_embedding.detach()
    out_rows = torch.empty((n1 - n0, O), device=dev)
    ee_sub = csr.shard_ee_sub(n0, n1)

    def run():
        with torch.no_grad():
            pkg.dist.encode_layer_rows(layer, csr, x, rel, table, n0, n1, ee_sub, out=out_rows)

    run()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    c.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(c) / 5
    # compulsory bytes of the rank's share: its per-edge rows + records, the x rows its slots gather (the 4 GB table is far
    # past every cache, so a gathered row is a DRAM access), its own x rows, its output rows
    gathered = rows * (4 * D + 4 * D + 16) + (n1 - n0) * (4 * D + 4 * O)
    # SURVEY 8(d)'s compulsory bytes of the same launch (every distinct byte once: the layer input counted ONCE, not per slot):
    # the contract's roofline fraction; the gathered-bytes fraction beside it says how fast real DRAM traffic moves
    bytes_8d = rows * (4 * D + 8) + 2 * (n1 - n0 + 1) * 4 + N * 4 * D + (2 * R + 1) * 4 * D + 16 * D * O + (n1 - n0) * 4 * O
    res = {'graph': 'N=%d E=%d R=%d dim %d->%d, slice %d of %d' % (N, E, R, D, O, part, parts), 'slots_this_rank': rows,
           'table_shard_GB': table.numel() * 4 / 1e9, 'whole_table_GB': 2 * E * D * 4 / 1e9, 'layer_ms': ms,
           'edges_per_s_per_rank': (rows + (n1 - n0)) / ms * 1e3,
           'bytes_8d_GB': bytes_8d / 1e9, 'frac_8d_of_8TBps': bytes_8d / ms / 1e6 / HBM_PEAK_GBS,
           'gathered_GBps': gathered / ms / 1e6, 'frac_gathered_of_8TBps': gathered / ms / 1e6 / HBM_PEAK_GBS,
           'full_size': 'profiles/r04_scale_shard_full.json: rank 0 of 8 of the TRUE 10M / 100M / dim-512 graph (tools/bench_scale_shard.py)',
           'path': 'fused' if pkg._native.fused_supported(D, O) else 'aggregate + dense'}
    del model, table, csr, out_rows
    torch.cuda.empty_cache()
    return res


def fb_sections(pkg, args, dev, world, rank, dist, barrier, D, O):
    """BASELINE configs[2] (the encoder step on the FB15k-237 shape, hub-heavy Zipf(1.1) tails: "denser graph,
    HBM-bound gather") and configs[3] (its full evaluation with the entity table sharded over the ranks)."""
    shape = SHAPES['fb15k237']
    N, R, E = shape['N'], shape['R'], shape['E']
    zipf = 1.1
    model, graph, params, edge_index, edge_attr = make_model(pkg, shape, dev, args.layers, 0, zipf)
    out = {}
    steps = max(args.steps, 20)
    elapsed = timed_steps(model, graph, steps, args.warmup, barrier, ramp_s=0.2)
    if rank == 0:
        eps = args.layers * (2 * E + N)
        step = {'workload': 'fb15k237-shape synthetic graph (N=%d, R=%d, E=%d, Zipf(%.1f) tails, hubs split), %d-layer '
                            'encoder %s' % (N, R, E, zipf, args.layers, '->'.join(map(str, [D] + [O] * args.layers))),
                'value': eps * steps / elapsed, 'unit': 'edges/s (this rank)', 'ms_per_step': 1e3 * elapsed / steps, 'steps': steps}
        step.update(kernel_breakdown(pkg, model, graph, steps, shape, 'fb15k237', args.layers, D, O))
        if not args.no_cpu_baseline and world == 1:
            step['cpu_baseline'] = cpu_baseline(model, edge_index, edge_attr, 'fb15k237', args.layers, N, R, E, D, O, 8.0,
                                                with_eval=False)
            step['gpu_over_cpu'] = step['value'] / step['cpu_baseline']['value']
        out['step'] = step
    else:
        out['step'] = None
    if not args.no_scale:
        out['scale'] = guarded(lambda: encode_sharded_timing(pkg, model, graph, shape, dev, world, rank, dist, O))
    if not args.no_eval:
        out['eval'] = fb_eval(pkg, model, graph, params, shape, dev, edge_index, edge_attr, world, rank, dist)
    return out


def fb_eval(pkg, model, graph, params, shape, dev, edge_index, edge_attr, world, rank, dist):
    """configs[3]: 2 x 20 466 queries in blocks of 128 against 14 541 entities; entity rows sharded over the W ranks
    (dist.evaluate_sharded: one all-gather of query embeddings / keys / objects, all-reduce of targets and int64 counts),
    with the encoder replicated and with it partitioned by destination (shard_encoder). `one_rank_*`: the same
    evaluation by rank 0 alone in the same process (a one-rank group, no collectives) — the N = 1 time to divide by."""
    N, R = shape['N'], shape['R']
    n_eval = shape['n_eval']
    rng = np.random.default_rng(7)
    s, r, o = rng.integers(0, N, 2 * n_eval), rng.integers(0, 2 * R, 2 * n_eval), rng.integers(0, N, 2 * n_eval)
    queries = torch.from_numpy(np.stack((s, r, o), axis=1))
    ei, et = edge_index.numpy(), edge_attr[0].numpy()
    half = len(et) // 2
    fwd = r < R                                                # a query with relation id r + R asks for the subject of (o, r, s)
    tri = np.concatenate([np.stack((ei[0][:half], et[:half], ei[1][:half]), axis=1),
                          np.stack((np.where(fwd, s, o), np.where(fwd, r, r - R), np.where(fwd, o, s)), axis=1)]).astype(np.int64)
    keys, ptr, tails = pkg._native.filter_index_build(torch.from_numpy(tri), R)   # both directions of every known triple
    filt = pkg.dist.FilterIndex(keys, ptr, tails, 2 * R).to(dev)
    scale_tables(model)
    params.cache_encoder = True
    out = {'queries': 2 * n_eval, 'batch': 'all queries in one count launch', 'world': world}
    solo = None
    if dist is not None:
        try:
            solo = dist.new_group([0])                           # (every rank must take part in creating it)
        except Exception as err:                                 # no one-rank reference then; the sharded numbers stand
            print('bench.py: dist.new_group([0]) failed: %s' % err, file=sys.stderr)

    def run(group, shard_encoder):
        best, res = None, None
        for _ in range(3):
            model._enc_cache = None
            if dist is not None and group is None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = pkg.dist.evaluate_sharded(model, graph, queries, filt, group=group, shard_encoder=shard_encoder)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
        return best, res

    for name, se in (('sharded_s', False), ('sharded_encoder_too_s', True)):
        t, res = run(None, se)
        if dist is not None:
            tt = torch.tensor([t], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        out[name] = t
        out[name.replace('_s', '_mrr')] = res['mrr']
    # where the evaluation's time goes (one extra run AFTER the timed ones — warm — with a synchronisation per part): the ConvE
    # trunk is stock torch (out of scope), the encoder + the target / filter / count kernels are the in-scope share
    parts = {}
    model._enc_cache = None
    pkg.dist.evaluate_sharded(model, graph, queries, filt, parts=parts)
    out['parts_s'] = {k: round(v, 6) for k, v in parts.items()}
    out['in_scope_share'] = round((parts.get('encoder_s', 0.0) + parts.get('kernels_s', 0.0)) / max(sum(parts.values()), 1e-12), 4)
    if world > 1:
        if rank == 0 and solo is not None:
            t1, res1 = run(solo, False)
            out['one_rank_s'] = t1
            out['one_rank_mrr'] = res1['mrr']
            out['speedup_vs_one_rank'] = t1 / out['sharded_s']
            # integer counts are summed exactly, so for the SAME query embeddings the sharded ranks are the one-rank ranks
            # (tests/test_gpu_parity.py::test_evaluate_sharded_two_ranks_one_gpu); here each rank runs the stock-torch
            # ConvE trunk on its own slice of the queries, whose last chunk has another size than the one-rank run's, and
            # torch's f32 results depend on the batch size in the last bits: the MRRs agree to north_star's 1e-4, not bitwise
            assert abs(res1['mrr'] - out['sharded_mrr']) <= 1e-4, (res1['mrr'], out['sharded_mrr'])
        dist.barrier()
    else:
        out['one_rank_s'], out['one_rank_mrr'] = out['sharded_s'], out['sharded_mrr']
    params.cache_encoder = False
    model._enc_cache = None
    return out


def source_fingerprint():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, 'kgc-gcn_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def kernel_breakdown(pkg, model, graph, K, shape, shape_name, n_layers, D, O):
    """Per-launch device time of the step's kernels, measured IN SEQUENCE (the same launches, order and
    operands as model.encode, so caches hold what they hold in the real step) with HIP events recorded on the
    launch stream between the kernels, over K steps; each launch priced against both roofs."""
    nat = pkg._native
    N, R, E = shape['N'], shape['R'], shape['E']
    csr = graph.csr(2 * R + 1)
    dev = model.entity_embedding.device
    layers = [model.conv1] + list(model.conv1_extra)
    tables = [model.edge_embeddings] + list(model.edge_embeddings_extra)
    fused = all(nat.fused_supported(l.in_channels, l.out_channels) for l in layers)
    bufs = [(torch.empty((N, 3 * l.in_channels), device=dev), torch.empty((N, O), device=dev), l.derived_weights()) for l in layers]
    rel_outs = [torch.empty((2 * R, O), device=dev) for _ in layers]

    def sequence(events):
        x, rel = model.entity_embedding, model.relation_embedding
        i = 0
        for li, (layer, table, (agg, out, (wcat, wpack))) in enumerate(zip(layers, tables, bufs)):
            bn = layer.ent_bn
            events[i].record(); i += 1
            if fused:     # one launch: the layer and the relation projection (model.py:107)
                nat.layer_fwd_fused(csr, x, rel, layer.loop_rel.reshape(-1), table, True, layer.loop_edge.reshape(-1), wpack,
                                    O, layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, out,
                                    rels_weight=layer.rels_weight.detach(), rel_out=rel_outs[li])
                events[i].record(); i += 1
                events[i].record(); i += 1
                rel, x = rel_outs[li], out
                continue
            nat.aggregate_fwd(csr, x, rel, table, True, layer.loop_edge.reshape(-1), agg, loop_rel=layer.loop_rel.reshape(-1))
            events[i].record(); i += 1
            nat.dense_bn_tanh_fwd(agg, wcat, layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, out)
            events[i].record(); i += 1
            rel = nat.matmul(rel, layer.rels_weight)
            x = out
        events[i].record()

    nev = 3 * len(layers) + 1
    with torch.no_grad():
        for _ in range(10):
            sequence([torch.cuda.Event(enable_timing=True) for _ in range(nev)])
        torch.cuda.synchronize()
        all_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nev)] for _ in range(K)]
        for ev in all_ev:
            sequence(ev)
        torch.cuda.synchronize()
    times = {}
    for li in range(len(layers)):
        for j, kind in enumerate(('aggregate', 'dense', 'relproj')):
            idx = 3 * li + j
            times['%s_l%d' % (kind, li + 1)] = float(np.mean([ev[idx].elapsed_time(ev[idx + 1]) for ev in all_ev])) * 1e3
    dims = [D] + [O] * (n_layers - 1)
    kern = {}
    for li, d in enumerate(dims):
        fl = 2.0 * N * 3 * d * O
        lb = layer_bytes(N, 2 * E, 2 * R, d, O)
        ta, td = times['aggregate_l%d' % (li + 1)], times['dense_l%d' % (li + 1)]
        if fused:       # the first slot holds the one fused launch, the second is empty
            sf = split_mfma_flops(N, d, O)
            kern['layer_fused_l%d' % (li + 1)] = {
                'us': ta, 'algorithmic_bytes': lb, 'GBps': lb / ta / 1e3, 'hbm_frac': lb / ta / 1e3 / HBM_PEAK_GBS,
                'algorithmic_flops': fl, 'TFLOPs': fl / ta / 1e6, 'mfma_f32_frac': fl / ta / 1e6 / MFMA_F32_PEAK_TFLOPS,
                'issued_bf16_flops': sf, 'mfma_bf16_frac': sf / ta / 1e6 / MFMA_BF16_PEAK_TFLOPS}
        else:
            ab = agg_kernel_bytes(N, 2 * E, 2 * R, d)
            kern['aggregate_l%d' % (li + 1)] = {'us': ta, 'algorithmic_bytes': ab, 'GBps': ab / ta / 1e3,
                                                'hbm_frac': ab / ta / 1e3 / HBM_PEAK_GBS}
            kern['dense_l%d' % (li + 1)] = {'us': td, 'algorithmic_flops': fl, 'TFLOPs': fl / td / 1e6,
                                            'mfma_f32_frac': fl / td / 1e6 / MFMA_F32_PEAK_TFLOPS}
            kern['layer%d' % (li + 1)] = {'us': ta + td, 'algorithmic_bytes': lb,
                                          'hbm_frac': lb / (ta + td) / 1e3 / HBM_PEAK_GBS}
            kern['relproj_l%d' % (li + 1)] = {'us': times['relproj_l%d' % (li + 1)]}
    if fused:
        # the HBM-bound part on its own: the aggregation launch of the unfused path (what training's forward and
        # shapes outside the fused kernel run) on the same operands, the layers alternating as in a real step
        with torch.no_grad():
            rels = [model.relation_embedding]
            for layer in layers[:-1]:
                rels.append(nat.matmul(rels[-1], layer.rels_weight))
            xs = [model.entity_embedding] + [b[1] for b in bufs[:-1]]

            def agg_sequence(events):
                for li, (layer, table, (agg, out, _)) in enumerate(zip(layers, tables, bufs)):
                    events[2 * li].record()
                    nat.aggregate_fwd(csr, xs[li], rels[li], table, True, layer.loop_edge.reshape(-1), agg,
                                      loop_rel=layer.loop_rel.reshape(-1))
                    events[2 * li + 1].record()

            mk = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(2 * len(layers))]
            agg_sequence(mk())
            torch.cuda.synchronize()
            evs = [mk() for _ in range(K)]
            for ev in evs:
                agg_sequence(ev)
            torch.cuda.synchronize()
            for li in range(len(layers)):
                us = float(np.mean([ev[2 * li].elapsed_time(ev[2 * li + 1]) for ev in evs])) * 1e3
                ab = agg_kernel_bytes(N, 2 * E, 2 * R, dims[li])
                kern['aggregate_only_l%d' % (li + 1)] = {'us': us, 'algorithmic_bytes': ab, 'GBps': ab / us / 1e3,
                                                         'hbm_frac': ab / us / 1e3 / HBM_PEAK_GBS,
                                                         'note': 'agg_fwd_kernel alone (not part of the timed step)'}
    cand = {k: v for k, v in kern.items() if not k.startswith(('relproj', 'layer1', 'layer2', 'aggregate_only'))}
    dom = max(cand, key=lambda k: cand[k]['us'])
    k = kern[dom]
    if dom.startswith('layer_fused'):
        # north_star's roof for this path: HBM (SURVEY §8d algorithmic bytes / launch time / 8 TB/s). The dense step rides
        # in the same launch: its algorithmic flops against the exact-f32 MFMA peak (what round 1 paid) and the flops
        # actually issued (six bf16 products) against the bf16 peak are reported beside it, for BOTH layers.
        gen = nat.tune_generation() or nat.lib().mgcn_fused_kernel_generation(dims[int(dom[-1]) - 1], O, N, 1)
        roof = {'kernel': 'layer_fused%d_kernel (%s)' % (gen, dom), 'bound': 'hbm', 'achieved': k['GBps'], 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': k['hbm_frac'], 'traffic': None,
                'per_layer': {n: {'us': v['us'], 'hbm_frac': v['hbm_frac'], 'mfma_f32_frac': v['mfma_f32_frac'],
                                  'mfma_bf16_frac': v['mfma_bf16_frac']}
                              for n, v in kern.items() if n.startswith('layer_fused')}}
    elif dom.startswith('aggregate'):
        roof = {'kernel': 'agg_fwd_kernel (%s)' % dom, 'bound': 'hbm', 'achieved': k['GBps'], 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': k['hbm_frac'], 'traffic': None}
    else:
        roof = {'kernel': 'tile_kernel<BN_TANH> (%s)' % dom, 'bound': 'mfma', 'achieved': k['TFLOPs'],
                'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': k['mfma_f32_frac'], 'traffic': None}
    # HBM-side bytes of the same launch from the PMC passes (tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    # in separate passes, corrected as MI355X_MICROARCH.md prescribes), committed under profiles/. The file carries the
    # fingerprint of the kernel sources it was measured on: a stale file is NOT quoted.
    try:
        with open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE)) as f:
            tj = json.load(f)
        traffic = tj.get(shape_name, {})
        if tj.get('source_fingerprint') != source_fingerprint():
            roof['traffic_note'] = 'profiles/%s was measured on other kernel sources (%s != %s): not quoted' % (
                TRAFFIC_FILE, tj.get('source_fingerprint'), source_fingerprint())
            print('bench.py: ' + roof['traffic_note'], file=sys.stderr)
        elif dom in traffic:
            roof['traffic'] = traffic[dom]['traffic_bytes']
            for n in roof.get('per_layer', {}):
                if n in traffic:
                    roof['per_layer'][n]['traffic'] = traffic[n]['traffic_bytes']
    except (OSError, KeyError, ValueError) as err:
        roof['traffic_note'] = 'no PMC traffic file: %s' % err
    return {'roofline': roof, 'kernels': kern}


def train_breakdown(pkg, model, graph, shape, D, O):
    """the aggregation backward (autograd through model.py:99-101, 111-118; main.py:66) on its own kernels, timed
    with HIP events on the launch stream: mgcn_aggregate_bwd = agg_bwd_gee (per-slot gradient of the per-edge table, streamed),
    agg_bwd_gx (+ hub pre-pass; by-source sums through the mirror map) and the by-type reduction for the relation table, for
    both layer widths of the benchmark; plus the training-mode forward aggregation (mgcn_aggregate_fwd, what `.train()` runs).
    Algorithmic bytes, every distinct byte once: g [N, 2D] + x [N, D] + per-edge table [2E, D] read, gx [N, D] + gee [2E, D]
    written, 16-byte records + mirror + slot_dst per slot: 16 N D + 16 E D + 48 E."""
    nat = pkg._native
    N, R, E = shape['N'], shape['R'], shape['E']
    dev = model.entity_embedding.device
    csr = graph.csr(2 * R + 1)
    if not csr.has_backward:
        return {'error': 'graph prepared without the backward indices'}
    out = {}
    g = torch.Generator().manual_seed(5)
    for name, d in (('layer1_D%d' % D, D), ('layer2_D%d' % O, O)):
        x = (torch.randn(N, d, generator=g) * 0.3).to(dev)
        rel = (torch.randn(2 * R + 1, d, generator=g) * 0.5).to(dev)
        ee = (torch.randn(2 * E, d, generator=g) * 0.5).to(dev)
        grad = (torch.randn(N, 3 * d, generator=g) * 0.1).to(dev)
        agg = torch.empty((N, 3 * d), device=dev)
        le = torch.ones(d, device=dev)
        K = 20
        res = {}
        for what, fn, nbytes in (
                ('aggregate_bwd', lambda: nat.aggregate_bwd(csr, x, rel, ee, grad), 16 * N * d + 16 * E * d + 48 * E),
                ('aggregate_fwd_train', lambda: nat.aggregate_fwd(csr, x, rel, ee, True, le, agg), agg_kernel_bytes(N, 2 * E, 2 * R, d))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(K):
                fn()
            c.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(c) * 1e3 / K
            res[what] = {'us': us, 'algorithmic_bytes': nbytes, 'GBps': nbytes / us / 1e3, 'hbm_frac': nbytes / us / 1e3 / HBM_PEAK_GBS}
        out[name] = res
        del x, rel, ee, grad, agg
    out['note'] = ('device time of the launches of one call (events on the launch stream, %d calls back to back: caches warm); '
                   'aggregate_bwd = gee + gx (+ hub pre-pass) + relation-table reduction, no float atomics' % K)
    return out


def scale_tables(model):
    """xavier tables of these sizes are tiny (|x| ~ 1e-2): an untrained model then scores every entity 0.5 +- 1e-6 and a rank
    check decides nothing. Scaled as tests/test_gpu_bench_shapes.py scales them, layer outputs are O(0.1 - 1) and the
    scores of one query spread over (0.05, 0.95): ranks have margins, so "same ranks as the reference order" has teeth.
    (In place, once, after the timed encoder step: kernel times do not depend on the values.)"""
    if getattr(model, '_bench_scaled', False):
        return
    with torch.no_grad():
        model.entity_embedding.mul_(30.0)
        model.relation_embedding.mul_(3.0)
        for t in [model.edge_embeddings] + list(model.edge_embeddings_extra):
            t.mul_(100.0)
    model._bench_scaled = True
    model._enc_cache = None


def eval_wallclock(pkg, model, graph, params, shape, dev, edge_index, edge_attr, world, rank):
    """Full filtered-MRR evaluation wall-clock: 2 x n_eval queries (tail + head side) in batches of 128 against all
    N entities. Three forms of the same computation, each timed on its second run:
      sharded_bits_s    dist.evaluate_sharded as it runs by default: encoder once (eval cache), filter bits built on the
                        device, ONE HIP score+filter+count launch over all of the rank's queries, entity table row-sharded
                        over the ranks (RCCL exchange if W > 1); sharded_bits_blocks128_s = the same in the reference's
                        blocks of 128 queries (49 count launches);
      fused_dense_s     (rank 0 only) HIP kernel fed by dense [B, N] label blocks already resident on the device;
      reference_order_s (rank 0 only) what main.py:117-126 does: encoder per batch, [B, N] scores, double argsort.
    The sharded form runs the ConvE trunk (stock torch, out of scope) in chunks of 2048 queries, the other two in
    batches of 128: its f32 query embeddings differ in the last bits, so its MRR may differ from theirs in the 5th
    significant digit (tests/test_gpu_bench_shapes.py pins the cause); asserted here within north_star's 1e-4."""
    N, R = shape['N'], shape['R']
    n_eval = min(shape['n_eval'], 4096)
    rng = np.random.default_rng(7)
    B = 128
    s = rng.integers(0, N, 2 * n_eval)
    r = rng.integers(0, 2 * R, 2 * n_eval)
    o = rng.integers(0, N, 2 * n_eval)
    queries = torch.from_numpy(np.stack((s, r, o), axis=1))
    known = {}
    ei, et = edge_index.numpy(), edge_attr[0].numpy()
    for a, t, bb in zip(ei[0], et, ei[1]):                   # every training edge (both directions) is a known answer
        known.setdefault((int(a), int(t)), set()).add(int(bb))
    for a, t, bb in zip(s, r, o):
        known.setdefault((int(a), int(t)), set()).add(int(bb))
    filt = pkg.dist.FilterIndex.from_known(known, 2 * R).to(dev)
    out = {'queries': 2 * n_eval, 'batch': B, 'world': world}
    scale_tables(model)
    params.cache_encoder = True
    for name, bs in (('sharded_bits_s', None), ('sharded_bits_blocks128_s', B)):
        for _ in range(2):
            model._enc_cache = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = pkg.dist.evaluate_sharded(model, graph, queries, filt, batch_size=bs)
            torch.cuda.synchronize()
            out[name] = time.perf_counter() - t0
        out[name.replace('_s', '_mrr')] = res['mrr']
    parts = {}                              # where the default evaluation's time goes (one more run, a synchronisation per part)
    model._enc_cache = None
    pkg.dist.evaluate_sharded(model, graph, queries, filt, parts=parts)
    out['parts_s'] = {k: round(v, 6) for k, v in parts.items()}
    out['in_scope_share'] = round((parts.get('encoder_s', 0.0) + parts.get('kernels_s', 0.0)) / max(sum(parts.values()), 1e-12), 4)
    if rank == 0:
        batches = []
        for i in range(0, 2 * n_eval, B):
            q = queries[i:i + B].to(dev)
            keys = filt.query_keys(q[:, 0], q[:, 1])
            mask = pkg._native.filter_mask(keys, filt.keys, filt.ptr, filt.tails, N)
            bits = (mask.view(torch.int32)[:, :, None] >> torch.arange(32, device=dev, dtype=torch.int32)) & 1
            lab = bits.reshape(q.size(0), -1)[:, :N].float().contiguous()
            batches.append((q, lab))
        for name, fused, cache in (('fused_dense_s', True, True), ('reference_order_s', False, False)):
            params.cache_encoder = cache
            for _ in range(2):
                model._enc_cache = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.no_grad():
                    acc = torch.zeros((), dtype=torch.float64, device=dev)
                    all_ranks = []
                    for q, lab in batches:
                        if fused:
                            counts, _ = model.rank_counts(q[:, 0], q[:, 1], q[:, 2].contiguous(), lab, graph)
                            ranks = 1 + counts[:, 0] + counts[:, 1]
                        else:
                            ranks = pkg.harness.ranks_from_scores(model(q[:, 0], q[:, 1], graph), lab, q[:, 2])
                        acc += (1.0 / ranks.double()).sum()
                        all_ranks.append(ranks.to(torch.int64))
                    mrr = float(acc.item()) / (2 * n_eval)
                torch.cuda.synchronize()
                out[name] = time.perf_counter() - t0
            out[name.replace('_s', '_mrr')] = mrr
            out[name.replace('_s', '_ranks')] = torch.cat(all_ranks)
        # parity with teeth: per-query ranks of the HIP count path against the reference's own order of operations
        # (main.py:117-126: scores, mask, double argsort) on a model whose scores have margins (scale_tables)
        ours, ref = out.pop('fused_dense_ranks'), out.pop('reference_order_ranks')
        out['rank_rows'] = int(ours.numel())
        out['rank_rows_equal_reference_order'] = int((ours == ref).sum())
        out['mean_rank_reference_order'] = float(ref.double().mean())
        assert out['rank_rows_equal_reference_order'] >= 0.98 * out['rank_rows'], out
        assert abs(out['sharded_bits_mrr'] - out['reference_order_mrr']) <= 1e-4, out
        assert abs(out['fused_dense_mrr'] - out['reference_order_mrr']) <= 1e-4, out
    params.cache_encoder = False
    model._enc_cache = None
    return out


def host_cores():
    """Cores this process may actually use: cgroup quota, then affinity, then cpu_count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get('MGCN_CPU_THREADS', min(n, 16)))   # a 1-GPU box's CPU share is 16


def cpu_baseline(model, edge_index, edge_attr, shape_name, n_layers, N, R, E, D, O, budget_s, with_eval=True):
    """The oracle (CPU restatement in the reference's order of operations: per-edge weight multiply, identity
    gathers, norms recomputed per call) on this box's host cores, same graph and parameters, bounded sample; one
    forward also on ONE thread (BASELINE.md §2's single-thread figure)."""
    oracle = importlib.import_module('oracle.mgcn_oracle')
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cores = host_cores()
    torch.set_num_threads(cores)
    prefixes = ['conv1.'] + ['conv1_extra.%d.' % i for i in range(n_layers - 1)]
    tables = ['edge_embeddings'] + ['edge_embeddings_extra.%d' % i for i in range(n_layers - 1)]

    def cpu_step():
        x, rel = sd['entity_embedding'].index_select(0, torch.arange(N)), sd['relation_embedding']
        for pre, tab in zip(prefixes, tables):
            ee = sd[tab].index_select(0, edge_attr[1])
            x, rel = oracle.layer_forward(sd, pre, x, edge_index, edge_attr[0], ee, rel)
        return x

    with torch.no_grad():
        cpu_step()
        times = []
        t_start = time.perf_counter()
        while len(times) < 20 and (time.perf_counter() - t_start < budget_s or len(times) < 3):
            t0 = time.perf_counter()
            cpu_step()
            times.append(time.perf_counter() - t0)
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        cpu_step()
        one_thread = time.perf_counter() - t0
        torch.set_num_threads(cores)
    best = min(times)
    res = {'value': n_layers * (2 * E + N) / best, 'unit': 'edges/s', 'cores': cores, 'kind': 'port',
           'sample': '%d full-graph %d-layer forwards of the same workload (min of %d, %.1f ms each)'
                     % (len(times), n_layers, len(times), best * 1e3),
           'one_thread_edges_per_s': n_layers * (2 * E + N) / one_thread}
    if not with_eval:
        return res
    # the evaluation in the reference's order (main.py:117-126: encoder per batch, [B, N] scores, double argsort) on
    # two batches of 128 queries, extrapolated to the benchmark's 2 x n_eval queries
    hp = {'gcn_out_dim': O, 'k_w': 10, 'k_h': 20}
    g = torch.Generator().manual_seed(7)
    B = 128
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(2):
            all_ent = cpu_step()
            rel = sd['relation_embedding']
            for pre in prefixes:
                rel = torch.matmul(torch.cat([rel, sd[pre + 'loop_rel']], dim=0), sd[pre + 'rels_weight'])[:-1]
            sub, rid, obj = (torch.randint(0, N, (B,), generator=g), torch.randint(0, 2 * R, (B,), generator=g),
                             torch.randint(0, N, (B,), generator=g))
            x = oracle.conve_trunk(sd, hp, all_ent.index_select(0, sub), rel.index_select(0, rid))
            pred = oracle.score_all(x, all_ent, sd['conv2.bias'])
            oracle.filtered_rank(pred, torch.zeros_like(pred), obj)
        per_batch = (time.perf_counter() - t0) / 2
    n_batches = -(-2 * min(SHAPES[shape_name]['n_eval'], 4096) // B)
    res['eval_reference_order_s'] = per_batch * n_batches
    res['eval_sample'] = '2 batches of %d queries in the reference order (%.2f s each), x %d batches (extrapolated)' % (
        B, per_batch, n_batches)
    return res


if __name__ == '__main__':
    main()
