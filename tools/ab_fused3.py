"""A/B of the fused layer kernel's geometries on the bench workload (not product code): per layer, for every `tune`
word given, the launch time, the output difference against the two-launch path (exact-f32 MFMA) and against float64,
and whether a destination-range launch reproduces the rows bit for bit.
Usage: python tools/ab_fused3.py [wn18rr|fb15k237] [zipf] ; AB_TUNES=0xc00,0xc25,... AB_DIMS=100,200,200"""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    shape_name = sys.argv[1] if len(sys.argv) > 1 else 'wn18rr'
    zipf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    dims = [int(v) for v in os.environ.get('AB_DIMS', '100,200,200').split(',')]
    tunes = [int(v, 0) for v in os.environ.get('AB_TUNES', '0x800,0xc00').split(',')]
    pkg = importlib.import_module('kgc-gcn_amd')
    nat = pkg._native
    dev = torch.device('cuda:0')
    shape = bench.SHAPES[shape_name]
    N, R, E = shape['N'], shape['R'], shape['E']
    ei, ea = bench.synth_graph(shape, seed=0, zipf=zipf)
    graph = pkg.Graph(edge_index=ei, edge_attr=ea)
    graph.entity, graph.num_nodes, graph.edge_norm = torch.arange(N), N, None
    graph.to(dev)
    csr = graph.csr(2 * R + 1)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(N, dims[0], generator=g) * 0.3).to(dev)
    rel = (torch.randn(2 * R, dims[0], generator=g) * 0.5).to(dev)
    res = {'shape': shape_name, 'zipf': zipf, 'layers': []}
    cold_src = torch.empty(40 << 20, device=dev)          # 160 MB read + 160 MB written between two timed launches
    cold_dst = torch.empty_like(cold_src)
    for li in range(len(dims) - 1):
        D, O = dims[li], dims[li + 1]
        layer = pkg.model.MGCNConv(D, O, 2 * R).to(dev).eval()
        with torch.no_grad():
            layer.ent_bn.running_mean.copy_(torch.randn(O, generator=g) * 0.05)
            layer.ent_bn.running_var.copy_(torch.rand(O, generator=g) * 0.5 + 0.05)
            layer.ent_bn.weight.copy_(torch.rand(O, generator=g) + 0.5)
            layer.ent_bn.bias.copy_(torch.randn(O, generator=g) * 0.1)
        ee = (torch.randn(2 * E, D, generator=g) * 0.5).to(dev)
        bn = layer.ent_bn
        wcat, _ = layer.derived_weights()
        agg = torch.empty((N, 3 * D), device=dev)
        ref = torch.empty((N, O), device=dev)
        lr, le = layer.loop_rel.reshape(-1), layer.loop_edge.reshape(-1)
        entry = {'D': D, 'O': O, 'tunes': {}}
        with torch.no_grad():
            nat.aggregate_fwd(csr, x, rel, ee, True, le, agg, loop_rel=lr)
            nat.dense_bn_tanh_fwd(agg, wcat, layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, ref)
            rel_ref = nat.matmul(rel, layer.rels_weight)
            ref64 = torch.tanh((((agg.double() @ wcat.double()) / 3 - bn.running_mean.double())
                                / torch.sqrt(bn.running_var.double() + bn.eps)) * bn.weight.double() + bn.bias.double())
            entry['two_launch_vs_f64'] = float((ref.double() - ref64).abs().max())
            n0, n1 = N // 3 + 5, (2 * N) // 3 + 11
            for tune in tunes:
                nat.FUSED_TUNE = tune
                wpack = nat.pack_weights(wcat)
                ldo = int(os.environ.get('AB_LDO', O))          # padded output rows (row stride in floats)
                out = torch.full((N, ldo), float('nan'), device=dev)[:, :O]
                rel_out = torch.empty((2 * R, O), device=dev)

                def run():
                    nat.layer_fwd_fused(csr, x, rel, lr, ee, True, le, wpack, O, layer.bias, bn.running_mean, bn.running_var,
                                        bn.weight, bn.bias, bn.eps, out, rels_weight=layer.rels_weight.detach(), rel_out=rel_out,
                                        tune=tune)
                try:
                    run()
                    torch.cuda.synchronize()
                except nat.NativeError as e:
                    entry['tunes'][hex(tune)] = {'error': str(e)[:120]}
                    continue
                part = torch.empty((n1 - n0, O), device=dev)
                nat.layer_fwd_fused(csr, x, rel, lr, csr.edge_table_shard(ee, n0, n1), True, le, wpack, O, layer.bias,
                                    bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, part, node_range=(n0, n1),
                                    ee_sub=csr.shard_ee_sub(n0, n1), tune=tune)
                torch.cuda.synchronize()
                r = {'max_abs_vs_two_launch': float((out - ref).abs().max()),
                     'max_abs_vs_f64': float((out.double() - ref64).abs().max()),
                     'rel_bit_equal': bool(torch.equal(rel_out, rel_ref)),
                     'range_bit_equal': bool(torch.equal(part, out[n0:n1]))}
                for _ in range(20):
                    run()
                torch.cuda.synchronize()
                K = 200
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(K):
                    run()
                b.record()
                torch.cuda.synchronize()
                r['us'] = round(1e3 * a.elapsed_time(b) / K, 2)
                if os.environ.get('AB_COLD'):     # as in the real step: the other layer's 120-200 MB pass through the caches in between
                    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
                    keep_input = bool(os.environ.get('AB_COLD_KEEP_INPUT'))   # as in the real step: the layer input was just written
                    for e0, e1 in evs:
                        cold_dst.copy_(cold_src)
                        if keep_input:
                            x.add_(0.0)                                          # (read + rewrite: resident again, like a fresh layer output)
                        e0.record()
                        run()
                        e1.record()
                    torch.cuda.synchronize()
                    r['us_cold'] = round(1e3 * sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs), 2)
                if hasattr(nat.lib(), 'mgcn_diag_fused3') and (tune >> 10) & 3 == 3:   # diagnostics build: who waits for whom
                    import ctypes
                    import numpy as np
                    buf = np.zeros((1024, 16, 8), dtype=np.uint64)
                    if nat.lib().mgcn_diag_fused3(ctypes.c_void_p(buf.ctypes.data)) == 0:
                        used = buf[:, 0, 0] > 0
                        d = buf[used].astype(np.float64) / 1e3
                        names = ['total', 'ring_wait', 'load_wait', 'batches', 'a', 'b', 'd', 'o']
                        r['diag'] = {'wgs': int(used.sum()),
                                     'multiply(total,ring+phase waits,-,-,convert,conv_wait,mfma,other)': [round(float(d[:, :8, i].mean()), 1) for i in range(8)],
                                     'gather(total,ring_wait,load_wait,kbatches,issue,prefetch_next,consume,other)': [round(float(d[:, 8:, i].mean()), 1) for i in range(8)],
                                     'gather_total_max': round(float(d[:, 8:, 0].max()), 1), 'multiply_total_max': round(float(d[:, :8, 0].max()), 1)}
                        if os.environ.get('AB_DUMP_WG'):       # per-workgroup totals (kcycles): gather waves' mean / slowest, load wait
                            r['diag']['per_wg_gather_mean'] = [round(float(v), 1) for v in d[:, 8:, 0].mean(1)]
                            r['diag']['per_wg_gather_max'] = [round(float(v), 1) for v in d[:, 8:, 0].max(1)]
                            r['diag']['per_wg_load_wait'] = [round(float(v), 1) for v in d[:, 8:, 2].mean(1)]
                entry['tunes'][hex(tune)] = r
        res['layers'].append(entry)
        x, rel = ref, rel_ref
        print(json.dumps(entry), flush=True)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
