"""A/B of the fused layer kernels on the bench workload (not product code): output difference against the two-launch
path (exact-f32 MFMA) and per-layer launch time. `MGCN_FUSED_V1=1 python tools/ab_fused2.py` times the first
generation; without it the second. Usage: python tools/ab_fused2.py [wn18rr|fb15k237] [zipf]"""
import importlib
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    shape_name = sys.argv[1] if len(sys.argv) > 1 else 'wn18rr'
    zipf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    dims = [int(v) for v in os.environ.get('AB_DIMS', '100,200,200').split(',')]
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
    res = {'shape': shape_name, 'zipf': zipf, 'v1': os.environ.get('MGCN_FUSED_V1', '0'), 'layers': []}
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
        wcat, wpack = layer.derived_weights()
        agg = torch.empty((N, 3 * D), device=dev)
        ref = torch.empty((N, O), device=dev)
        out = torch.empty((N, O), device=dev)
        rel_out = torch.empty((2 * R, O), device=dev)
        lr, le = layer.loop_rel.reshape(-1), layer.loop_edge.reshape(-1)
        with torch.no_grad():
            nat.aggregate_fwd(csr, x, rel, ee, True, le, agg, loop_rel=lr)
            nat.dense_bn_tanh_fwd(agg, wcat, layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, ref)
            rel_ref = nat.matmul(rel, layer.rels_weight)

            def run():
                nat.layer_fwd_fused(csr, x, rel, lr, ee, True, le, wpack, O, layer.bias, bn.running_mean, bn.running_var,
                                    bn.weight, bn.bias, bn.eps, out, rels_weight=layer.rels_weight.detach(), rel_out=rel_out)
            run()
            torch.cuda.synchronize()
            err = float((out - ref).abs().max())
            # exact reference in float64 on the same aggregate (what f32 itself loses)
            ref64 = torch.tanh((((agg.double() @ wcat.double()) / 3 - bn.running_mean.double())
                                / torch.sqrt(bn.running_var.double() + bn.eps)) * bn.weight.double() + bn.bias.double())
            e64_fused = float((out.double() - ref64).abs().max())
            e64_f32 = float((ref.double() - ref64).abs().max())
            rel_equal = bool(torch.equal(rel_out, rel_ref))
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
        with torch.no_grad():
            def run2():
                nat.aggregate_fwd(csr, x, rel, ee, True, le, agg, loop_rel=lr)
                nat.dense_bn_tanh_fwd(agg, wcat, layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, ref)
                nat.matmul(rel, layer.rels_weight)
            for _ in range(10):
                run2()
            torch.cuda.synchronize()
            a2, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a2.record()
            for _ in range(K):
                run2()
            b2.record()
            torch.cuda.synchronize()
        res['layers'].append({'D': D, 'O': O, 'us': 1e3 * a.elapsed_time(b) / K, 'two_launch_us': 1e3 * a2.elapsed_time(b2) / K,
                              'max_abs_vs_two_launch': err,
                              'max_abs_vs_f64': e64_fused, 'two_launch_vs_f64': e64_f32, 'rel_bit_equal': rel_equal})
        x, rel = ref, rel_ref
    print(json.dumps(res))


if __name__ == '__main__':
    main()
