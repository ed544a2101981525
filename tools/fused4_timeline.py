"""Phase timeline of the generation-4 fused layer kernel from the diagnostics build's s_memtime stamps (not product code).
MGCN_LIB=.../libmgcn_hip_diag.so python tools/fused4_timeline.py [layer: 0|1] [tune ...]"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    tunes = [int(v, 0) for v in sys.argv[2:]] or [0x400]
    pkg = importlib.import_module('kgc-gcn_amd')
    nat = pkg._native
    dev = torch.device('cuda:0')
    shape = bench.SHAPES['wn18rr']
    N, R, E = shape['N'], shape['R'], shape['E']
    ei, ea = bench.synth_graph(shape, seed=0)
    graph = pkg.Graph(edge_index=ei, edge_attr=ea)
    graph.entity, graph.num_nodes, graph.edge_norm = torch.arange(N), N, None
    graph.to(dev)
    csr = graph.csr(2 * R + 1)
    D, O = (100, 200) if which == 0 else (200, 200)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(N, D, generator=g) * 0.3).to(dev)
    rel = (torch.randn(2 * R, D, generator=g) * 0.5).to(dev)
    layer = pkg.model.MGCNConv(D, O, 2 * R).to(dev).eval()
    ee = (torch.randn(2 * E, D, generator=g) * 0.5).to(dev)
    bn = layer.ent_bn
    wcat, _ = layer.derived_weights()
    out = torch.empty((N, O), device=dev)
    rel_out = torch.empty((2 * R, O), device=dev)
    fn = nat.lib().mgcn_diag_fused4
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    nst = 1 if 3 * D <= 320 else -(-(-(-3 * D // 32)) // 10)
    for tune in tunes:
        wpack = nat.pack_weights(wcat, generation=4)
        with torch.no_grad():
            for _ in range(30):
                nat.layer_fwd_fused(csr, x, rel, layer.loop_rel.reshape(-1), ee, True, layer.loop_edge.reshape(-1), wpack, O,
                                    layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, out,
                                    rels_weight=layer.rels_weight.detach(), rel_out=rel_out, tune=tune)
            torch.cuda.synchronize()
        buf = np.zeros((1024, 16, 64), dtype=np.uint64)
        assert fn(buf.ctypes.data) == 0
        st = buf[:256].astype(np.int64)                      # [wg][wave][stamp]
        t0 = st[:, :, 0].min()
        print('tune %#x: launch span %d cycles (first entry -> last exit), entry skew %d' % (tune, st[:, :, 63].max() - t0, st[:, :, 0].max() - t0))
        names = ['G start', 'self loop done', 'in-half done', 'out-half done', 'B1 passed', 'k-loop done']
        rows = []
        nphase = 0
        while 2 + 8 * nphase < 58 and st[0, 0, 2 + 8 * nphase] > 0:
            nphase += 1
        print('  phases (tile, stage) per workgroup: %d' % nphase)
        prev = st[:, :, 1]
        print('  init (entry -> first barrier passed): mean %6d' % (st[:, :, 1] - st[:, :, 0]).mean())
        for ph in range(nphase):
            b = 2 + 8 * ph
            seg = []
            last = prev
            for i in range(8):
                cur = st[:, :, b + i]
                cur = np.where(cur > 0, cur, last)           # (a stamp a wave did not take: segment absent)
                seg.append((cur - last))
                last = cur
            prev = last
            print('  phase %d: epilogue %6d | self loop %6d | in-half %6d (max wave %6d) | out-half %6d (max %6d) | B1 wait %6d | k-loop %6d (max %6d) | prefetch issue %6d | B2 wait %6d'
                  % (ph, seg[0].mean(), seg[1].mean(), seg[2].mean(), seg[2].max(1).mean(), seg[3].mean(), seg[3].max(1).mean(),
                     seg[4].mean(), seg[5].mean(), seg[5].max(1).mean(), seg[6].mean(), seg[7].mean()))
        print('  tail: last k-loop -> stamp 62 (epilogue done) mean %6d, rel projection %6d' % ((st[:, :, 62] - prev).mean(), (st[:, :, 63] - st[:, :, 62]).mean()))
        print('  per-workgroup total: mean %d, min %d, max %d cycles' % ((st[:, :, 63].max(1) - st[:, :, 0].min(1)).mean(),
                                                                          (st[:, :, 63].max(1) - st[:, :, 0].min(1)).min(), (st[:, :, 63].max(1) - st[:, :, 0].min(1)).max()))


if __name__ == '__main__':
    main()
