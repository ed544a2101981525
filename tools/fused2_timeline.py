"""Stage timeline of the fused layer kernel from the diagnostics build's s_memtime stamps (not product code).
MGCN_LIB=.../libmgcn_hip_diag.so MGCN_FUSED_STAMPS=1 python tools/fused2_timeline.py [layer: 0|1]"""
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
    wcat, wpack = layer.derived_weights()
    out = torch.empty((N, O), device=dev)
    rel_out = torch.empty((2 * R, O), device=dev)
    with torch.no_grad():
        for _ in range(30):
            nat.layer_fwd_fused(csr, x, rel, layer.loop_rel.reshape(-1), ee, True, layer.loop_edge.reshape(-1), wpack, O,
                                layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, out,
                                rels_weight=layer.rels_weight.detach(), rel_out=rel_out)
        torch.cuda.synchronize()
    buf = np.zeros((1024, 2, 128), dtype=np.uint64)
    fn = nat.lib().mgcn_diag_read_stamps
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    assert fn(buf.ctypes.data) == 0
    st = buf[:256].astype(np.int64)
    t0 = st[:, :, 0].min()
    nst = int((st[0, 1] > 0).sum()) // 2
    print('stages per block: %d; launch span %d cycles' % (nst, st.max() - t0))
    for b in (0, 100, 255):
        print('block', b)
        for s in range(nst):
            g0, g1, m0, m1 = st[b, 1, 2 * s], st[b, 1, 2 * s + 1], st[b, 0, 2 * s], st[b, 0, 2 * s + 1]
            print('  stage %2d  gather start %7d work %6d | multiply start %7d work %6d' % (s, g0 - t0, g1 - g0, m0 - t0, m1 - m0))
    b = 100
    base = st[b, 0, 120]
    ev = [('entry', st[b, 0, 120])] + [('m%d start' % i, st[b, 0, 2 * i]) for i in range(nst)] + \
         [('m%d end' % i, st[b, 0, 2 * i + 1]) for i in range(nst)] + [('g%d start' % i, st[b, 1, 2 * i]) for i in range(nst)] + \
         [('g%d end' % i, st[b, 1, 2 * i + 1]) for i in range(nst)] + [('epi0 start', st[b, 0, 100]), ('epi0 end', st[b, 0, 101]),
          ('epi1 start', st[b, 0, 102]), ('epi1 end', st[b, 0, 103]), ('m exit', st[b, 0, 122]), ('g exit', st[b, 1, 122])]
    print('block 100 events (cycles since entry):', ', '.join('%s %d' % (n, t - base) for n, t in sorted(ev, key=lambda e: e[1])))
    print('block 100 gather stage 1 (in-half): start', st[b, 1, 2] - base, 'setup done', st[b, 1, 63] - base, 'batch events (issued, landed, ...):',
          [int(v - base) for v in st[b, 1, 64:80] if v > st[b, 1, 2] and v < st[b, 1, 3] + 100000], 'stage end', st[b, 1, 3] - base)
    ss = [int(v - base) for v in st[b, 1, 64:118] if v > 0]
    if ss:
        print('block 100 gather step stamps (start, consumed, issued; cycles since entry):', ss)
        d = np.diff(np.array(ss))
        print('  consume / issue / fetch+rest durations per step:', [(int(d[i]), int(d[i + 1]), int(d[i + 2])) for i in range(0, len(d) - 2, 3)])
    we = st[b, 0, 32:96].reshape(8, 8)
    if (we > 0).any():
        print('block 100 multiply stage ends per MFMA wave 0..7 (cycles since entry), stages 0..7:')
        for s_ in range(min(8, nst)):
            print('   stage %d:' % s_, [int(v - base) if v > 0 else None for v in we[s_]])
    gw = (st[:, 1, 1:2 * nst:2] - st[:, 1, 0:2 * nst:2])
    mw = (st[:, 0, 1:2 * nst:2] - st[:, 0, 0:2 * nst:2])
    print('median gather work per stage :', np.median(gw, axis=0).astype(int).tolist())
    print('median multiply work per stage:', np.median(mw, axis=0).astype(int).tolist())
    for role, name in ((0, 'multiply'), (1, 'gather')):
        dc = (st[:, role, 122] - st[:, role, 120]).astype(float)
        dr = (st[:, role, 123] - st[:, role, 121]).astype(float) / 100.0      # us (100 MHz)
        print('%s role: lifetime median %.1f us (max %.1f), %d cycles => clock %.2f GHz; first stage starts %d cycles after entry'
              % (name, np.median(dr), dr.max(), np.median(dc), np.median(dc / dr) / 1e3, np.median(st[:, role, 0] - st[:, role, 120])))
    print('epilogue of tile 0 / 1 (cycles, median):', int(np.median(st[:, 0, 101] - st[:, 0, 100])), int(np.median(st[:, 0, 103] - st[:, 0, 102])),
          '; last stage end -> epilogue start:', int(np.median(st[:, 0, 102] - st[:, 0, 2 * nst - 1])),
          '; tile-0 epilogue end -> next stage start:', int(np.median(st[:, 0, 2 * (nst // 2)] - st[:, 0, 101])))
    r0, r1 = st[:, :, 121].min(), st[:, :, 123].max()
    print('all workgroups: first entry to last exit %.1f us; entry spread %.1f us' % ((r1 - r0) / 100.0, (st[:, 0, 121].max() - r0) / 100.0))
    print('median end (cycles): gather %d multiply %d' % (np.median(st[:, 1, 2 * nst - 1] - t0), np.median(st[:, 0, 2 * nst - 1] - t0)))


if __name__ == '__main__':
    main()
