"""Ablation table of the fused layer launch on the WN18RR shape (both layers of the bench workload): which role costs
what. MGCN_FUSED_ABLATE bits: 1 = no gather, 2 = no MFMA, 4 = no row stores, 8 = no epilogue. MGCN_FUSED_GRID = blocks.

    python tools/fused_ablate.py [--grids 512,256]
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--grids', default='')
ap.add_argument('--masks', default='0,1,2,4,8,5,9,13,3,14')
ap.add_argument('--nodes', type=int, default=0, help='override N (tiles = N/32); E stays')
ap.add_argument('--dims', default='100x200,200x200')
args = ap.parse_args()
pkg = importlib.import_module('kgc-gcn_amd')
nat = pkg._native
dev = torch.device('cuda:0')
shape = bench.SHAPES['wn18rr']
N, R, E = shape['N'], shape['R'], shape['E']
if args.nodes:
    shape = dict(shape, N=args.nodes)
    N = args.nodes
ei, ea = bench.synth_graph(shape, seed=0)
csr = pkg.GraphCSR(N, 2 * R + 1, ei, ea[0], dev, with_backward=False)
torch.manual_seed(0)


def timed(fn, n=30):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for D, O in [tuple(int(v) for v in d.split('x')) for d in args.dims.split(',')]:
    conv = pkg.MGCNConv(D, O, 2 * R).to(dev).eval()
    x = torch.randn(N, D, device=dev) * 0.1
    rel = torch.randn(2 * R, D, device=dev) * 0.3
    ee = torch.randn(2 * E, D, device=dev)
    _, wpack = conv.derived_weights()
    bn = conv.ent_bn
    out = torch.empty((N, O), device=dev)
    fn = lambda: nat.layer_fwd_fused(csr, x, rel, conv.loop_rel.reshape(-1), ee, True, conv.loop_edge.reshape(-1), wpack, O,
                                     None, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, out)
    for grid in [g for g in args.grids.split(',') if g] or [None]:
        if grid is None:
            os.environ.pop('MGCN_FUSED_GRID', None)
        else:
            os.environ['MGCN_FUSED_GRID'] = grid
        row = []
        for mask in args.masks.split(','):
            os.environ['MGCN_FUSED_ABLATE'] = mask
            row.append('%s:%6.1f' % (mask, timed(fn)))
        os.environ.pop('MGCN_FUSED_ABLATE', None)
        print('D=%d O=%d grid=%s  us by ablate mask  %s' % (D, O, grid or 'default', '  '.join(row)))
