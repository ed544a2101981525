import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('kgc-gcn_amd')
nat = pkg._native
dev = 'cuda:0'
def t(fn, n=50):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M = 40943
for N in (200, 128, 64):
    for K in (16, 64, 128, 304, 608, 1216):
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
        us = t(lambda: nat.matmul(A, B))
        print('M=%d K=%4d N=%3d  %7.1f us  %6.1f TF  A-read+C-write %5.1f MB' % (M, K, N, us, 2.0*M*K*N/us/1e6, (M*K+M*N)*4/1e6))
for Mx in (8192, 20480, 81920):
    A = torch.randn(Mx, 304, device=dev); B = torch.randn(304, 200, device=dev)
    us = t(lambda: nat.matmul(A, B))
    print('M=%d K=304 N=200  %7.1f us  %6.1f TF' % (Mx, us, 2.0*Mx*304*200/us/1e6))
