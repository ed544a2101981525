"""ConvE trunk (model.py:161-175, torch / MIOpen / hipBLASLt — outside the HIP path) on 6268 queries: chunk size and
MIOpen find mode, to see what the evaluation's largest share responds to."""
import importlib, os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('kgc-gcn_amd')
dev = torch.device('cuda:0')
params = types.SimpleNamespace(gcn_in_dim=100, gcn_out_dim=200, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=10, k_h=20,
                               num_filter=200, kernel_size=7, bias=False, lbl_smooth=0.1, gcn_layers=1, device=dev)
torch.manual_seed(0)
conv = pkg.model.ConvE(params, 1000).to(dev).eval()
Q = 6268
src, rel = torch.randn(Q, 200, device=dev), torch.randn(Q, 200, device=dev)
def run(chunk):
    with torch.no_grad():
        return torch.cat([conv.trunk(src[i:i + chunk], rel[i:i + chunk]) for i in range(0, Q, chunk)])
def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    for chunk in (512, 1024, 2048, 4096, 6268):
        print('cudnn.benchmark=%s chunk %5d: %.3f ms' % (bench_mode, chunk, t(lambda: run(chunk))))
