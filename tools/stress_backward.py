"""Randomised check (not part of the test suite) of the aggregation forward + backward kernels against a float64 torch
restatement (index_add in edge order): random graphs, dims, hub settings; gradients w.r.t. x, the relation table and
the per-edge table."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('kgc-gcn_amd')
dev = torch.device('cuda:0')


def run(seed=0, trials=50):
    """-> (all trials ok, worst relative deviation); tests/test_gpu_random_shapes.py runs a short instance of it."""
    rng = np.random.default_rng(seed)
    worst = 0.0
    for trial in range(trials):
        N = int(rng.integers(1, 500)); R = int(rng.integers(1, 7)); E = int(rng.integers(1, 3000))
        D = int(rng.choice([3, 4, 36, 100, 128, 200, 300]))
        s = rng.integers(0, N, E)
        if rng.integers(0, 2) and N > 1:
            pz = 1.0 / np.arange(1, N + 1) ** 1.2
            o = rng.choice(N, size=E, p=pz / pz.sum())
        else:
            o = rng.integers(0, N, E)
        r = rng.integers(0, R, E)
        ei = torch.from_numpy(np.stack((np.concatenate((s, o)), np.concatenate((o, s))))).long()
        et = torch.from_numpy(np.concatenate((r, r + R))).long()
        thr = int(rng.choice([0, 3, 64])); chk = int(rng.choice([2, 5, 64]))
        csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, dev, hub_threshold=thr, hub_chunk=chk)
        g = torch.Generator().manual_seed(trial)
        x = torch.randn(N, D, generator=g); rel = torch.randn(2 * R + 1, D, generator=g); ee = torch.randn(2 * E, D, generator=g)
        G = torch.randn(N, 2 * D, generator=g)
        # float64 restatement (model.py:72-80, 99-100, 111-118 without the weight multiply)
        xd, rd, ed = (t.double().requires_grad_(True) for t in (x, rel, ee))
        outs = []
        for h in range(2):
            sl = slice(h * E, (h + 1) * E)
            src, dst, typ = ei[0, sl], ei[1, sl], et[sl]
            deg = torch.zeros(N, dtype=torch.float64).index_add_(0, src, torch.ones(E, dtype=torch.float64))
            c = deg.pow(-0.5); c[torch.isinf(c)] = 0
            norm = (c[src].float() * c[dst].float()).double()              # the f32 norm the feeder folds into the records
            m = xd[src] * rd[typ] * ed[sl] * norm.unsqueeze(1)
            outs.append(torch.zeros(N, D, dtype=torch.float64).index_add_(0, dst, m))
        ref = torch.cat(outs, 1)
        (ref * G.double()).sum().backward()
        xg, rg = x.to(dev).requires_grad_(True), rel.to(dev).requires_grad_(True)
        eg = ee.to(dev).index_select(0, csr.perm).requires_grad_(True)            # slot order
        out = pkg.model._AggregateFn.apply(xg, rg, eg, csr)
        (out * G.to(dev)).sum().backward()
        def dev_max(a, b):
            scale = float(b.abs().max()) + 1e-30
            return float((a.detach().cpu().double() - b).abs().max()) / scale
        errs = (dev_max(out, ref.detach()), dev_max(xg.grad, xd.grad), dev_max(rg.grad, rd.grad),
                dev_max(eg.grad.index_select(0, csr.inv_perm), ed.grad))
        worst = max(worst, max(errs))
        ok = max(errs) < 2e-5
        print('trial %2d N=%3d R=%d E=%4d D=%3d hubs(thr=%d, chunks=%d)  rel.err fwd %.1e gx %.1e grel %.1e gee %.1e %s'
              % ((trial, N, R, E, D, thr, csr.num_chunks) + errs + ('ok' if ok else 'MISMATCH',)))
        if not ok:
            return False, worst
    print('worst relative deviation %.2e' % worst)
    return True, worst


if __name__ == '__main__':
    ok, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 50)
    sys.exit(0 if ok else 1)
