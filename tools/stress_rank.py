"""Randomised check (not part of the test suite) of the scoring kernels: score_target / score_rank (bit mask and dense
label filters, entity shards) against a torch recount over score_fwd's scores, which share the tile arithmetic."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('kgc-gcn_amd'); nat = pkg._native
dev = torch.device('cuda:0')


def run(seed=0, trials=60):
    """Asserts on a mismatch; tests/test_gpu_random_shapes.py runs a short instance of it."""
    rng = np.random.default_rng(seed)
    for trial in range(trials):
        B = int(rng.choice([1, 3, 4, 17, 64, 128, 200, 209, 500])); N = int(rng.integers(1, 3000)); O = int(rng.choice([4, 36, 100, 200, 256]))
        g = torch.Generator().manual_seed(trial)
        scale = float(rng.choice([0.05, 0.5, 4.0]))
        x = (torch.randn(B, O, generator=g) * scale).to(dev); ent = (torch.randn(N, O, generator=g)).to(dev)
        if N > 40: ent[N // 2:N // 2 + 10] = ent[:10]                      # exact ties
        bias = (torch.randn(N, generator=g) * 0.1).to(dev)
        if N > 40: bias[N // 2:N // 2 + 10] = bias[:10]
        obj = torch.randint(0, N, (B,), generator=g).to(dev)
        label = (torch.rand(B, N, generator=g) < 0.03).float().to(dev)
        words = (N + 31) // 32
        bits = torch.zeros(B, words * 32, dtype=torch.int64, device=dev); bits[:, :N] = label.long()
        mask = (bits.view(B, words, 32) << torch.arange(32, device=dev)).sum(2).to(torch.int32)   # wraps into the sign bit
        score = nat.score_fwd(x, ent, bias)
        target = nat.score_target(x, ent, bias, obj)
        rows = torch.arange(B, device=dev)
        assert torch.equal(target, score[rows, obj]), 'target'
        ids = torch.arange(N, device=dev).unsqueeze(0)
        keep = (label == 0) & (ids != obj.unsqueeze(1))
        want = torch.stack([((score > target.unsqueeze(1)) & keep).sum(1),
                            ((score == target.unsqueeze(1)) & keep & (ids < obj.unsqueeze(1))).sum(1),
                            ((score == target.unsqueeze(1)) & keep).sum(1)], 1)
        for kw in (dict(label=label), dict(mask=mask)):
            got = nat.score_rank(x, ent, bias, obj, target, **kw)
            assert torch.equal(got, want), ('full', trial, kw.keys())
        cut = int(rng.integers(0, N + 1))
        acc = torch.zeros((B, 3), dtype=torch.int64, device=dev)
        for lo, hi in ((0, cut), (cut, N)):
            if hi > lo:
                nat.score_rank(x, ent[lo:hi].contiguous(), bias[lo:hi].contiguous(), obj, target, label=label[:, lo:hi].contiguous(),
                               ent_row0=lo, counts=acc)
        assert torch.equal(acc, want), ('sharded', trial)
        print('trial %2d B=%3d N=%4d O=%3d scale %.2f  ties %d  ok' % (trial, B, N, O, scale, int(want[:, 2].sum())))
    print('all trials ok')
    return True


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60)
