"""world_size-2 and -3 rehearsal of the entity-sharded scoring exchange on CPU (gloo). The three local kernels are
swapped for plain torch restatements (the HIP ones need a GPU); what is under test is the sharding, the
collectives and the bookkeeping: sharded counts must equal the unsharded ones exactly (integers)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .conftest import ROOT, load_pkg


class TorchKernels(object):
    """Same interface as kgc-gcn_amd._native for the scoring entry points (test stand-in, CPU)."""

    @staticmethod
    def score_target(x, ent, bias, obj, ent_row0=0, out=None):
        o = obj - ent_row0
        own = (o >= 0) & (o < ent.size(0))
        oc = o.clamp(0, ent.size(0) - 1)
        s = torch.sigmoid((x * ent[oc]).sum(1) + bias[oc])
        out[own] = s[own]
        return out

    @staticmethod
    def filter_mask(qkey, keys, ptr, tails, n_local, ent_row0=0, out=None):
        dense = torch.zeros((qkey.numel(), n_local), dtype=torch.bool)
        for b, k in enumerate(qkey.tolist()):
            i = int(torch.searchsorted(keys, torch.tensor(k)))
            if i < keys.numel() and int(keys[i]) == k:
                t = tails[ptr[i]:ptr[i + 1]].long() - ent_row0
                t = t[(t >= 0) & (t < n_local)]
                dense[b, t] = True
        return dense

    @staticmethod
    def score_rank(x, ent, bias, obj, target, label=None, ent_row0=0, counts=None, mask=None):
        s = torch.sigmoid(x @ ent.t() + bias)
        idx = torch.arange(ent.size(0))[None, :] + ent_row0
        live = ~mask & (idx != obj[:, None])
        gt = (live & (s > target[:, None])).sum(1)
        eq = live & (s == target[:, None])
        return torch.stack([gt, (eq & (idx < obj[:, None])).sum(1), eq.sum(1)], dim=1)


def _problem(seed=0, B=6, N=101, O=8, world=2):
    g = torch.Generator().manual_seed(seed)
    x = [torch.randn(B, O, generator=g) for _ in range(world)]
    ent, bias = torch.randn(N, O, generator=g) * 0.5, torch.randn(N, generator=g) * 0.1
    sub = [torch.randint(0, N, (B,), generator=g) for _ in range(world)]
    rel = [torch.randint(0, 4, (B,), generator=g) for _ in range(world)]
    obj = [torch.randint(0, N, (B,), generator=g) for _ in range(world)]
    known = {}
    for r in range(world):
        for b in range(B):
            known.setdefault((int(sub[r][b]), int(rel[r][b])), set()).update(
                {int(obj[r][b])} | {int(v) for v in torch.randint(0, N, (3,), generator=g)})
    return x, ent, bias, sub, rel, obj, known


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    pkg = load_pkg()
    x, ent, bias, sub, rel, obj, known = _problem(world=world)
    filt = pkg.dist.FilterIndex.from_known(known, 4)
    b = pkg.dist.shard_bounds(ent.size(0), world)
    counts, target = pkg.dist.sharded_rank_counts(
        x[rank], filt.query_keys(sub[rank], rel[rank]), obj[rank], ent[b[rank]:b[rank + 1]], bias[b[rank]:b[rank + 1]],
        b[rank], filt, kernels=TorchKernels)
    q.put((rank, counts.clone(), target.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])   # 3: uneven entity shards (101 rows: 34 + 34 + 33), three query blocks
def test_sharded_counts_equal_unsharded_gloo(world):
    port = 29500 + (os.getpid() + 7 * world) % 2000
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((r, (c, t)) for r, c, t in [q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pkg = load_pkg()
    x, ent, bias, sub, rel, obj, known = _problem(world=world)
    filt = pkg.dist.FilterIndex.from_known(known, 4)
    for r in range(world):
        want_c, want_t = pkg.dist.sharded_rank_counts(x[r], filt.query_keys(sub[r], rel[r]), obj[r], ent, bias, 0, filt,
                                                      kernels=TorchKernels)
        assert torch.equal(got[r][0], want_c)
        assert torch.equal(got[r][1], want_t)
        assert int(want_c.sum()) > 0                         # the comparison is not over all-zero counts


def test_shard_bounds_and_filter_index(pkg):
    assert pkg.dist.shard_bounds(10, 4) == [0, 3, 6, 9, 10]
    assert pkg.dist.shard_bounds(3, 4) == [0, 1, 2, 3, 3]
    assert pkg.dist.shard_bounds(14541, 8)[-1] == 14541
    f = pkg.dist.FilterIndex.from_known({(3, 1): [5, 2], (0, 0): [7]}, 4)
    assert f.keys.tolist() == [0, 13] and f.ptr.tolist() == [0, 1, 3] and f.tails.tolist() == [7, 2, 5]
    assert f.query_keys(torch.tensor([3]), torch.tensor([1])).tolist() == [13]


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    pkg = load_pkg()
    d = pkg.dist
    t = torch.arange(6, dtype=torch.float32).reshape(3, 2) + 10 * rank
    out = d._gather(t, None, world)                                        # [world * 3, 2] in rank order
    empty = d._gather(t[:0], None, world)                                  # zero rows on every rank: no collective, no error
    keys = list(d._INTO_TENSOR)                                            # the capability was probed exactly once
    ints = d._gather(torch.tensor([rank, rank + 7]), None, world)
    # a rank-local failure of the collective must RAISE, not make this rank take another collective than its peers are in
    # (ADVICE r3): all ranks pass a wrongly sized output here, so the error is the same everywhere and nobody is left waiting
    raised = False
    try:
        d._gather_into(torch.empty((world * 3 + 1, 2)), t, None)
    except Exception:                                                      # noqa: BLE001 (torch raises RuntimeError / ValueError)
        raised = True
    q.put((rank, out.clone(), tuple(empty.shape), keys, ints.clone(), raised))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_helper_gloo():
    """dist._gather_into: one collective in rank order; the all_gather_into_tensor capability is decided once per backend by a
    probe every rank runs, zero rows are no collective, and a failing collective raises on the rank it fails on."""
    world = 2
    port = 29500 + (os.getpid() + 991) % 2000
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((r, rest) for r, *rest in [q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = torch.cat([torch.arange(6, dtype=torch.float32).reshape(3, 2) + 10 * r for r in range(world)])
    for r in range(world):
        out, empty_shape, keys, ints, raised = got[r]
        assert torch.equal(out, want)
        assert empty_shape == (0, 2)
        assert keys == [('gloo', 'cpu')]
        assert ints.tolist() == [0, 7, 1, 8]
        assert raised
