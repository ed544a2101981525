"""CPU-only tests: the C-ABI library loads and exports what include/mgcn_hip.h declares, the host feeder
(integer work, bit-exact), the loader surface, and the loud failure of the product path without a GPU."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from .conftest import ALL_CASES, FULL_CASES, GOLDEN, ROOT, golden


def test_library_exports_every_declared_symbol(pkg):
    header = open(os.path.join(ROOT, 'include', 'mgcn_hip.h')).read()
    declared = set(re.findall(r'\b(mgcn_[a-z0-9_]+)\s*\(', header))
    assert declared == set(pkg._native.EXPORTS)
    handle = ctypes.CDLL(pkg._native.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert pkg._native.lib().mgcn_abi_version() == 4


@pytest.mark.parametrize('case', ALL_CASES)
def test_feeder_bit_exact(pkg, case):
    g = golden(case)
    ei, et = g.t('dl_edge_index'), g.t('dl_edge_attr')[0]
    N, R, E = int(g['dl_num_entity']), int(g['dl_num_relation']), int(g['dl_num_edge'])
    h = pkg._native.csr_build_host(N, 2 * R + 1, ei, et, hub_threshold=0)      # strict layout: no destination is split
    assert h['num_chunks'] == 0 and bool((h['hubinfo'][:, :, 0] == -1).all())
    for half in range(2):
        lo = half * E
        dst = ei[1, lo:lo + E].numpy()
        order = np.argsort(dst, kind='stable') + lo                       # (dst, edge id) order
        assert np.array_equal(h['perm'][lo:lo + E].numpy(), order)
        assert np.array_equal(h['rowptr'][half].numpy(), lo + np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=N))]))
        assert np.array_equal(h['rec'][lo:lo + E, 0].numpy(), ei[0].numpy()[order])
        assert np.array_equal(h['rec'][lo:lo + E, 1].numpy(), et.numpy()[order])
        assert np.array_equal(h['rec'][lo:lo + E, 3].numpy(), order)
        sd = h['slot_dst'][lo:lo + E].numpy()
        assert np.array_equal(sd & 0x7fffffff, dst[order - lo]) and bool((((sd >> 31) & 1) == half).all())
    perm, mirror = h['perm'].numpy(), h['mirror'].numpy()
    assert np.array_equal(perm[mirror], (perm + E) % (2 * E))              # slot of the reverse edge
    assert np.array_equal(h['rec'][:, 0].numpy()[mirror], h['slot_dst'].numpy() & 0x7fffffff)   # src of reverse = dst
    typ = h['rec'][:, 1].numpy()
    assert np.array_equal(h['typeslots'].numpy(), np.argsort(typ, kind='stable'))
    assert np.array_equal(h['typeptr'].numpy(), np.concatenate([[0], np.cumsum(np.bincount(typ, minlength=2 * R + 1))]))
    norms = h['rec'][:, 2].contiguous().view(torch.float32).numpy()
    want = np.concatenate([g['norm_in'], g['norm_out']])[h['perm'].numpy()]
    assert np.array_equal(norms, want)                                    # f32 bit-exact (model.py:72-80)


@pytest.mark.parametrize('case', ['syn_b', 'syn_c'])
def test_feeder_hub_splitting(pkg, case):
    """Destinations with more than `hub_threshold` slots in a half leave the main CSR; their slots, still in edge-id
    order, are cut into chunks of `hub_chunk`. Every edge lands in exactly one slot."""
    g = golden(case)
    ei, et = g.t('dl_edge_index'), g.t('dl_edge_attr')[0]
    N, R, E = int(g['dl_num_entity']), int(g['dl_num_relation']), int(g['dl_num_edge'])
    T, C = 8, 5
    h = pkg._native.csr_build_host(N, 2 * R + 1, ei, et, hub_threshold=T, hub_chunk=C)
    perm = h['perm'].numpy()
    assert np.array_equal(np.sort(perm), np.arange(2 * E))
    assert np.array_equal(h['rec'][:, 3].numpy(), perm) and np.array_equal(h['rec'][:, 0].numpy(), ei[0].numpy()[perm])
    hubs = 0
    for half in range(2):
        lo = half * E
        dst = ei[1, lo:lo + E].numpy()
        cnt = np.bincount(dst, minlength=N)
        rp = h['rowptr'][half].numpy()
        for n in range(N):
            want = np.nonzero(dst == n)[0] + lo                          # this destination's edges, edge-id order
            first, nch = h['hubinfo'][half, n].tolist()
            if cnt[n] > T:
                hubs += 1
                assert rp[n + 1] == rp[n] and nch == -(-cnt[n] // C)
                ranges = h['chunks'][first:first + nch, :2].numpy()
                assert bool((h['chunks'][first:first + nch, 2] == first).all()) and bool((h['chunks'][first:first + nch, 3] == nch).all())
                assert all(0 < e - b <= C for b, e in ranges) and all(ranges[k][1] == ranges[k + 1][0] for k in range(nch - 1))
                assert np.array_equal(perm[ranges[0][0]:ranges[-1][1]], want)
            else:
                assert (first, nch) == (-1, 0)
                assert np.array_equal(perm[rp[n]:rp[n + 1]], want)
    assert hubs > 0 and h['num_chunks'] == int(h['hubinfo'][:, :, 1].sum())
    assert h['rowptr'][1, 0] == h['rowptr'][0, N]                            # out-half main region follows the in-half one


def test_destination_shards_with_hubs(pkg):
    """Host index for the destination partition: the three runs of a rank's table shard (in-half, out-half, hub region)
    hold exactly the slots whose destination the rank owns, at row = slot - ee_sub[region]; the rank's hub chunks are
    one run of the chunk table (hub region in node order)."""
    g = golden('syn_c')
    ei, et = g.t('dl_edge_index'), g.t('dl_edge_attr')[0]
    N, R, E = int(g['dl_num_entity']), int(g['dl_num_relation']), int(g['dl_num_edge'])
    csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, 'cpu', hub_threshold=8, hub_chunk=5)
    assert csr.num_chunks > 0
    firsts = csr.hubinfo[:, :, 0].t().reshape(-1)                            # (node, half) order
    firsts = firsts[firsts >= 0]
    assert bool((firsts[1:] > firsts[:-1]).all())
    table = torch.arange(2 * E, dtype=torch.float32).unsqueeze(1)           # row value = slot id
    dst = csr.slot_dst & 0x7fffffff
    b = pkg.dist.shard_bounds(N, 3)
    chunks_seen = 0
    for r in range(3):
        n0, n1 = b[r], b[r + 1]
        shard = csr.edge_table_shard(table, n0, n1)[:, 0].long()
        counts, sub = csr.shard_slot_counts(n0, n1), csr.shard_ee_sub(n0, n1)
        assert shard.numel() == sum(counts)
        own = torch.nonzero((dst >= n0) & (dst < n1)).reshape(-1)
        assert torch.equal(torch.sort(shard).values, own)
        row = 0
        for region in range(3):
            assert torch.equal(shard[row:row + counts[region]] - torch.arange(row, row + counts[region]),
                               torch.full((counts[region],), sub[region], dtype=torch.long))
            row += counts[region]
        c0, c1 = csr.chunk_range(n0, n1)
        assert c0 == chunks_seen
        chunks_seen = c1
        if c1 > c0:
            hub_slots = shard[counts[0] + counts[1]:]
            assert int(csr.chunks[c0, 0]) == int(hub_slots[0]) and int(csr.chunks[c1 - 1, 1]) == int(hub_slots[-1]) + 1
    assert chunks_seen == csr.num_chunks


@pytest.mark.parametrize('thr', [0, 8])
def test_balanced_destination_bounds(pkg, thr):
    """ranges balanced by work (slots, hub slots included, + nodes), not by node count."""
    g = golden('syn_c')                                                     # the skewed golden graph (max degree 409)
    ei, et = g.t('dl_edge_index'), g.t('dl_edge_attr')[0]
    N, R = int(g['dl_num_entity']), int(g['dl_num_relation'])
    csr = pkg.GraphCSR(N, 2 * R + 1, ei, et, 'cpu', hub_threshold=thr, hub_chunk=5)
    dst = csr.slot_dst & 0x7fffffff
    def spread(b):
        work = [int(((dst >= b[r]) & (dst < b[r + 1])).sum()) + (b[r + 1] - b[r]) for r in range(len(b) - 1)]
        return max(work) / (sum(work) / len(work))
    for W in (1, 2, 3):
        b = csr.balanced_bounds(W, align=4)
        assert b[0] == 0 and b[-1] == N and len(b) == W + 1 and all(x <= y for x, y in zip(b, b[1:]))
        assert all(x % 4 == 0 for x in b[1:-1])
        if W > 1:
            assert spread(b) < 1.05 < spread(pkg.dist.shard_bounds(N, W))
    assert csr.balanced_bounds(3, align=4) == pkg.GraphCSR(N, 2 * R + 1, ei, et, 'cpu', hub_threshold=8 - thr,
                                                          hub_chunk=5).balanced_bounds(3, align=4)


def test_feeder_rejects_bad_input(pkg):
    ei = torch.tensor([[0, 5], [1, 0]])
    with pytest.raises(pkg._native.NativeError, match='outside'):
        pkg._native.csr_build_host(3, 3, ei, torch.tensor([0, 1]))
    with pytest.raises(pkg._native.NativeError, match='type'):
        pkg._native.csr_build_host(6, 1, ei, torch.tensor([0, 1]))
    with pytest.raises(pkg._native.NativeError, match='self-loop row'):    # the last relation row belongs to the self-loop pass
        pkg._native.csr_build_host(6, 3, ei, torch.tensor([0, 2]))
    h = pkg._native.csr_build_host(4, 3, torch.empty((2, 0), dtype=torch.int64), torch.empty(0, dtype=torch.int64))
    assert h['rowptr'].abs().sum() == 0                                    # empty graph


def _loader(pkg, g):
    cwd = os.getcwd()
    os.chdir(GOLDEN)
    try:
        params = types.SimpleNamespace(**g.hp)
        return pkg.DataLoader(os.path.basename(g.data_dir), params), params
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize('case', ALL_CASES)
def test_data_loader_matches_reference(pkg, case):
    g = golden(case)
    dl, params = _loader(pkg, g)
    assert (dl.num_entity, dl.num_relation, dl.num_edge) == (int(g['dl_num_entity']), int(g['dl_num_relation']), int(g['dl_num_edge']))
    assert sorted(dl.entity2id, key=dl.entity2id.get) == list(g['dl_entity_names'])
    assert sorted(dl.relation2id, key=dl.relation2id.get) == list(g['dl_relation_names'])
    gr = dl.graph
    assert np.array_equal(gr.edge_index.numpy(), g['dl_edge_index'])
    assert np.array_equal(gr.edge_attr.numpy(), g['dl_edge_attr'])
    assert np.array_equal(gr.edge_norm.numpy(), g['dl_edge_norm'])
    assert np.array_equal(gr.entity.numpy(), g['dl_entity']) and gr.num_nodes == int(g['dl_num_nodes'])
    for split in ('train', 'valid_tail', 'valid_head', 'test_tail', 'test_head'):
        qs = dl.triplets[split]
        assert np.array_equal(np.array([q['triple'] for q in qs], dtype=np.int64).reshape(-1, 3), g['dl_q_%s_triple' % split])
        ptr = g['dl_q_%s_label_ptr' % split]
        idx = g['dl_q_%s_label_idx' % split]
        for i, q in enumerate(qs):
            assert sorted(q['label']) == list(idx[ptr[i]:ptr[i + 1]])
    iters = dl.get_data_loaders(4, 0, params)
    assert set(iters) == {'train', 'valid_head', 'valid_tail', 'test_head', 'test_tail'}
    trip, lab = next(iter(iters['valid_tail']))
    assert trip.dtype == torch.int64 and trip.shape[1] == 3 and lab.shape == (trip.shape[0], dl.num_entity)
    with pytest.raises(ValueError):
        dl._get_dataset('nope', params)


@pytest.mark.parametrize('case', FULL_CASES)
def test_label_rows_and_smoothing(pkg, case):
    g = golden(case)
    dl, params = _loader(pkg, g)
    ds = dl._get_dataset('train', params)
    assert np.array_equal(ds[0][1].numpy(), g['smooth_label0'])           # (1-eps)*y + 1/N (Q6)
    assert ds[0][0].tolist()[2] == -1
    ev = dl._get_dataset('test_tail', params)
    row = ev[0][1]
    assert set(row.unique().tolist()) <= {0.0, 1.0}


def _write_splits(root, name, texts):
    d = root / 'data' / name
    d.mkdir(parents=True)
    for split, text in zip(('train', 'valid', 'test'), texts):
        (d / (split + '.txt')).write_bytes(text.encode('utf-8') if isinstance(text, str) else text)
    return d


def _load_in(pkg, root, name, native):
    cwd = os.getcwd()
    os.chdir(root)
    old = os.environ.get('MGCN_NATIVE_INGEST')
    os.environ['MGCN_NATIVE_INGEST'] = '1' if native else '0'
    try:
        return pkg.DataLoader(name, types.SimpleNamespace(lbl_smooth=0.0))
    finally:
        os.chdir(cwd)
        if old is None:
            os.environ.pop('MGCN_NATIVE_INGEST', None)
        else:
            os.environ['MGCN_NATIVE_INGEST'] = old


@pytest.mark.parametrize('case', ALL_CASES)
def test_native_ingest_equals_python_reader(pkg, case):
    """the C++ reader assigns the ids the two Python passes assign (goldens = the reference's own), and the
    sort-based known-answer index equals the dict-of-sets one."""
    g = golden(case)
    root, name = os.path.dirname(os.path.dirname(g.data_dir)), os.path.basename(g.data_dir)
    nat, py = _load_in(pkg, root, name, True), _load_in(pkg, root, name, False)
    assert nat._id_triples is not None and py._id_triples is None
    assert list(nat.entity2id.items()) == list(py.entity2id.items())        # same names, same ids, same insertion order
    assert list(nat.relation2id.items()) == list(py.relation2id.items())
    assert nat.triplets == py.triplets
    assert np.array_equal(nat.graph.edge_index.numpy(), py.graph.edge_index.numpy())
    a, b = nat.filter_index(), py.filter_index()
    assert torch.equal(a.keys, b.keys) and torch.equal(a.ptr, b.ptr) and torch.equal(a.tails, b.tails)
    assert a.num_rel_ids == b.num_rel_ids


def test_ingest_only_loader(pkg):
    """The scalable subset (ids, graph, filter index, evaluation queries) without the per-query dict lists: identical
    tensors to the full loader's."""
    g = golden('syn_b')
    root, name = os.path.dirname(os.path.dirname(g.data_dir)), os.path.basename(g.data_dir)
    full = _load_in(pkg, root, name, True)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        lean = pkg.DataLoader(name, types.SimpleNamespace(lbl_smooth=0.0, ingest_only=True))
    finally:
        os.chdir(cwd)
    assert lean.triplets is None and (lean.num_entity, lean.num_relation, lean.num_edge) == (full.num_entity, full.num_relation, full.num_edge)
    assert np.array_equal(lean.graph.edge_index.numpy(), full.graph.edge_index.numpy())
    assert np.array_equal(lean.graph.edge_attr.numpy(), full.graph.edge_attr.numpy())
    for split in ('valid', 'test'):
        rows = [q['triple'] for q in full.triplets[split + '_tail']] + [q['triple'] for q in full.triplets[split + '_head']]
        assert torch.equal(lean.eval_queries(split), torch.tensor(rows, dtype=torch.int64).reshape(-1, 3))
        assert torch.equal(full.eval_queries(split), lean.eval_queries(split))
    a, b = lean.filter_index(), pkg.dist.FilterIndex.from_known(full._known_all, 2 * full.num_relation)
    assert torch.equal(a.keys, b.keys) and torch.equal(a.ptr, b.ptr) and torch.equal(a.tails, b.tails)
    with pytest.raises(ValueError):
        lean.get_data_loaders(4, 0, types.SimpleNamespace(lbl_smooth=0.0))


@pytest.mark.parametrize('case', ['toy_small', 'syn_b'])
def test_train_index_and_queries(pkg, case):
    """N2 for training, host side: the train-split index marks exactly the train dataset's label sets, from either reader."""
    g = golden(case)
    root, name = os.path.dirname(os.path.dirname(g.data_dir)), os.path.basename(g.data_dir)
    for native in (True, False):
        dl = _load_in(pkg, root, name, native)
        idx, q = dl.train_index(), dl.train_queries()
        assert q.shape == (len(dl.triplets['train']), 2)
        keys = idx.query_keys(q[:, 0], q[:, 1])
        pos = torch.searchsorted(idx.keys, keys)
        assert torch.equal(idx.keys[pos], keys)
        for i, item in enumerate(dl.triplets['train']):
            lo, hi = int(idx.ptr[pos[i]]), int(idx.ptr[pos[i] + 1])
            assert idx.tails[lo:hi].tolist() == sorted(item['label'])
    cwd = os.getcwd()
    os.chdir(root)
    try:
        lean = pkg.DataLoader(name, types.SimpleNamespace(lbl_smooth=0.0, ingest_only=True))
    finally:
        os.chdir(cwd)
    assert sorted(map(tuple, lean.train_queries().tolist())) == sorted(map(tuple, q.tolist()))


def test_native_ingest_line_semantics(pkg, tmp_path):
    """Whitespace, newline and error behaviour of data_loader.py:61-70 as the Python reader shows it."""
    ok = 'a  r1\tb\r\nb r2 c\n c\tr1\ta \rd r2 d'                          # mixed separators, \r\n, lone \r, no final newline
    _write_splits(tmp_path, 'ok', (ok, 'a r1 c\n', 'e r3 a\n'))
    nat, py = _load_in(pkg, tmp_path, 'ok', True), _load_in(pkg, tmp_path, 'ok', False)
    assert nat._id_triples is not None
    assert list(nat.entity2id.items()) == list(py.entity2id.items()) == [('a', 0), ('b', 1), ('c', 2), ('d', 3), ('e', 4)]
    assert nat.triplets == py.triplets and nat.num_edge == 4
    _write_splits(tmp_path, 'short', ('a r b\na r\n', 'a r b\n', 'a r b\n'))
    _write_splits(tmp_path, 'blank', ('a r b\n\na r b\n', 'a r b\n', 'a r b\n'))
    _write_splits(tmp_path, 'long', ('a r b c\n', 'a r b\n', 'a r b\n'))
    for name in ('short', 'blank', 'long'):
        for native in (True, False):
            with pytest.raises(ValueError):
                _load_in(pkg, tmp_path, name, native)
    _write_splits(tmp_path, 'upper', ('a r b\n', 'a r B\n', 'a r b\n'))
    for native in (True, False):
        with pytest.raises(KeyError):
            _load_in(pkg, tmp_path, 'upper', native)
    _write_splits(tmp_path, 'utf', ('caf\u00e9 r b\n', 'b r caf\u00e9\n', 'b r b\n'))      # declined by the native reader
    nat = _load_in(pkg, tmp_path, 'utf', True)
    assert nat._id_triples is None and nat.entity2id == {'caf\u00e9': 0, 'b': 1}
    (tmp_path / 'data' / 'missing').mkdir()
    for native in (True, False):
        with pytest.raises(FileNotFoundError):
            _load_in(pkg, tmp_path, 'missing', native)


def test_native_ingest_random_200k(pkg, tmp_path):
    """A larger file: ids and query lists of the native reader equal the Python reader's; prints both wall-clocks."""
    import time
    rng = np.random.default_rng(5)
    def lines(n):
        s, r, o = rng.integers(0, 30000, n), rng.integers(0, 200, n), rng.integers(0, 30000, n)
        return ''.join('e%d\tr%d\te%d\n' % t for t in zip(s, r, o))
    _write_splits(tmp_path, 'big', (lines(200000), lines(5000), lines(5000)))
    t0 = time.perf_counter()
    nat = _load_in(pkg, tmp_path, 'big', True)
    t1 = time.perf_counter()
    py = _load_in(pkg, tmp_path, 'big', False)
    t2 = time.perf_counter()
    paths = [str(tmp_path / 'data' / 'big' / (sp + '.txt')) for sp in ('train', 'valid', 'test')]
    t3 = time.perf_counter()
    _, _, ids = pkg._native.ingest(*paths)
    t4 = time.perf_counter()
    pkg._native.filter_index_build(torch.cat(list(ids.values())), len(nat.relation2id) // 2)
    t5 = time.perf_counter()
    scratch = object.__new__(pkg.DataLoader)                  # the two Python passes alone, on a throw-away object
    scratch.data_dir = str(tmp_path / 'data' / 'big')
    t6 = time.perf_counter(); scratch._read_ids_python(); t7 = time.perf_counter()
    t8 = time.perf_counter(); pkg.dist.FilterIndex.from_known(py._known_all, 2 * py.num_relation); t9 = time.perf_counter()
    print('210k triples: ids native %.3f s vs Python %.3f s; known-answer index native %.3f s vs Python %.3f s; '
          'whole DataLoader (query lists built in Python in both) %.2f s vs %.2f s'
          % (t4 - t3, t7 - t6, t5 - t4, t9 - t8, t1 - t0, t2 - t1))
    assert list(nat.entity2id.items()) == list(py.entity2id.items())
    assert list(nat.relation2id.items()) == list(py.relation2id.items())
    assert np.array_equal(nat.graph.edge_index.numpy(), py.graph.edge_index.numpy())
    assert np.array_equal(nat.graph.edge_attr.numpy(), py.graph.edge_attr.numpy())
    a, b = nat.filter_index(), py.filter_index()
    assert torch.equal(a.keys, b.keys) and torch.equal(a.ptr, b.ptr) and torch.equal(a.tails, b.tails)


def test_uppercase_token_raises_like_reference(pkg, tmp_path):
    d = tmp_path / 'data' / 'u'
    d.mkdir(parents=True)
    for split in ('train', 'valid', 'test'):
        (d / (split + '.txt')).write_text('A\tr\tb')
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        with pytest.raises(KeyError):
            pkg.DataLoader('u', types.SimpleNamespace(lbl_smooth=0.0))
    finally:
        os.chdir(cwd)


def test_graph_to_is_in_place(pkg):
    g = golden('toy_small')
    dl, _ = _loader(pkg, g)
    gr = dl.graph
    assert gr.to('cpu') is gr
    edge_type, edge_ids = gr.edge_attr                                      # model.py:26 unpacking
    assert edge_type.shape == edge_ids.shape


def test_state_dict_keys_match_reference(pkg):
    g = golden('toy_small')
    dl, params = _loader(pkg, g)
    model = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
    want = set(g.state_dict())
    assert set(model.state_dict()) == want
    model.load_state_dict(g.state_dict())                                  # a reference checkpoint loads
    two = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, types.SimpleNamespace(gcn_layers=2, **g.hp))
    assert set(two.state_dict()) - want == {k for k in two.state_dict() if 'extra' in k}


def test_checkpoint_round_trip_with_numpy_measure(pkg, tmp_path):
    """main.py:158-162 stores `measure` = np.round(np.float64, 5): a checkpoint written that way (by the reference or by
    this package's training flow) must load through the no-code loader, restore the model and return the measure."""
    g = golden('toy_small')
    dl, params = _loader(pkg, g)
    model = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
    model.load_state_dict(g.state_dict())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    measure = np.round(np.float64(0.4299600001), 5)
    # (a) exactly what the reference's utils.save_checkpoint does: torch.save of the dict, np.float64 inside
    ref_style = os.path.join(tmp_path, 'ref.ckpt')
    torch.save({'state_dict': model.state_dict(), 'optim_dict': opt.state_dict(), 'measure': measure}, ref_style)
    # (b) this package's writer
    pkg.utils.save_checkpoint({'state_dict': model.state_dict(), 'optim_dict': opt.state_dict(), 'measure': measure}, True,
                              str(tmp_path))
    for path in (ref_style, os.path.join(tmp_path, 'last.ckpt'), os.path.join(tmp_path, 'best.ckpt')):
        fresh = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
        fresh_opt = torch.optim.Adam(fresh.parameters(), lr=1e-3)
        got = pkg.utils.load_checkpoint(path, fresh, fresh_opt)
        assert got == float(measure)
        for k, v in g.state_dict().items():
            assert torch.equal(fresh.state_dict()[k].reshape(-1), v.reshape(-1)), k
    with pytest.raises(FileNotFoundError):
        pkg.utils.load_checkpoint(os.path.join(tmp_path, 'missing.ckpt'), model)


def test_seeded_init_matches_reference(pkg):
    """Same construction order and initialisers as the reference (utils.py:113-118, model.py:12-22,49-70,
    132-157): with the same seed the parameters are identical, so seeded runs are comparable."""
    g = golden('toy_small')
    if not g.has('init_sum_entity_embedding'):
        pytest.skip('golden has no init sums')
    dl, params = _loader(pkg, g)
    torch.manual_seed(2020)
    model = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
    for k, p in model.named_parameters():
        assert float(p.double().sum()) == float(g['init_sum_' + k]), k


def test_product_path_has_no_cpu_fallback(pkg):
    g = golden('toy_small')
    dl, params = _loader(pkg, g)
    model = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params).eval()
    trip = g.t('dl_q_test_tail_triple')
    with torch.no_grad(), pytest.raises(pkg._native.NativeError, match='GPU'):
        model(trip[:, 0], trip[:, 1], dl.graph)


def test_product_does_not_import_oracle():
    pkg_dir = os.path.join(ROOT, 'kgc-gcn_amd')
    for fn in os.listdir(pkg_dir):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg_dir, fn)).read()
            assert 'oracle' not in src.replace('no oracle', ''), fn


def test_dropin_module_names(pkg):
    import sys
    saved = {k: sys.modules.get(k) for k in ('model', 'data_loader', 'utils')}
    try:
        pkg.dropin.install()
        import data_loader
        import model
        import utils
        assert model.MGCN is pkg.MGCN and data_loader.DataLoader is pkg.DataLoader and hasattr(utils, 'get_param')
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_abi_argument_validation_without_a_gpu(pkg):
    """Every device entry point validates its arguments before any HIP call: bad sizes / null pointers come back as
    MGCN_EINVAL (1) with a message, on a machine without a GPU too."""
    lib = pkg._native.lib()
    N = None
    cases = [
        ('mgcn_aggregate_fwd', (-1, 0, 4, 3, N, N, N, 4, N, N, N, 1, N, N, 12, 0, 0, N, N, 0, 0, N, 0, 0, 0, N), 'bad sizes'),
        ('mgcn_aggregate_fwd', (4, 2, 4, 3, N, N, N, 4, N, N, N, 1, N, N, 12, 0, 4, N, N, 0, 0, N, 0, 0, 0, N), 'null pointer'),
        ('mgcn_aggregate_bwd', (4, 2, 0, 3, N, N, N, N, N, N, 0, N, N, N, 4, N, N, N, 8, N, N, N, N, 0, N), 'bad sizes'),
        ('mgcn_dense_bn_tanh_fwd', (4, 4, 4, N, 12, N, N, N, N, N, N, 1e-5, N, 4, N), 'null pointer'),
        ('mgcn_layer_fwd_fused', (4, 2, 4, 4, 3, N, N, N, 4, N, N, N, 1, N, N, N, N, N, N, N, 1e-5, N, 4, 2, 1, 0, 0, 0, N, N, 0,
                                  0, N, N, N, N, 0, 0, N, N), 'bad node range'),
        ('mgcn_score_fwd', (4, 8, 4, N, 4, N, 4, N, N, 8, N), 'null pointer'),
        ('mgcn_score_rank', (4, 8, 0, 4, N, 4, N, 4, N, N, N, N, 0, N, 0, N, N), 'null pointer'),
        ('mgcn_filter_mask', (4, N, 0, N, N, N, 0, 8, N, 1, N), 'null pointer'),
        ('mgcn_label_rows', (4, N, 0, N, N, N, 0, 8, 1.0, 0.0, N, 4, N), 'leading|sizes|null'),
        ('mgcn_score_bce_fwd', (4, 8, 4, N, 4, N, 4, N, N, 1, 1.0, 0.0, 0.1, N, 4, N, N), 'null pointer'),
        ('mgcn_matmul_f32', (4, 4, 4, N, 4, N, 4, N, 4, N), 'null pointer'),
        ('mgcn_pack_weights', (4, 4, N, N, 0, N), 'bad arguments'),
    ]
    for name, args, pattern in cases:
        rc = getattr(lib, name)(*args)
        msg = lib.mgcn_last_error().decode()
        assert rc == 1, (name, rc, msg)
        assert re.search(pattern, msg), (name, msg)
    # fused layer: 3 modes x 4 k-blocks of 32 (100 -> 128 columns) x 13 column tiles x 3 bf16 pieces x 1 KiB
    assert lib.mgcn_packed_weights_bytes(100, 200) == 3 * 4 * 13 * 3 * 64 * 16
    assert lib.mgcn_packed_weights_bytes(200, 200) == 3 * 7 * 13 * 3 * 64 * 16   # ceil(200 / 32) = 7 k-blocks per mode
    assert lib.mgcn_aggregate_bwd_workspace(10, 4, 3, 2) == (2 + 3 + 2) * 4 * 4      # ceil(20/16) chunks + rows + hub chunks
    assert lib.mgcn_score_bce_partials(128, 40943) == 1280
    assert lib.mgcn_hub_partial_floats(10, 100) == 10 * 100 + 2 * 10 and lib.mgcn_hub_partial_floats(0, 100) == 0


def test_chunkwise_xavier_table_rows(pkg):
    """dist.xavier_rows: the per-edge table of a destination-partitioned run is DEFINED chunk by chunk, so a rank can
    materialise exactly the rows it owns (any order, any subset) without the table: every subset equals the same rows of
    the full materialisation, and the values are xavier-uniform for a [rows, dim] parameter (utils.py:113-118)."""
    rows, dim, seed = 1000, 12, 5
    full = pkg.dist.xavier_rows(torch.arange(rows), rows, dim, seed, 'cpu', chunk=128)
    assert full.shape == (rows, dim)
    bound = (6.0 / (rows + dim)) ** 0.5
    assert float(full.abs().max()) <= bound and float(full.abs().max()) > 0.9 * bound
    assert abs(float(full.mean())) < 0.05 * bound and abs(float(full.std()) - bound / 3 ** 0.5) < 0.05 * bound
    ids = torch.tensor([999, 0, 127, 128, 129, 640, 5, 5, 998])
    assert torch.equal(pkg.dist.xavier_rows(ids, rows, dim, seed, 'cpu', chunk=128), full.index_select(0, ids))
    assert not torch.equal(pkg.dist.xavier_rows(ids, rows, dim, seed + 1, 'cpu', chunk=128), full.index_select(0, ids))
    assert pkg.dist.xavier_rows(torch.zeros(0, dtype=torch.int64), rows, dim, seed, 'cpu').shape == (0, dim)


def test_checkpoint_written_under_numpy_1_loads(pkg, tmp_path):
    """a checkpoint the reference wrote under numpy 1.x pickles `measure` through
    numpy.core.multiarray.scalar (numpy 2.x: numpy._core...). Both spellings are on the weights-only allow-list:
    rewrite the GLOBAL in data.pkl to the other spelling and load through the no-code loader."""
    import zipfile
    g = golden('toy_small')
    dl, params = _loader(pkg, g)
    model = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
    src = os.path.join(tmp_path, 'np2.ckpt')
    torch.save({'state_dict': model.state_dict(), 'optim_dict': {}, 'measure': np.round(np.float64(0.25), 5)}, src)
    for old, new in ((b'numpy._core.multiarray', b'numpy.core.multiarray'), (b'numpy.core.multiarray', b'numpy._core.multiarray')):
        dst = os.path.join(tmp_path, 'other.ckpt')
        hits = 0
        with zipfile.ZipFile(src) as zin, zipfile.ZipFile(dst, 'w', zipfile.ZIP_STORED) as zout:
            for item in zin.infolist():
                data = zin.read(item.filename)
                if item.filename.endswith('data.pkl'):
                    hits += data.count(old)
                    data = data.replace(old, new)
                zout.writestr(item.filename, data)
        if not hits:
            continue                      # this numpy writes the other spelling
        fresh = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
        assert pkg.utils.load_checkpoint(dst, fresh) == 0.25
        return
    raise AssertionError('data.pkl names neither numpy.core nor numpy._core')


def test_resumed_optimizer_state_follows_the_table_rows(pkg, tmp_path):
    """resume = load_state_dict, then utils.load_checkpoint's optimizer load, then the first
    encode() lays the tables out in slot order. The Adam moments must move with the rows: one step after the resume
    equals the same step of the uninterrupted run, bit for bit, under a DIFFERENT slot layout."""
    import types
    g = golden('toy_small')
    dl, params = _loader(pkg, g)

    def fake_csr(seed):
        perm = torch.randperm(2 * dl.num_edge, generator=torch.Generator().manual_seed(seed))
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel())
        return types.SimpleNamespace(perm=perm, inv_perm=inv)

    def step(model, opt, csr, seed):
        # the same reference-order gradient for every parameter, laid out like the tables
        gen = torch.Generator().manual_seed(seed)
        tables = {id(p) for _, p in model._edge_tables()}
        for p in model.parameters():
            gr = torch.randn(p.shape, generator=gen)
            p.grad = gr.index_select(0, csr.perm) if id(p) in tables else gr
        opt.step()

    model = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
    model.load_state_dict(g.state_dict())
    opt = model.attach_optimizer(torch.optim.Adam(model.parameters(), lr=1e-2))
    csr_a = fake_csr(1)
    model._use_slot_order(csr_a)
    step(model, opt, csr_a, 10)
    step(model, opt, csr_a, 11)
    pkg.utils.save_checkpoint({'state_dict': model.state_dict(), 'optim_dict': model.optimizer_state_dict(opt), 'measure': 0.5},
                              False, str(tmp_path))
    step(model, opt, csr_a, 12)                       # the uninterrupted run's next step

    resumed = pkg.MGCN(dl.num_entity, dl.num_relation, dl.num_edge, params)
    ropt = torch.optim.Adam(resumed.parameters(), lr=1e-2)
    pkg.utils.load_checkpoint(os.path.join(tmp_path, 'last.ckpt'), resumed, ropt)      # slot layout unknown at this point
    csr_b = fake_csr(2)
    resumed._use_slot_order(csr_b)                    # what the first encode() does
    step(resumed, ropt, csr_b, 12)
    want, got = model.state_dict(), resumed.state_dict()
    for k in want:
        assert torch.equal(want[k], got[k]), k
    # a layout switch in mid-training (a graph with permuted edge ids -> reference order) keeps them tied too
    resumed._use_reference_order()
    model._use_slot_order(csr_b)
    ident = types.SimpleNamespace(perm=torch.arange(2 * dl.num_edge), inv_perm=torch.arange(2 * dl.num_edge))
    step(resumed, ropt, ident, 13)
    step(model, opt, csr_b, 13)
    want, got = model.state_dict(), resumed.state_dict()
    for k in want:
        assert torch.equal(want[k], got[k]), k


def test_workgroup_bounds_balance_the_work(pkg):
    """GraphCSR.workgroup_bounds (the elastic fused launch's per-workgroup runs): strictly increasing from 0 to the
    range's length, at most `groups` runs, and no run carries much more than the mean work (slots + 8 per row)."""
    from oracle import mgcn_oracle as oracle
    N, R, E = 3000, 5, 40000
    tri = oracle.synthetic_triples(N, R, E, seed=3, zipf=1.1)
    ei, ea = oracle.build_edge_list(tri, R)
    csr = pkg.GraphCSR(N, 2 * R + 1, torch.from_numpy(ei), torch.from_numpy(ea)[0], torch.device('cpu'))
    rp = csr.rowptr.to(torch.int64)
    for n0, n1, groups in ((0, N, 256), (100, 2900, 64), (5, 25, 256), (7, 8, 4)):
        b = csr.workgroup_bounds(n0, n1, groups, min_gain=0.0).to(torch.int64)
        n = n1 - n0
        assert b[0] == 0 and b[-1] == n and bool((b[1:] > b[:-1]).all()) and b.numel() - 1 == min(groups, n)
        work = (rp[0, n0 + b[1:]] - rp[0, n0 + b[:-1]]) + (rp[1, n0 + b[1:]] - rp[1, n0 + b[:-1]]) + 8 * (b[1:] - b[:-1])
        rowmax = int(((rp[0, n0 + 1:n1 + 1] - rp[0, n0:n1]) + (rp[1, n0 + 1:n1 + 1] - rp[1, n0:n1])).max()) + 8
        assert int(work.max()) <= float(work.sum()) / (b.numel() - 1) + rowmax      # mean + one row
    assert csr.workgroup_bounds(0, N, 128) is not None                              # 32-row equal runs fill 94 of 128 groups
    uni = oracle.synthetic_triples(N, R, E, seed=4, zipf=0.0)
    ei2, ea2 = oracle.build_edge_list(uni, R)
    csr2 = pkg.GraphCSR(N, 2 * R + 1, torch.from_numpy(ei2), torch.from_numpy(ea2)[0], torch.device('cpu'))
    assert csr2.workgroup_bounds(0, N, 4) is None                                   # uniform tails, long runs: nothing to gain


def test_fused_kernel_generation_rule(pkg):
    """mgcn_fused_kernel_generation (no device work): wide shapes take the elastic kernel (generation 3); a lockstep shape
    takes it too when the caller brings work-balanced runs, its lockstep tiling has fewer than two tiles per CU and O > 128
    (FB15k-237: 182 tiles of 80 rows), and stays on the lockstep kernel (generation 2) otherwise (WN18RR: 512 tiles).
    Generation 4 (round 4's experiment) is never dispatched: `tune` only. Packing sizes per generation."""
    gen = pkg._native.lib().mgcn_fused_kernel_generation
    assert gen(100, 200, 40943, 1) == 2 and gen(200, 200, 40943, 0) == 2
    assert gen(100, 200, 14541, 1) == 3 and gen(100, 200, 14541, 0) == 2
    assert gen(100, 64, 14541, 1) == 2                      # O <= 128: the two kernels read different packings
    assert gen(512, 512, 250000, 0) == 3 and gen(512, 200, 250000, 1) == 3 and gen(200, 256, 40943, 0) == 3
    nb = pkg._native.lib().mgcn_packed_weights_bytes_gen
    assert nb(0, 100, 200) == nb(2, 100, 200) == nb(3, 100, 200) == 12 * 13 * 3 * 64 * 16    # generations 2 and 3: one packing
    assert nb(4, 100, 200) == 10 * 13 * 3 * 64 * 16 and nb(4, 200, 200) == 19 * 13 * 3 * 64 * 16   # K = 3 D padded ONCE
    assert nb(0, 512, 512) == nb(3, 512, 512)


def test_hub_fold_waits_for_its_row_stores_before_the_counter(pkg, tmp_path):
    """in agg_hub_kernel the write-through (sc1) chunk-sum / span-total stores must be acknowledged before
    the agent-scope counter increment that tells the last arriver to read them; a workgroup-scope release fence emits no
    wait for global memory on gfx950, so the wait is written out — checked here in the device ISA: between any vector
    store and a following global_atomic_add there is an `s_waitcnt vmcnt(0)`."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, 'kgc-gcn_amd', 'csrc', 'aggregate.hip')
    asm = str(tmp_path / 'aggregate.s')
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-S', '--cuda-device-only', '-o', asm,
                    src], check=True, cwd=str(tmp_path), stderr=subprocess.DEVNULL)
    kernels, cur = {}, None
    for line in open(asm):
        m = re.match(r'^(_Z\w*agg_hub_kernel\w*):', line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
        elif line.startswith('.Lfunc_end'):
            cur = None
        elif cur is not None:
            cur.append(line.strip())
    assert kernels, 'no agg_hub_kernel in the ISA'
    for name, body in kernels.items():
        atomics = [i for i, l in enumerate(body) if l.startswith('global_atomic_add')]
        assert atomics, name
        for i in atomics:
            waited = False
            for l in reversed(body[:i]):            # walk back to the previous vector-memory store: a full wait must come first
                if re.match(r's_waitcnt vmcnt\(0\)', l):
                    waited = True
                    break
                if l.startswith(('global_store', 'global_atomic', 'scratch_store', 'buffer_store')):
                    break
            assert waited, '%s: global_atomic_add at ISA line %d is not behind s_waitcnt vmcnt(0)' % (name, i)
