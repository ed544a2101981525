"""Entity-sharded scoring and ranking across the GPUs of one node: one process per GPU,
torch.distributed ("nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU rehearsal tests).

Every rank keeps rows [row0, row0 + n_local) of the encoder output (its shard of the entity table), the matching
slice of the decoder bias and a replica of the (small) filter index. Per step each rank brings its own block of B
queries; the exchange is
    all-gather   x [B, O], obj [B], key [B]          (query embeddings, 100 KB per rank: latency-bound)
    all-reduce   target [W*B] f32  (sum; exactly one rank — the owner of obj — contributes a non-zero, so exact)
    all-reduce   counts [W*B, 3] int64 (sum; integers, so sharded ranks are bit-identical to 1-GPU ranks)
and the three local kernels (target, filter bits, score+filter+count) run on the shard. There is no collective
in the aggregation itself when every rank holds the whole graph (FB15k-237: 12 MB of encoder output).
"""
import torch
import torch.distributed as dist

from . import _native


def shard_bounds(n, world):
    """Contiguous row ranges of equal length ceil(n / world) (the last ones shorter or empty): rank r owns
    [b[r], b[r+1]). Equal chunks let a layer output be exchanged with one all-gather."""
    chunk = (n + world - 1) // world
    return [min(r * chunk, n) for r in range(world + 1)]


class FilterIndex(object):
    """Known (subject, relation) -> tails as a sorted key array + CSR (the loader's sr2o over train+valid+test,
    data_loader.py:80-96), for building the evaluation filter on the device instead of dense [B, N] label rows."""

    def __init__(self, keys, ptr, tails, num_rel_ids):
        self.keys, self.ptr, self.tails, self.num_rel_ids = keys, ptr, tails, int(num_rel_ids)

    @classmethod
    def from_known(cls, known, num_rel_ids):
        items = sorted(((s * num_rel_ids + r), sorted(ts)) for (s, r), ts in known.items())
        keys = torch.tensor([k for k, _ in items], dtype=torch.int64)
        lens = torch.tensor([len(t) for _, t in items], dtype=torch.int64)
        ptr = torch.zeros(len(items) + 1, dtype=torch.int64)
        ptr[1:] = torch.cumsum(lens, 0)
        tails = torch.tensor([t for _, ts in items for t in ts], dtype=torch.int32)
        return cls(keys, ptr, tails, num_rel_ids)

    def to(self, device):
        self.keys, self.ptr, self.tails = self.keys.to(device), self.ptr.to(device), self.tails.to(device)
        return self

    def query_keys(self, sub, rel):
        return sub.to(torch.int64) * self.num_rel_ids + rel.to(torch.int64)


_INTO_TENSOR = {}


def _into_tensor_ok(group, like):
    """Whether the group's backend has all_gather_into_tensor — decided ONCE per (backend, device type) by a one-element probe
    that every rank of the group runs at the same point (the first exchange), never by catching an error of a real
    collective: a rank whose collective fails for its own reasons (an RCCL error, a size mismatch) must raise, not
    quietly switch to another collective than its peers are in."""
    key = (dist.get_backend(group), like.device.type)
    if key not in _INTO_TENSOR:
        ok = hasattr(dist, 'all_gather_into_tensor')
        if ok:
            world = dist.get_world_size(group)
            try:
                dist.all_gather_into_tensor(like.new_zeros(world), like.new_zeros(1), group=group)
            except (RuntimeError, NotImplementedError):   # a capability of the backend: the same answer on every rank
                ok = False
        _INTO_TENSOR[key] = ok
    return _INTO_TENSOR[key]


def _gather_into(out, t, group):
    """out [world * n, ...] <- every rank's t [n, ...], in rank order: ONE collective writing straight into `out`
    (all_gather_into_tensor; the list form costs a staging copy per rank on RCCL — 20.5 GB per layer at BASELINE
    configs[4]). Backends without it take the list form over views of `out`. Every rank passes the same n; n = 0 is no
    collective at all (nothing to exchange, on any rank). Errors of the collective propagate."""
    t = t.contiguous()
    if t.size(0) == 0:
        return out
    if _into_tensor_ok(group, t):
        dist.all_gather_into_tensor(out, t, group=group)
    else:
        dist.all_gather(list(out.chunk(out.size(0) // t.size(0), dim=0)), t, group=group)
    return out


def _gather(t, group, world):
    return _gather_into(t.new_empty((world * t.size(0),) + tuple(t.shape[1:])), t, group)


def sharded_rank_counts(x, qkey, obj, ent_shard, bias_shard, row0, filt, group=None, kernels=_native):
    """Filtered-rank counts of this rank's B queries against the WHOLE entity table, which is sharded by rows.
    x [B, O] query embeddings (ConvE trunk output), qkey [B] filter keys, obj [B] target entity (global id).
    Returns (counts [B, 3] int64 = gt / ties_lower / ties, target [B] f32). All ranks must pass the same B."""
    world = dist.get_world_size(group) if (group is not None or dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    B = x.size(0)
    if world > 1:
        x_all, key_all, obj_all = _gather(x, group, world), _gather(qkey, group, world), _gather(obj, group, world)
    else:
        x_all, key_all, obj_all = x.contiguous(), qkey, obj
    n_local = ent_shard.size(0)
    target = torch.zeros(x_all.size(0), dtype=torch.float32, device=x.device)
    kernels.score_target(x_all, ent_shard, bias_shard, obj_all, ent_row0=row0, out=target)
    if world > 1:
        dist.all_reduce(target, op=dist.ReduceOp.SUM, group=group)
    mask = kernels.filter_mask(key_all, filt.keys, filt.ptr, filt.tails, n_local, ent_row0=row0)
    counts = kernels.score_rank(x_all, ent_shard, bias_shard, obj_all, target, mask=mask, ent_row0=row0)
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts[rank * B:(rank + 1) * B], target[rank * B:(rank + 1) * B]


def xavier_rows(edge_ids, num_rows, dim, seed, device, chunk=1 << 16):
    """Rows `edge_ids` (reference edge ids, int64) of a [num_rows, dim] xavier-uniform table (utils.get_param's
    initialiser, utils.py:113-118) that is DEFINED chunk-wise: rows [c * chunk, (c + 1) * chunk) come from a generator
    seeded with (seed, c). Any rank can therefore materialise exactly the rows it owns, in any order, with one chunk
    of scratch — the whole table (410 GB at configs[4]) never exists anywhere."""
    bound = (6.0 / (num_rows + dim)) ** 0.5
    edge_ids = edge_ids.to(device)
    out = torch.empty((edge_ids.numel(), dim), dtype=torch.float32, device=device)
    which = torch.div(edge_ids, chunk, rounding_mode='floor')
    for c in torch.unique(which).tolist():
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed) * 1000003 + int(c))
        block = (torch.rand((chunk, dim), generator=gen, device=device) * 2 - 1) * bound
        sel = (which == c).nonzero(as_tuple=True)[0]
        out[sel] = block.index_select(0, edge_ids.index_select(0, sel) - c * chunk)
    return out


@torch.no_grad()
def shard_model_tables(model, csr, n0, n1, source):
    """Fill a model built with params.edge_table_rows = sum(csr.shard_slot_counts(n0, n1)) with the rows of destinations
    [n0, n1): in-half slots, out-half slots, hub slots (slot order). `source(layer, edge_ids) -> rows [len, D]` returns
    table rows by reference edge id — e.g. lambda l, ids: xavier_rows(ids, 2E, D_l, seed + l, device), or a slice of
    a state dict that is streamed from disk. Nothing of size [2E, D] is allocated."""
    (i0, i1), (o0, o1), (h0, h1) = csr._shard_bounds(n0, n1)
    ids = torch.cat([csr.perm[i0:i1], csr.perm[o0:o1], csr.perm[h0:h1]])
    tables = [model.edge_embeddings] + list(model.edge_embeddings_extra)
    for li, t in enumerate(tables):
        if t.size(0) != ids.numel():
            raise _native.NativeError('shard_model_tables: table %d has %d rows, destinations [%d, %d) need %d'
                                      % (li, t.size(0), n0, n1, ids.numel()))
        t.data.copy_(source(li, ids).to(t.device))
    model._edge_shard = (csr, int(n0), int(n1))
    model._slot_csr = None
    model._enc_cache = None
    return model


def encode_layer_rows(layer, csr, x, rel, table_shard, n0, n1, ee_sub, out=None):
    """Rows [n0, n1) of one layer's eval output (model.py:82-106) from this rank's table shard: the fused launch where
    the shape allows, else the aggregation + dense launches on the range. `out` [n1 - n0, O] optional."""
    O, bn = layer.out_channels, layer.ent_bn
    if out is None:
        out = torch.empty((n1 - n0, O), dtype=torch.float32, device=x.device)
    wcat, wpack = layer.derived_weights()
    x, rel = x.contiguous(), rel.contiguous()
    fused = wpack is not None
    if fused:
        try:
            _native.layer_fwd_fused(csr, x, rel, layer.loop_rel.reshape(-1), table_shard, True, layer.loop_edge.reshape(-1),
                                    wpack, O, layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, out,
                                    node_range=(n0, n1), ee_sub=ee_sub)
        except _native.FusedUnsupported:      # e.g. a misaligned operand: the two launches on the range, as MGCNConv.forward
            fused = False
    if not fused and n1 > n0:
        # the aggregate of the range only ([n1 - n0, 3D]); the kernel writes rows by global node id, hence the offset view
        agg = torch.empty((n1 - n0, 3 * layer.in_channels), dtype=torch.float32, device=x.device)
        _native.aggregate_fwd(csr, x, rel, table_shard, True, layer.loop_edge.reshape(-1), agg, loop_rel=layer.loop_rel.reshape(-1),
                              node_range=(n0, n1), ee_sub=ee_sub, out_row0=n0)
        _native.dense_bn_tanh_fwd(agg, wcat, layer.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, out)
    return out


@torch.no_grad()
def encode_sharded(model, graph, group=None):
    """Destination-partitioned encoder: rank r computes rows [b_r, b_{r+1}) of every layer's output,
    reading only ITS shard of the slot-ordered per-edge tables (1/W of their bytes — the table is what does not fit one
    GPU at 10^8 triples); every destination's sum is formed wholly on one rank, so the rows are bit-identical to the
    single-GPU ones. Sources are arbitrary, so each layer output is all-gathered (RCCL) before the next layer / the
    scorer reads it. Returns (all_ent [N, O], all_rel [2R, O]) complete on every rank.
    A model built with params.edge_table_rows (dist.shard_model_tables) holds nothing but its shard; a model with
    whole tables is sliced once per table version (a convenience for small graphs: no memory is saved then)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    model.eval()
    edge_type, edge_ids = graph.edge_attr
    layers = [model.conv1] + list(model.conv1_extra)
    tables = [model.edge_embeddings] + list(model.edge_embeddings_extra)
    csr = graph.csr(model.relation_embedding.size(0) + 1)
    N = csr.num_nodes
    b = csr.balanced_bounds(world)          # equal work (slots + nodes) per rank, not equal node counts (degree skew)
    n0, n1, chunk = b[rank], b[rank + 1], max(b[r + 1] - b[r] for r in range(world))
    if model._edge_shard is not None:
        ent_identity, edge_identity = model._graph_facts(graph)
        if not (ent_identity and edge_identity):
            # shard_model_tables filled the shard by csr.perm = edge-list POSITIONS: only right when edge k has id k
            raise _native.NativeError('encode_sharded: a model that holds a table shard needs a graph whose entity and edge ids '
                                      'are the identity (data_loader.py:113,147-149 builds them so)')
        if model._edge_shard[0] is not csr or model._edge_shard[1:] != (n0, n1):
            raise _native.NativeError('encode_sharded: the model holds the table shard of destinations %s, this rank owns (%d, %d)'
                                      % (model._edge_shard[1:], n0, n1))
        shards = [t.detach() for t in tables]
    else:
        ent_identity, edge_identity = model._graph_facts(graph)
        if world == 1 or not (ent_identity and edge_identity):
            return model.encode(graph)                          # nothing to partition / ids that need gathers: replicated
        model._use_slot_order(csr)
        cache = model.__dict__.setdefault('_ee_shard_cache', {})
        shards = []
        for li, table in enumerate(tables):
            key = (li, n0, n1, id(csr))
            hit = cache.get(key)
            if hit is None or hit[0] != table._version or hit[1].device != table.device:
                hit = (table._version, csr.edge_table_shard(table.detach(), n0, n1))
                cache[key] = hit
            shards.append(hit[1])
    x, rel = model.entity_embedding.detach(), model.relation_embedding.detach()
    ee_sub = csr.shard_ee_sub(n0, n1)
    for layer, shard in zip(layers, shards):
        local = torch.zeros((chunk, layer.out_channels), dtype=torch.float32, device=x.device)
        encode_layer_rows(layer, csr, x, rel, shard, n0, n1, ee_sub, out=local[:n1 - n0])
        if world > 1:
            full = torch.empty((world * chunk, layer.out_channels), dtype=torch.float32, device=x.device)
            _gather_into(full, local, group)                                    # equal (padded) chunks, gathered in place
            if all(b[r + 1] - b[r] == chunk for r in range(world - 1)):
                x = full[:N]
            else:                                                              # drop each rank's padding rows
                x = torch.cat([full[r * chunk:r * chunk + b[r + 1] - b[r]] for r in range(world)], dim=0)
        else:
            x = local[:N]
        rel = _native.matmul(rel.contiguous(), layer.rels_weight)
    return x, rel


@torch.no_grad()
def evaluate_sharded(model, graph, queries, filt, batch_size=None, group=None, trunk_chunk=2048, shard_encoder=False,
                     parts=None):
    """Filtered MR / MRR / hits@{1,3,10} of `queries` ([Q, 3] int64: subject, relation id, object; both directions
    already expanded, as the loader's *_tail + *_head lists) with the entity table sharded over the group.

    Rank r owns the contiguous query range [c_r, c_{r+1}) (padded to a common length so every rank runs the same
    collectives). The queries go to the device once, the ConvE trunk runs over them in chunks of `trunk_chunk`, and the
    exchange happens ONCE for the whole evaluation: one all-gather of the query embeddings / keys / objects, one
    all-reduce of the target scores, one all-reduce of the integer counts. The score + filter + count kernel takes ALL
    queries in one launch whenever their filter bits fit 512 MB (one launch of 6 268 queries costs 0.8 ms, 49 launches
    of 128 cost 1.6 ms; the result does not depend on the split); `batch_size` forces blocks of that many queries (the
    reference's 128, main.py:117, for like-for-like timing). `shard_encoder=True` also partitions the encoder by
    destination (encode_sharded). `parts`: a dict that receives the wall-clock seconds of the encoder, the ConvE trunk
    (stock torch, out of scope), the exchange and the HIP kernels — it adds a device synchronisation per part, so the
    evaluation's total is NOT to be timed with it."""
    import time
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    model.eval()
    clock = [None]

    def lap(name):
        if parts is not None:
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            now = time.perf_counter()
            if clock[0] is not None and name is not None:
                parts[name] = parts.get(name, 0.0) + now - clock[0]
            clock[0] = now
    lap(None)
    # encoder: replicated (default; 12-33 MB of output at FB15k-237 / WN18RR) or destination-partitioned
    all_ent, all_rel = encode_sharded(model, graph, group) if shard_encoder else model.encode(graph)
    lap('encoder_s')
    N = all_ent.size(0)
    b = shard_bounds(N, world)
    ent_shard = all_ent[b[rank]:b[rank + 1]].contiguous()
    bias_shard = model.conv2.bias[b[rank]:b[rank + 1]].contiguous()
    dev = all_ent.device
    Q = queries.size(0)
    per = (Q + world - 1) // world
    lo, hi = min(rank * per, Q), min((rank + 1) * per, Q)
    real = hi - lo
    q = torch.zeros((per, 3), dtype=torch.int64, device=dev)   # padding rows are harmless queries, dropped below
    q[:real] = queries[lo:hi].to(dev)
    sub, rel, obj = q[:, 0], q[:, 1], q[:, 2].contiguous()
    x = torch.cat([model.conv2.trunk(all_ent.index_select(0, sub[i:i + trunk_chunk]),
                                     all_rel.index_select(0, rel[i:i + trunk_chunk]))
                   for i in range(0, per, trunk_chunk)], dim=0) if per > 0 else all_ent.new_zeros((0, all_ent.size(1)))
    keys = filt.query_keys(sub, rel)
    lap('trunk_s')
    if world > 1:       # every rank's (padded) queries to every rank, once
        x_all, key_all, obj_all = _gather(x, group, world), _gather(keys, group, world), _gather(obj, group, world)
    else:
        x_all, key_all, obj_all = x.contiguous(), keys, obj
    lap('exchange_s')
    total, n_local = x_all.size(0), ent_shard.size(0)
    target = torch.zeros(total, dtype=torch.float32, device=dev)
    counts = torch.zeros((total, 3), dtype=torch.int64, device=dev)
    if total > 0 and n_local > 0:
        _native.score_target(x_all, ent_shard, bias_shard, obj_all, ent_row0=b[rank], out=target)
    lap('kernels_s')
    if world > 1:
        dist.all_reduce(target, op=dist.ReduceOp.SUM, group=group)   # exactly one rank holds each target entity
    lap('exchange_s')
    words = (n_local + 31) // 32
    fit = max(1, (512 << 20) // max(words * 4, 1))             # queries whose filter bits fit the budget at once
    step = min(total, fit) if not batch_size else int(batch_size)
    mask_all = _native.filter_mask(key_all, filt.keys, filt.ptr, filt.tails, n_local, ent_row0=b[rank]) \
        if (total <= fit and total > 0 and n_local > 0) else None
    for i in range(0, total, max(step, 1)):
        if n_local == 0:
            break
        mask = mask_all[i:i + step] if mask_all is not None else _native.filter_mask(
            key_all[i:i + step], filt.keys, filt.ptr, filt.tails, n_local, ent_row0=b[rank])
        _native.score_rank(x_all[i:i + step], ent_shard, bias_shard, obj_all[i:i + step], target[i:i + step], mask=mask,
                           ent_row0=b[rank], counts=counts[i:i + step])
    lap('kernels_s')
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    lap('exchange_s')
    mine = counts[rank * per:(rank + 1) * per]
    ranks = (1 + mine[:, 0] + mine[:, 1]).double()[:real]
    sums = torch.zeros(13, dtype=torch.float64, device=dev)    # count, sum rank, sum 1/rank, hits@1..10
    sums[0] = real
    if real > 0:
        sums[1] = ranks.sum()
        sums[2] = (1.0 / ranks).sum()
        sums[3:] = (ranks.view(-1, 1) <= torch.arange(1, 11, device=dev, dtype=torch.float64)).sum(0)
    if world > 1:
        dist.all_reduce(sums, group=group)
    sums = sums.tolist()
    lap('metrics_s')
    if dev.type == 'cuda':
        _native.check_fused_status(dev)         # (the host has just waited for the results: the fused launches' status word)
    count = sums[0]
    res = {'count': count, 'mr': sums[1] / count, 'mrr': sums[2] / count}
    for k in (1, 3, 10):
        res['hits@%d' % k] = sums[2 + k] / count
    return res
