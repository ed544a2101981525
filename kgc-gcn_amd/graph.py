"""Graph containers: the attribute bag the reference's loops pass around, and the device-side CSR.

`Graph` stands where torch_geometric.data.Data stands in the reference (data_loader.py:151-155,
model.py:25-26, main.py:206): same attribute names, `.to(device)` moves everything IN PLACE
(main.py:206 discards the return value). `GraphCSR` is the build's own layout of the same edges:
per-half CSR by destination with the degree norms folded into 16-byte slot records.
"""
import torch

from . import _native


class GraphCSR(object):
    """Device-resident slot arrays produced by mgcn_csr_build_host (include/mgcn_hip.h (1))."""

    _FIELDS = ('rowptr', 'rec', 'perm', 'hubinfo', 'chunks', 'slot_dst', 'mirror', 'typeptr', 'typeslots')

    def __init__(self, num_nodes, num_rel_rows, edge_index, edge_type, device, with_backward=True, hub_threshold=None,
                 hub_chunk=None):
        host = _native.csr_build_host(num_nodes, num_rel_rows, edge_index, edge_type, with_backward, hub_threshold,
                                      hub_chunk)
        self.num_chunks = host['num_chunks']                     # hub chunks (0: no destination is a hub)
        self.num_nodes = int(num_nodes)
        self.num_edges_half = int(edge_index.size(1)) // 2
        self.num_rel_rows = int(num_rel_rows)
        self.has_backward = with_backward
        # False for an edge list whose second half is not the first half reversed (only possible at the operator
        # seam): forward, gee and grel are exact, the by-source sums of gx have no mirror map to walk (include (1))
        self.mirrored = bool(with_backward and (host['mirror'].numel() == 0 or int(host['mirror'][0]) >= 0))
        for k in self._FIELDS:
            setattr(self, k, host[k].to(device) if k in host else None)
        self._inv_perm = None
        # hubs sit in node order, so destinations [n0, n1) own chunks [chunkptr[n0], chunkptr[n1]) (host-side index)
        if self.num_chunks:
            per_node = host['hubinfo'][:, :, 1].sum(0, dtype=torch.int64)
            self._chunkptr = torch.cat([per_node.new_zeros(1), per_node.cumsum(0)]).tolist()
            self._chunk_slot0 = host['chunks'][:self.num_chunks, 0].tolist() + [2 * self.num_edges_half]
        else:
            self._chunkptr, self._chunk_slot0 = None, None

    @property
    def device(self):
        return self.rowptr.device

    def to(self, device):
        for k in self._FIELDS:
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, v.to(device))
        self._inv_perm = None
        self.__dict__.pop('_shard_cache', None)      # (holds device tensors of the old device: workgroup_bounds)
        self.__dict__.pop('_hub_partials', None)
        return self

    @property
    def inv_perm(self):
        """reference edge id -> slot."""
        if self._inv_perm is None:
            inv = torch.empty_like(self.perm)
            inv[self.perm] = torch.arange(self.perm.numel(), device=self.perm.device)
            self._inv_perm = inv
        return self._inv_perm

    # -- destination partition ------------------------------------------------------
    def _shard_bounds(self, n0, n1):
        key = (int(n0), int(n1))
        cache = self.__dict__.setdefault('_shard_cache', {})
        if key not in cache:
            rp = self.rowptr[:, [key[0], key[1]]].cpu().tolist()      # [[in0, in1], [out0, out1]] slot positions
            c0, c1 = self.chunk_range(*key)
            hub = (self._chunk_slot0[c0], self._chunk_slot0[c1]) if self.num_chunks else (0, 0)
            cache[key] = ((rp[0][0], rp[0][1]), (rp[1][0], rp[1][1]), hub)
        return cache[key]

    def chunk_range(self, n0, n1):
        """Hub chunks [c0, c1) that belong to destinations [n0, n1)."""
        if not self.num_chunks:
            return 0, 0
        return self._chunkptr[int(n0)], self._chunkptr[int(n1)]

    def balanced_bounds(self, world, node_weight=1.0, align=32):
        """Destination ranges with (about) equal WORK instead of equal node counts ("balance the ranges by
        edge count"): work(n) = slots of n in both halves, hub slots included, + node_weight (the self-loop message and
        the output row). Boundaries are rounded to multiples of `align` (the fused kernel's tile height). Returns
        W + 1 non-decreasing bounds from 0 to N."""
        N, W = self.num_nodes, int(world)
        key = ('balanced', W, float(node_weight), int(align))
        cache = self.__dict__.setdefault('_shard_cache', {})
        if key not in cache:
            rp = self.rowptr.cpu().to(torch.int64)
            work = (rp[0, 1:] - rp[0, :-1]) + (rp[1, 1:] - rp[1, :-1])
            if self.num_chunks:
                ch = self.chunks.cpu().to(torch.int64)
                hub = self.hubinfo.cpu().to(torch.int64)                       # [2, N, 2] (first chunk, count)
                first, cnt = hub[:, :, 0], hub[:, :, 1]
                has = cnt > 0
                beg = ch[first.clamp(min=0), 0]
                end = ch[(first + cnt - 1).clamp(min=0), 1]
                work = work + torch.where(has, end - beg, torch.zeros_like(beg)).sum(0)
            prefix = torch.cat([work.new_zeros(1), torch.cumsum(work.double() + node_weight, 0)])
            total = float(prefix[-1])
            bounds = [0]
            for r in range(1, W):
                cut = int(torch.searchsorted(prefix, torch.tensor(total * r / W, dtype=prefix.dtype)))
                cut = min(N, max(bounds[-1], (cut + align // 2) // align * align))
                bounds.append(cut)
            bounds.append(N)
            cache[key] = bounds
        return list(cache[key])

    def workgroup_bounds(self, n0, n1, groups, row_weight=8, min_gain=1.08):
        """Row offsets (relative to n0) that cut destinations [n0, n1) into at most `groups` contiguous runs of about
        equal WORK for the elastic fused launch (include/mgcn_hip.h (2b) row_bounds_dev): work(row) = its slots in both
        halves + row_weight (the self-loop message, the row's share of the multiply, the output row; a hub's slots are
        summed by the pre-pass, so a hub counts as a row without slots). No run is longer than the equal split rounded up to whole
        16-row tiles (at most 80 rows: one tile) or, past that, to whole 80-row tiles. int32 device tensor [g + 1], strictly increasing from 0 to n1 - n0, or None when the heaviest of the
        EQUAL runs carries less than `min_gain` times the mean work (nothing to gain); cached per (range, groups)."""
        n0, n1, groups = int(n0), int(n1), int(groups)
        key = ('wg', n0, n1, groups, int(row_weight), float(min_gain))
        cache = self.__dict__.setdefault('_shard_cache', {})
        if key not in cache:
            n = n1 - n0
            g = max(1, min(groups, n))
            rp = self.rowptr[:, n0:n1 + 1].cpu().to(torch.int64)
            prefix = (rp[0] - rp[0, 0]) + (rp[1] - rp[1, 0]) + row_weight * torch.arange(n + 1, dtype=torch.int64)
            # equal runs (the kernel's own split: ceil(n / groups) rows rounded up to 16) are kept when their heaviest run is
            # within `min_gain` of the mean: balanced cuts then only add ragged last tiles (configs[4] slice: +2 %)
            rpw = (-(-n // groups) + 15) // 16 * 16
            eq = torch.arange(0, n + rpw, rpw, dtype=torch.int64).clamp(max=n)
            eq_work = prefix[eq[1:]] - prefix[eq[:-1]]
            if float(eq_work.max()) <= min_gain * float(prefix[-1]) / groups:
                cache[key] = None
                return None
            idx = torch.arange(1, g, dtype=torch.int64)
            cuts = torch.searchsorted(prefix, (idx * int(prefix[-1])) // g)
            cuts = torch.minimum(torch.maximum(cuts, idx), n - (g - idx))          # every run keeps at least one row
            cuts = torch.cummax(cuts, 0).values if g > 1 else cuts
            # no run longer than the whole tiles an equal split would take (80-row tiles: a run one row longer costs a
            # whole extra tile — its own pass over the weights): forward and backward clamps, feasible since g * cap >= n
            per = -(-n // g)
            cap = (per + 15) // 16 * 16 if per <= 80 else (per + 79) // 80 * 80      # (the launch sizes its tiles by the same rule)
            b = [0] + cuts.tolist() + [n]
            for i in range(1, g):
                b[i] = min(max(b[i], b[i - 1] + 1), b[i - 1] + cap)
            for i in range(g - 1, 0, -1):
                b[i] = max(b[i], b[i + 1] - cap)
            bounds = torch.tensor(b, dtype=torch.int32)
            cache[key] = bounds.to(self.rowptr.device)
        return cache[key]

    def shard_slot_counts(self, n0, n1):
        (i0, i1), (o0, o1), (h0, h1) = self._shard_bounds(n0, n1)
        return i1 - i0, o1 - o0, h1 - h0

    def shard_ee_sub(self, n0, n1):
        """(ee_sub_in, ee_sub_out, ee_sub_hub) for a table that holds only the slots of destinations [n0, n1):
        the row of absolute slot s in that table is s - ee_sub[region]."""
        (i0, i1), (o0, o1), (h0, h1) = self._shard_bounds(n0, n1)
        return i0, o0 - (i1 - i0), h0 - (i1 - i0) - (o1 - o0)

    def edge_table_shard(self, table_slot_order, n0, n1):
        """Rows of a slot-ordered per-edge table that destinations [n0, n1) need: their in-half slots, their out-half
        slots, then the slots of their hubs (each a contiguous run). 1/W of the table per rank for a balanced partition."""
        (i0, i1), (o0, o1), (h0, h1) = self._shard_bounds(n0, n1)
        return torch.cat([table_slot_order[i0:i1], table_slot_order[o0:o1], table_slot_order[h0:h1]], dim=0).contiguous()

    def norms(self):
        """Per-slot degree norm (f32 view of the record's third word)."""
        return self.rec[:, 2].contiguous().view(torch.float32)


class Graph(object):
    def __init__(self, edge_index=None, edge_attr=None, **kwargs):
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        for k, v in kwargs.items():
            setattr(self, k, v)
        self._csr = {}

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        for c in self._csr.values():
            c.to(device)
        return self

    def csr(self, num_rel_rows, with_backward=True):
        """CSR of this graph's edges for a relation table with `num_rel_rows` rows (cached)."""
        dev = self.edge_index.device
        key = (int(num_rel_rows), self.edge_index.data_ptr(), self.edge_attr.data_ptr(), str(dev))
        hit = self._csr.get(key)
        if hit is None or (with_backward and not hit.has_backward):
            self._csr.clear()
            n = self.num_nodes if getattr(self, 'num_nodes', None) is not None else int(self.edge_index.max()) + 1
            hit = GraphCSR(n, num_rel_rows, self.edge_index, self.edge_attr[0], dev, with_backward)
            self._csr[key] = hit
        return hit


_tensor_csr_cache = {}


def csr_for_tensors(num_nodes, num_rel_rows, edge_index, edge_type):
    """Operator-level seam (MGCNConv.forward called with bare tensors, model.py:82): cache by identity."""
    key = (int(num_nodes), int(num_rel_rows), edge_index.data_ptr(), edge_type.data_ptr(), tuple(edge_index.shape),
           edge_index._version, edge_type._version, str(edge_index.device))
    hit = _tensor_csr_cache.get(key)
    if hit is None:
        if len(_tensor_csr_cache) > 8:
            _tensor_csr_cache.clear()
        hit = GraphCSR(num_nodes, num_rel_rows, edge_index, edge_type, edge_index.device)
        _tensor_csr_cache[key] = hit
    return hit
