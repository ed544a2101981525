"""ctypes binding of libmgcn_hip.so (include/mgcn_hip.h) — the only way the package reaches the GPU.

There is deliberately NO fallback: if the library is missing or a tensor is not resident on a GPU the
call raises. torch is used here for device memory and the current HIP stream only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MGCN_LIB') or os.path.join(_HERE, 'csrc', 'libmgcn_hip.so')   # MGCN_LIB: A/B builds
ABI_VERSION = 4

_lib = None

_i32, _i64, _f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_ptr = ctypes.c_void_p

_SIGNATURES = {
    'mgcn_abi_version': (ctypes.c_int, []),
    'mgcn_last_error': (ctypes.c_char_p, []),
    'mgcn_csr_build_host': (ctypes.c_int, [_i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64,
                                           _ptr] + [_ptr] * 4),
    'mgcn_aggregate_fwd': (ctypes.c_int, [_i64, _i64, _i32, _i32, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i32,
                                          _ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _ptr, _i64, _i64, _i64, _ptr]),
    'mgcn_aggregate_bwd': (ctypes.c_int, [_i64, _i64, _i32, _i32] + [_ptr] * 6 + [_i64, _ptr, _ptr] +
                           [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    'mgcn_aggregate_bwd_workspace': (ctypes.c_size_t, [_i64, _i32, _i32, _i64]),
    'mgcn_dense_bn_tanh_fwd': (ctypes.c_int, [_i64, _i32, _i32, _ptr, _i64] + [_ptr] * 6 + [_f32, _ptr, _i64, _ptr]),
    'mgcn_layer_fwd_fused': (ctypes.c_int, [_i64, _i64, _i32, _i32, _i32, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i32,
                                            _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _ptr, _i64, _i64, _i64,
                                            _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr,
                                            _ptr]),
    'mgcn_pack_weights': (ctypes.c_int, [_i32, _i32, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    'mgcn_packed_weights_bytes': (ctypes.c_size_t, [_i32, _i32]),
    'mgcn_pack_weights_gen': (ctypes.c_int, [_i32, _i32, _i32, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    'mgcn_packed_weights_bytes_gen': (ctypes.c_size_t, [_i32, _i32, _i32]),
    'mgcn_matmul_f32': (ctypes.c_int, [_i64, _i32, _i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr]),
    'mgcn_bn_tanh_train_workspace': (ctypes.c_size_t, [_i64, _i32]),
    'mgcn_bn_tanh_train_fwd': (ctypes.c_int, [_i64, _i32, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _ptr,
                                              _ptr, _ptr, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    'mgcn_bn_tanh_train_bwd': (ctypes.c_int, [_i64, _i32] + [_ptr] * 11 + [ctypes.c_size_t, _ptr]),
    'mgcn_matmul_tn_workspace': (ctypes.c_size_t, [_i64, _i32, _i32]),
    'mgcn_matmul_tn_f32': (ctypes.c_int, [_i64, _i32, _i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, ctypes.c_size_t, _ptr]),
    'mgcn_score_fwd': (ctypes.c_int, [_i32, _i64, _i32, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr]),
    'mgcn_score_target': (ctypes.c_int, [_i32, _i64, _i64, _i32, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr]),
    'mgcn_score_rank': (ctypes.c_int, [_i32, _i64, _i64, _i32, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64,
                                       _ptr, _i64, _ptr, _ptr]),
    'mgcn_filter_mask': (ctypes.c_int, [_i32, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _i64, _ptr]),
    'mgcn_score_bce_partials': (_i64, [_i32, _i64]),
    'mgcn_hub_partial_floats': (_i64, [_i64, _i32]),
    'mgcn_fused_kernel_generation': (ctypes.c_int, [_i32, _i32, _i64, _i32]),
    'mgcn_score_bce_fwd': (ctypes.c_int, [_i32, _i64, _i32, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _f32, _f32, _f32, _ptr,
                                          _i64, _ptr, _ptr]),
    'mgcn_label_rows': (ctypes.c_int, [_i32, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _f32, _f32, _ptr, _i64, _ptr]),
    'mgcn_ingest_open': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(_ptr)]),
    'mgcn_ingest_close': (None, [_ptr]),
    'mgcn_ingest_count': (_i64, [_ptr, _i32]),
    'mgcn_ingest_triples': (ctypes.c_int, [_ptr, _i32, _ptr]),
    'mgcn_ingest_names_bytes': (_i64, [_ptr, _i32]),
    'mgcn_ingest_names': (ctypes.c_int, [_ptr, _i32, _ptr, _ptr]),
    'mgcn_filter_index_build': (ctypes.c_int, [_i64, _ptr, _i64, _ptr, _ptr, _ptr, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
}

EXPORTS = tuple(sorted(_SIGNATURES))


class NativeError(RuntimeError):
    pass


class FusedUnsupported(NativeError):
    """mgcn_layer_fwd_fused returned MGCN_EUNSUPPORTED (misaligned operand, shape outside the kernel): callers take the
    aggregation + dense launches instead."""


def lib():
    """Load (once) and return the shared library; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError = symbol missing from the build
            fn.restype, fn.argtypes = res, args
        if handle.mgcn_abi_version() != ABI_VERSION:
            raise NativeError('libmgcn_hip.so ABI %d, binding expects %d' % (handle.mgcn_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def _check(rc, what):
    if rc != 0:
        raise NativeError('%s failed (%d): %s' % (what, rc, lib().mgcn_last_error().decode()))


def _dev(t, dtype, what, allow_none=False):
    """Borrowed device pointer of tensor `t` (last dimension contiguous)."""
    if t is None:
        if allow_none:
            return None
        raise NativeError('%s: tensor required' % what)
    if not t.is_cuda:
        raise NativeError('%s must live on a GPU (got %s): the M-GCN hot path has no CPU fallback' % (what, t.device))
    if t.dtype != dtype:
        raise NativeError('%s: dtype %s, expected %s' % (what, t.dtype, dtype))
    if t.dim() > 0 and t.numel() > 0 and t.stride(-1) != 1:
        raise NativeError('%s: last dimension must be contiguous' % what)
    return t.data_ptr()


def _ld(t):
    return t.stride(0) if t.dim() == 2 and t.size(0) > 1 else t.size(-1)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _same_device(*ts):
    devs = {t.device for t in ts if t is not None}
    if len(devs) > 1:
        raise NativeError('tensors on different devices: %s' % sorted(map(str, devs)))


# ------------------------------------------------------------------------------------------------
# slots per (half, destination) above which it is a hub. 32: below a gather group's fair share of an FB15k-237 tile (~42 slots per
# half and stage), so that no single row outlasts its stage; measured on that shape's step: 64 -> 0.323, 48 -> 0.311, 32 -> 0.303, 24 -> 0.314,
# 16 -> 0.331 ms (tools/hub_sweep.sh). WN18RR / uniform graphs have no such rows either way.
HUB_THRESHOLD = int(os.environ.get('MGCN_HUB_THRESHOLD', '32'))
HUB_CHUNK = int(os.environ.get('MGCN_HUB_CHUNK', '64'))             # slots per hub chunk


def csr_build_host(num_nodes, num_rel_rows, edge_index, edge_type, with_backward=True, hub_threshold=None,
                   hub_chunk=None):
    """(1) Feeder. edge_index [2, 2E] int64, edge_type [2E] int64 (any device; copied to host).
    Returns a dict of HOST tensors (see mgcn_csr_build_host); 'chunks' is trimmed to the chunks in use."""
    ei = edge_index.detach().to('cpu', torch.int64).contiguous()
    et = edge_type.detach().to('cpu', torch.int64).contiguous()
    if ei.dim() != 2 or ei.size(0) != 2 or ei.size(1) % 2 or et.numel() != ei.size(1):
        raise NativeError('csr_build: edge_index must be [2, 2E] and edge_type [2E]')
    E2, E, N = ei.size(1), ei.size(1) // 2, int(num_nodes)
    thr = HUB_THRESHOLD if hub_threshold is None else int(hub_threshold)
    chk = HUB_CHUNK if hub_chunk is None else int(hub_chunk)
    max_chunks = (E2 // max(chk, 1) + E2 // max(thr, 1) + 2) if thr > 0 else 0
    out = dict(rowptr=torch.empty((2, N + 1), dtype=torch.int32), rec=torch.empty((E2, 4), dtype=torch.int32),
               perm=torch.empty(E2, dtype=torch.int64), hubinfo=torch.empty((2, N, 2), dtype=torch.int32),
               chunks=torch.empty((max(max_chunks, 1), 4), dtype=torch.int32))
    nch = ctypes.c_int64(0)
    if with_backward:
        out.update(slot_dst=torch.empty(E2, dtype=torch.int32), mirror=torch.empty(E2, dtype=torch.int32),
                   typeptr=torch.empty(num_rel_rows + 1, dtype=torch.int32),
                   typeslots=torch.empty(E2, dtype=torch.int32))
    p = lambda k: out[k].data_ptr() if k in out else None
    _check(lib().mgcn_csr_build_host(N, E, int(num_rel_rows), ei.data_ptr(), et.data_ptr(), thr, chk, p('rowptr'),
                                     p('rec'), p('perm'), p('hubinfo'), p('chunks'), max_chunks, ctypes.byref(nch),
                                     p('slot_dst'), p('mirror'), p('typeptr'), p('typeslots')),
           'mgcn_csr_build_host')
    out['num_chunks'] = int(nch.value)
    out['chunks'] = out['chunks'][:max(int(nch.value), 1)].contiguous()
    return out


def _hub_args(csr, d, device, n0, n1):
    """(hubinfo ptr, chunks ptr, chunk_begin, chunk_end, partial tensor) for a launch over destinations [n0, n1). The chunk
    sums and the fold's arrival counters live in a buffer kept on the graph per (width, chunk range): the counters are
    zero when it is made and every launch leaves them zero (include/mgcn_hip.h (2)), and its address is stable for a
    captured launch. Launches that share it must not overlap: a launch on another stream than the last one waits for
    that stream first."""
    c0, c1 = csr.chunk_range(n0, n1)
    if c1 == c0:
        return None, None, 0, 0, None
    cache = csr.__dict__.setdefault('_hub_partials', {})
    key = (int(d), c0, c1, str(device))
    hit = cache.get(key)
    stream = torch.cuda.current_stream(device)
    if hit is None:            # (never evicted: a captured launch keeps the buffer's address; one entry per width and range)
        hit = cache[key] = [torch.zeros(int(lib().mgcn_hub_partial_floats(c1 - c0, int(d))), dtype=torch.float32, device=device),
                            stream]
    elif hit[1] != stream and not torch.cuda.is_current_stream_capturing():
        stream.wait_stream(hit[1])
        hit[1] = stream
    return _dev(csr.hubinfo, torch.int32, 'hubinfo'), _dev(csr.chunks, torch.int32, 'chunks'), c0, c1, hit[0]


def _hub_failed(csr, d, device, n0, n1):
    """A launch that was handed the (width, chunk range) hub buffer returned an error: its fold may have stopped between two
    arrivals and left counters non-zero, which would silently switch the NEXT launch's fold off. Forget the buffer: the next
    launch gets a fresh zeroed one (a captured graph that holds the old address is invalid after a failed launch anyway)."""
    c0, c1 = csr.chunk_range(n0, n1)
    csr.__dict__.get('_hub_partials', {}).pop((int(d), c0, c1, str(device)), None)


def aggregate_fwd(csr, x, rel, ee, ee_in_slot_order, loop_edge, out, loop_rel=None, node_range=None, ee_sub=(0, 0, 0),
                  out_row0=0):
    """(2) out[:, 0:D | D:2D | 2D:3D) = in / out / self-loop aggregates. `csr` is a graph.GraphCSR.
    `rel` is either the whole relation table [num_rel_rows, D] (last row = self-loop row) or, with
    `loop_rel` [D] given separately, its first num_rel_rows-1 rows (no concatenation needed).
    With `node_range` = (n0, n1), `ee` may be that range's shard of the slot-ordered table
    (graph.GraphCSR.edge_table_shard) and `ee_sub` its three slot offsets (GraphCSR.shard_ee_sub). `out_row0`: `out`
    holds rows [out_row0, out_row0 + out.size(0)) of the aggregate (the kernel indexes rows by global node id)."""
    N, E, D = csr.num_nodes, csr.num_edges_half, x.size(1)
    _same_device(csr.rowptr, x, rel, ee, loop_edge, out, loop_rel)
    if loop_rel is None:
        if rel.size(0) != csr.num_rel_rows:
            raise NativeError('aggregate_fwd: rel has %d rows, graph expects %d' % (rel.size(0), csr.num_rel_rows))
        loop_rel, rel = rel[-1], rel[:-1]
    if x.size(0) != N or rel.size(0) != csr.num_rel_rows - 1 or rel.size(1) != D or loop_rel.numel() != D:
        raise NativeError('aggregate_fwd: x %s / rel %s do not match graph (N=%d, rel rows=%d)'
                          % (tuple(x.shape), tuple(rel.shape), N, csr.num_rel_rows))
    ee_sub = tuple(int(v) for v in ee_sub)
    sharded = ee_sub != (0, 0, 0) or (ee is not None and ee.size(0) != 2 * E)
    if ee is not None and not sharded and tuple(ee.shape) != (2 * E, D):
        raise NativeError('aggregate_fwd: per-edge table %s, expected (%d, %d)' % (tuple(ee.shape), 2 * E, D))
    if not rel.is_contiguous() or (ee is not None and not ee.is_contiguous()):
        raise NativeError('aggregate_fwd: relation and per-edge tables must be contiguous')
    modes = 3 if loop_edge is not None else 2
    if loop_edge is not None and loop_edge.numel() != D:
        raise NativeError('aggregate_fwd: loop_edge must have %d elements' % D)
    n0, n1 = (0, N) if node_range is None else (int(node_range[0]), int(node_range[1]))
    if not 0 <= n0 <= n1 <= N:
        raise NativeError('aggregate_fwd: node range (%d, %d) outside [0, %d]' % (n0, n1, N))
    out_row0 = int(out_row0)
    if out.size(1) < modes * D or out_row0 > n0 or out_row0 + out.size(0) < n1 or (out_row0 == 0 and node_range is None and out.size(0) != N):
        raise NativeError('aggregate_fwd: out %s (rows from %d) does not cover destinations [%d, %d) x %d columns'
                          % (tuple(out.shape), out_row0, n0, n1, modes * D))
    if sharded:
        rows = csr.shard_slot_counts(n0, n1)
        if ee is None or not ee_in_slot_order or tuple(ee.shape) != (sum(rows), D) or not ee.is_contiguous() or \
                ee_sub != csr.shard_ee_sub(n0, n1):
            raise NativeError('aggregate_fwd: per-edge shard does not match destinations [%d, %d)' % (n0, n1))
        if ee.numel() == 0:
            ee = x.new_zeros((1, D))
    hub_info, hub_chunks, hub_c0, hub_c1, hub_partial = _hub_args(csr, D, x.device, n0, n1)
    rc = lib().mgcn_aggregate_fwd(
        N, E, D, csr.num_rel_rows, _dev(csr.rowptr, torch.int32, 'rowptr'), _dev(csr.rec, torch.int32, 'rec'),
        _dev(x, torch.float32, 'x'), _ld(x), _dev(rel, torch.float32, 'rel'), _dev(loop_rel, torch.float32, 'loop_rel'),
        _dev(ee, torch.float32, 'ee', True), int(bool(ee_in_slot_order)), _dev(loop_edge, torch.float32, 'loop_edge', True),
        _dev(out, torch.float32, 'out') - out_row0 * _ld(out) * 4, _ld(out), n0, n1, hub_info, hub_chunks, hub_c0, hub_c1,
        _dev(hub_partial, torch.float32, 'partial', True), ee_sub[0], ee_sub[1], ee_sub[2], _stream(x))
    if rc != 0 and hub_partial is not None:
        _hub_failed(csr, D, x.device, n0, n1)
    _check(rc, 'mgcn_aggregate_fwd')
    return out


def aggregate_bwd(csr, x, rel, ee, g, want_gx=True, want_gee=True, want_grel=True):
    """(3) Gradients of aggregate_fwd's first 2D columns w.r.t. x, the per-edge table (slot order) and rel."""
    N, E, D = csr.num_nodes, csr.num_edges_half, x.size(1)
    _same_device(csr.rowptr, x, rel, ee, g)
    if not csr.has_backward:
        raise NativeError('aggregate_bwd: graph was prepared without the backward indices')
    if want_gx and not csr.mirrored:
        raise NativeError('aggregate_bwd: the edge list is not mirror-symmetric (edge e + E is not the reverse of edge e, '
                          'data_loader.py:143-149), so the gradient w.r.t. the layer input cannot be formed from the '
                          'destination runs; forward, per-edge and relation gradients are unaffected')
    if g.size(0) != N or g.size(1) < 2 * D:
        raise NativeError('aggregate_bwd: g %s too small' % (tuple(g.shape),))
    if x.size(0) != N or tuple(rel.shape) != (csr.num_rel_rows, D) or not rel.is_contiguous():
        raise NativeError('aggregate_bwd: x / rel do not match the graph')
    if ee is not None and (tuple(ee.shape) != (2 * E, D) or not ee.is_contiguous()):
        raise NativeError('aggregate_bwd: per-edge table must be contiguous (%d, %d) in slot order' % (2 * E, D))
    gx = torch.empty((N, D), dtype=torch.float32, device=x.device) if want_gx else None
    gee = torch.empty((2 * E, D), dtype=torch.float32, device=x.device) if (want_gee and ee is not None) else None
    grel = torch.empty((csr.num_rel_rows, D), dtype=torch.float32, device=x.device) if want_grel else None
    need_ws = want_grel or (want_gx and csr.num_chunks > 0)
    ws_bytes = lib().mgcn_aggregate_bwd_workspace(E, D, csr.num_rel_rows, csr.num_chunks) if need_ws else 0
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=x.device) if need_ws else None
    hubs = csr.num_chunks > 0
    _check(lib().mgcn_aggregate_bwd(
        N, E, D, csr.num_rel_rows, _dev(csr.rowptr, torch.int32, 'rowptr'), _dev(csr.rec, torch.int32, 'rec'),
        _dev(csr.slot_dst, torch.int32, 'slot_dst'), _dev(csr.mirror, torch.int32, 'mirror'),
        _dev(csr.hubinfo, torch.int32, 'hubinfo') if hubs else None,
        _dev(csr.chunks, torch.int32, 'chunks') if hubs else None, csr.num_chunks,
        _dev(csr.typeptr, torch.int32, 'typeptr'), _dev(csr.typeslots, torch.int32, 'typeslots'),
        _dev(x, torch.float32, 'x'), _ld(x), _dev(rel, torch.float32, 'rel'), _dev(ee, torch.float32, 'ee', True),
        _dev(g, torch.float32, 'g'), _ld(g), _dev(gx, torch.float32, 'gx', True), _dev(gee, torch.float32, 'gee', True),
        _dev(grel, torch.float32, 'grel', True), _dev(ws, torch.float32, 'ws', True), ws_bytes, _stream(x)),
        'mgcn_aggregate_bwd')
    return gx, gee, grel


def dense_bn_tanh_fwd(a, w_cat, bias, bn_mean, bn_var, bn_gamma, bn_beta, eps, out):
    """(4) out = tanh(BN_eval((A @ w_cat) / 3 + bias)), w_cat [3D, O] = W_in, W_out, W_loop stacked by rows."""
    N, O = a.size(0), w_cat.size(1)
    D = w_cat.size(0) // 3
    _same_device(a, w_cat, bias, bn_mean, bn_var, bn_gamma, bn_beta, out)
    if w_cat.dim() != 2 or w_cat.size(0) != 3 * D or not w_cat.is_contiguous():
        raise NativeError('dense_bn_tanh_fwd: w_cat must be contiguous (3D, O)')
    for v in (bn_mean, bn_var, bn_gamma, bn_beta) + ((bias,) if bias is not None else ()):
        if v.numel() != O:
            raise NativeError('dense_bn_tanh_fwd: per-column vectors must have %d elements' % O)
    if a.size(1) < 3 * D or tuple(out.shape) != (N, O):
        raise NativeError('dense_bn_tanh_fwd: a %s / out %s do not match' % (tuple(a.shape), tuple(out.shape)))
    _check(lib().mgcn_dense_bn_tanh_fwd(
        N, D, O, _dev(a, torch.float32, 'a'), _ld(a), _dev(w_cat, torch.float32, 'w_cat'),
        _dev(bias, torch.float32, 'bias', True), _dev(bn_mean, torch.float32, 'bn_mean'),
        _dev(bn_var, torch.float32, 'bn_var'), _dev(bn_gamma, torch.float32, 'bn_gamma'),
        _dev(bn_beta, torch.float32, 'bn_beta'), float(eps), _dev(out, torch.float32, 'out'), _ld(out), _stream(a)),
        'mgcn_dense_bn_tanh_fwd')
    return out


# `tune` argument of mgcn_layer_fwd_fused (include/mgcn_hip.h): 0 = automatic. Read ONCE at import, for A/B tools only (see
# INTEGRATION.md "Environment switches"): bits 0-3 row tiles per tile, 4-7 staging buffers / slots per batch, 8-9 relation
# table in LDS, 10-11 a forced kernel generation (1 = 4, 2, 3), 12-13 columns per slot walk.
FUSED_TUNE = int(os.environ.get('MGCN_FUSED_TUNE', '0'), 0)


def tune_generation(tune=None):
    """Kernel generation a `tune` word forces (0 = the shape's own)."""
    f = ((FUSED_TUNE if tune is None else int(tune)) >> 10) & 3
    return 0 if f == 0 else (4 if f == 1 else f)


_FUSED_STATUS = {}


def fused_status(device):
    """The device's status word handed to every fused launch (mgcn_layer_fwd_fused status_dev): one int32, zero while no
    bounded spin of the generation-3 kernel has run out."""
    key = str(torch.device(device))
    if key not in _FUSED_STATUS:
        _FUSED_STATUS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _FUSED_STATUS[key]


def check_fused_status(device):
    """Synchronising check of the status word (call where the host waits for results anyway: end of an evaluation, a
    benchmark's timed region, tests): raises if a fused launch since the last check reported a spin timeout, and clears it."""
    key = str(torch.device(device))
    if key in _FUSED_STATUS:
        v = int(_FUSED_STATUS[key].item())
        if v:
            _FUSED_STATUS[key].zero_()
            raise NativeError('a fused layer launch on %s reported status %d: a bounded LDS-counter spin ran out '
                              '(layer_fused3.hip), rows of that launch are invalid' % (key, v))
FUSED_ENABLED = os.environ.get('MGCN_FUSED', '1') != '0'


_CU_COUNT = {}


def _cu_count(device):
    key = str(device)
    if key not in _CU_COUNT:
        _CU_COUNT[key] = int(torch.cuda.get_device_properties(device).multi_processor_count)
    return _CU_COUNT[key]


def fused_supported(d_in, d_out):
    """Shapes the one-launch layer kernel handles (else: aggregate_fwd + dense_bn_tanh_fwd)."""
    return FUSED_ENABLED and d_in % 4 == 0 and d_in <= 1024 and d_out % 4 == 0 and d_out <= 512


def pack_weights(w_cat, out=None, generation=None):
    """Stacked [3D, O] weights -> MFMA fragment order for layer_fwd_fused (re-pack whenever a weight changes).
    `generation`: the kernel generation to pack for (None = what MGCN_FUSED_TUNE forces, else the shape's own)."""
    D, O = w_cat.size(0) // 3, w_cat.size(1)
    if w_cat.dim() != 2 or w_cat.size(0) != 3 * D or not w_cat.is_contiguous():
        raise NativeError('pack_weights: w_cat must be contiguous (3D, O)')
    gen = tune_generation() if generation is None else int(generation)
    nbytes = lib().mgcn_packed_weights_bytes_gen(gen, D, O)
    if out is None:
        out = torch.empty(nbytes // 4, dtype=torch.float32, device=w_cat.device)
    if out.numel() * 4 < nbytes:
        raise NativeError('pack_weights: out too small')
    _same_device(w_cat, out)
    _check(lib().mgcn_pack_weights_gen(gen, D, O, _dev(w_cat, torch.float32, 'w_cat'), _dev(out, torch.float32, 'wp'),
                                       out.numel() * 4, _stream(w_cat)), 'mgcn_pack_weights')
    return out


def layer_fwd_fused(csr, x, rel, loop_rel, ee, ee_in_slot_order, loop_edge, w_packed, d_out, bias, bn_mean, bn_var,
                    bn_gamma, bn_beta, eps, out, node_range=None, ee_sub=(0, 0, 0), rels_weight=None, rel_out=None,
                    tune=None, balance=True):
    """(2)+(4) in one launch: out = tanh(BN_eval((aggregates @ W) / 3 + bias)), aggregates kept in LDS.
    `w_packed` = pack_weights(stacked [3D, O] weights). With `node_range` = (n0, n1) only those destinations are
    computed and `out` is [n1 - n0, O]; `ee` may then be this range's shard of the slot-ordered table (see
    graph.GraphCSR.edge_table_shard) with `ee_sub` its three slot offsets (in-half, out-half, hub region).
    `balance`: hand the launch the graph's work-balanced per-workgroup row runs (GraphCSR.workgroup_bounds, one run per
    CU, None when equal runs are balanced already); results do not depend on it."""
    N, E, D, O = csr.num_nodes, csr.num_edges_half, x.size(1), int(d_out)
    n0, n1 = (0, N) if node_range is None else (int(node_range[0]), int(node_range[1]))
    if not 0 <= n0 <= n1 <= N:
        raise NativeError('layer_fwd_fused: node range (%d, %d) outside [0, %d]' % (n0, n1, N))
    ee_sub = tuple(int(v) for v in ee_sub) + (0,) * (3 - len(ee_sub))
    sharded = ee_sub != (0, 0, 0) or (ee is not None and ee.size(0) != 2 * E)
    _same_device(csr.rowptr, x, rel, loop_rel, ee, loop_edge, w_packed, bias, bn_mean, bn_var, bn_gamma, bn_beta, out)
    if x.size(0) != N or tuple(rel.shape) != (csr.num_rel_rows - 1, D) or loop_rel.numel() != D or loop_edge.numel() != D:
        raise NativeError('layer_fwd_fused: x / rel / loop rows do not match the graph')
    if ee is not None and not sharded and (tuple(ee.shape) != (2 * E, D) or not ee.is_contiguous()):
        raise NativeError('layer_fwd_fused: per-edge table must be contiguous (%d, %d)' % (2 * E, D))
    if ee is not None and sharded:
        rows = csr.shard_slot_counts(n0, n1)
        if not ee_in_slot_order or tuple(ee.shape) != (sum(rows), D) or not ee.is_contiguous() or \
                ee_sub != csr.shard_ee_sub(n0, n1):
            raise NativeError('layer_fwd_fused: per-edge shard does not match destinations [%d, %d)' % (n0, n1))
    tune = FUSED_TUNE if tune is None else int(tune)
    if not rel.is_contiguous() or w_packed.numel() * 4 < lib().mgcn_packed_weights_bytes_gen(tune_generation(tune), D, O):
        raise NativeError('layer_fwd_fused: rel must be contiguous and w_packed sized by mgcn_packed_weights_bytes')
    for v in (bn_mean, bn_var, bn_gamma, bn_beta) + ((bias,) if bias is not None else ()):
        if v.numel() != O:
            raise NativeError('layer_fwd_fused: per-column vectors must have %d elements' % O)
    if tuple(out.shape) != (n1 - n0, O):
        raise NativeError('layer_fwd_fused: out must be (%d, %d)' % (n1 - n0, O))
    if (rels_weight is None) != (rel_out is None):
        raise NativeError('layer_fwd_fused: give rels_weight and rel_out together')
    if rel_out is not None:
        _same_device(x, rels_weight, rel_out)
        if tuple(rels_weight.shape) != (D, O) or not rels_weight.is_contiguous() or \
                tuple(rel_out.shape) != (csr.num_rel_rows - 1, O) or not rel_out.is_contiguous():
            raise NativeError('layer_fwd_fused: rels_weight must be contiguous (%d, %d) and rel_out (%d, %d)'
                              % (D, O, csr.num_rel_rows - 1, O))
    if n1 == n0 and rel_out is None:
        return out                                   # an empty destination range: nothing to launch
    if ee is not None and ee.numel() == 0:           # a range whose destinations have no slots: the kernel still wants
        ee = x.new_zeros((1, D))                     # a valid (never read) table pointer
    hub_info, hub_chunks, hub_c0, hub_c1, hub_partial = _hub_args(csr, D, x.device, n0, n1)
    bounds = csr.workgroup_bounds(n0, n1, _cu_count(x.device)) if balance and n1 > n0 else None
    rc = lib().mgcn_layer_fwd_fused(
        N, E, D, O, csr.num_rel_rows, _dev(csr.rowptr, torch.int32, 'rowptr'), _dev(csr.rec, torch.int32, 'rec'),
        _dev(x, torch.float32, 'x'), _ld(x), _dev(rel, torch.float32, 'rel'), _dev(loop_rel, torch.float32, 'loop_rel'),
        _dev(ee, torch.float32, 'ee', True), int(bool(ee_in_slot_order)), _dev(loop_edge, torch.float32, 'loop_edge'),
        _dev(w_packed, torch.float32, 'w_packed'), _dev(bias, torch.float32, 'bias', True),
        _dev(bn_mean, torch.float32, 'bn_mean'), _dev(bn_var, torch.float32, 'bn_var'),
        _dev(bn_gamma, torch.float32, 'bn_gamma'), _dev(bn_beta, torch.float32, 'bn_beta'), float(eps),
        _dev(out, torch.float32, 'out'), _ld(out), n0, n1, int(ee_sub[0]), int(ee_sub[1]), int(ee_sub[2]),
        hub_info, hub_chunks, hub_c0, hub_c1,
        _dev(hub_partial, torch.float32, 'partial', True), _dev(rels_weight, torch.float32, 'rels_weight', True),
        _dev(rel_out, torch.float32, 'rel_out', True), _dev(bounds, torch.int32, 'row_bounds', True),
        bounds.numel() - 1 if bounds is not None else 0, tune, _dev(fused_status(x.device), torch.int32, 'status'),
        _stream(x))
    if rc != 0 and hub_partial is not None:
        _hub_failed(csr, D, x.device, n0, n1)
    if rc == 3:
        raise FusedUnsupported('mgcn_layer_fwd_fused: %s' % lib().mgcn_last_error().decode())
    _check(rc, 'mgcn_layer_fwd_fused')
    return out


def matmul(a, b):
    """C = A @ B on the f32 MFMA tile kernel."""
    _same_device(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.size(1) != b.size(0):
        raise NativeError('matmul: shapes %s @ %s' % (tuple(a.shape), tuple(b.shape)))
    c = torch.empty((a.size(0), b.size(1)), dtype=torch.float32, device=a.device)
    _check(lib().mgcn_matmul_f32(a.size(0), a.size(1), b.size(1), _dev(a, torch.float32, 'a'), _ld(a),
                                 _dev(b, torch.float32, 'b'), _ld(b), _dev(c, torch.float32, 'c'), _ld(c), _stream(a)),
           'mgcn_matmul_f32')
    return c


def matmul_tn(a, b):
    """C = A^T @ B for A [K, M], B [K, N] (the weight gradient dW = aggregate^T g): split-K exact-f32 MFMA kernel."""
    _same_device(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.size(0) != b.size(0):
        raise NativeError('matmul_tn: shapes %s^T @ %s' % (tuple(a.shape), tuple(b.shape)))
    K, M, N = a.size(0), a.size(1), b.size(1)
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    nbytes = lib().mgcn_matmul_tn_workspace(K, M, N)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=a.device)
    _check(lib().mgcn_matmul_tn_f32(K, M, N, _dev(a, torch.float32, 'a'), _ld(a), _dev(b, torch.float32, 'b'), _ld(b),
                                    _dev(c, torch.float32, 'c'), _ld(c), _dev(ws, torch.float32, 'ws'), nbytes, _stream(a)),
           'mgcn_matmul_tn_f32')
    return c


def matmul_tn_supported(m, n):
    return m <= 208 and n <= 256


def bn_tanh_train_fwd(u_in, u_out, u_loop, bias, gamma, beta, running_mean, running_var, momentum, eps):
    """(4t) z = (u_in + u_out + u_loop) / 3 (+ bias); y = tanh(BN_batch(z)). Returns (y, z, save_mean, save_rstd); updates
    the running statistics in place when given."""
    N, O = u_in.shape
    _same_device(u_in, u_out, u_loop, bias, gamma, beta, running_mean, running_var)
    for t in (u_in, u_out, u_loop):
        if tuple(t.shape) != (N, O) or t.stride(1) != 1 or t.stride(0) != u_in.stride(0):
            raise NativeError('bn_tanh_train_fwd: the three products must be [N, O] with the same row stride')
    z = torch.empty((N, O), dtype=torch.float32, device=u_in.device)
    y = torch.empty_like(z)
    mean = torch.empty(O, dtype=torch.float32, device=u_in.device)
    rstd = torch.empty_like(mean)
    nbytes = lib().mgcn_bn_tanh_train_workspace(N, O)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=u_in.device)
    _check(lib().mgcn_bn_tanh_train_fwd(
        N, O, _dev(u_in, torch.float32, 'u_in'), _dev(u_out, torch.float32, 'u_out'), _dev(u_loop, torch.float32, 'u_loop'),
        u_in.stride(0), _dev(bias, torch.float32, 'bias', True), _dev(gamma, torch.float32, 'gamma'), _dev(beta, torch.float32, 'beta'),
        _dev(running_mean, torch.float32, 'running_mean', True), _dev(running_var, torch.float32, 'running_var', True),
        float(momentum), float(eps), _dev(z, torch.float32, 'z'), _dev(y, torch.float32, 'y'), _dev(mean, torch.float32, 'mean'),
        _dev(rstd, torch.float32, 'rstd'), _dev(ws, torch.float32, 'ws'), nbytes, _stream(u_in)), 'mgcn_bn_tanh_train_fwd')
    return y, z, mean, rstd


def bn_tanh_train_bwd(z, y, gy, mean, rstd, gamma):
    """Backward of bn_tanh_train_fwd: (gz [N, O], gu = gz / 3, ggamma [O], gbeta [O])."""
    N, O = z.shape
    _same_device(z, y, gy, mean, rstd, gamma)
    gy = gy.contiguous()
    gz, gu = torch.empty_like(z), torch.empty_like(z)
    gg, gb = torch.empty_like(mean), torch.empty_like(mean)
    nbytes = lib().mgcn_bn_tanh_train_workspace(N, O)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=z.device)
    _check(lib().mgcn_bn_tanh_train_bwd(
        N, O, _dev(z, torch.float32, 'z'), _dev(y, torch.float32, 'y'), _dev(gy, torch.float32, 'gy'), _dev(mean, torch.float32, 'mean'),
        _dev(rstd, torch.float32, 'rstd'), _dev(gamma, torch.float32, 'gamma'), _dev(gz, torch.float32, 'gz'), _dev(gu, torch.float32, 'gu'),
        _dev(gg, torch.float32, 'ggamma'), _dev(gb, torch.float32, 'gbeta'), _dev(ws, torch.float32, 'ws'), nbytes, _stream(z)),
        'mgcn_bn_tanh_train_bwd')
    return gz, gu, gg, gb


def _score_args(x, ent, bias):
    _same_device(x, ent, bias)
    if x.dim() != 2 or ent.dim() != 2 or x.size(1) != ent.size(1) or bias.numel() != ent.size(0):
        raise NativeError('score: x %s, ent %s, bias %s do not match' % (tuple(x.shape), tuple(ent.shape), tuple(bias.shape)))
    return x.size(0), ent.size(0), x.size(1)


def score_fwd(x, ent, bias):
    """(5) score [B, n_local] = sigmoid(x @ ent^T + bias)."""
    B, n, O = _score_args(x, ent, bias)
    out = torch.empty((B, n), dtype=torch.float32, device=x.device)
    _check(lib().mgcn_score_fwd(B, n, O, _dev(x, torch.float32, 'x'), _ld(x), _dev(ent, torch.float32, 'ent'), _ld(ent),
                                _dev(bias, torch.float32, 'bias'), _dev(out, torch.float32, 'score'), _ld(out),
                                _stream(x)), 'mgcn_score_fwd')
    return out


def score_target(x, ent, bias, obj, ent_row0=0, out=None):
    """target[b] = score[b, obj[b]] for the queries whose obj is a row of this shard (others keep `out`)."""
    B, n, O = _score_args(x, ent, bias)
    if obj.numel() != B:
        raise NativeError('score_target: obj must have %d elements' % B)
    if out is None:
        out = torch.zeros(B, dtype=torch.float32, device=x.device)
    _check(lib().mgcn_score_target(B, n, int(ent_row0), O, _dev(x, torch.float32, 'x'), _ld(x),
                                   _dev(ent, torch.float32, 'ent'), _ld(ent), _dev(bias, torch.float32, 'bias'),
                                   _dev(obj, torch.int64, 'obj'), _dev(out, torch.float32, 'target'), _stream(x)),
           'mgcn_score_target')
    return out


def score_rank(x, ent, bias, obj, target, label=None, ent_row0=0, counts=None, mask=None):
    """counts [B, 3] int64 += (gt, ties_lower, ties) over this shard's entities. Filter = dense `label` rows [B, n]
    (reference loader) or the bit-packed `mask` [B, ceil(n/32)] int32 from filter_mask()."""
    B, n, O = _score_args(x, ent, bias)
    if (label is None) == (mask is None):
        raise NativeError('score_rank: give exactly one of label / mask')
    if obj.numel() != B or target.numel() != B:
        raise NativeError('score_rank: obj/target do not match batch %d' % B)
    if label is not None and (label.dim() != 2 or label.size(0) != B or label.size(1) != n):
        raise NativeError('score_rank: label must be (%d, %d)' % (B, n))
    if mask is not None and (mask.dim() != 2 or mask.size(0) != B or mask.size(1) < (n + 31) // 32 or not mask.is_contiguous()):
        raise NativeError('score_rank: mask must be contiguous (%d, >= %d)' % (B, (n + 31) // 32))
    if counts is None:
        counts = torch.zeros((B, 3), dtype=torch.int64, device=x.device)
    _same_device(x, ent, bias, obj, target, label, mask, counts)
    _check(lib().mgcn_score_rank(B, n, int(ent_row0), O, _dev(x, torch.float32, 'x'), _ld(x),
                                 _dev(ent, torch.float32, 'ent'), _ld(ent), _dev(bias, torch.float32, 'bias'),
                                 _dev(obj, torch.int64, 'obj'), _dev(target, torch.float32, 'target'),
                                 _dev(label, torch.float32, 'label', True), _ld(label) if label is not None else 0,
                                 _dev(mask, torch.int32, 'mask', True), mask.size(1) if mask is not None else 0,
                                 _dev(counts, torch.int64, 'counts'), _stream(x)), 'mgcn_score_rank')
    return counts


def filter_mask(qkey, keys, ptr, tails, n_local, ent_row0=0, out=None):
    """Bit-packed filter rows [B, ceil(n_local/32)] int32 for queries with keys `qkey` (see mgcn_filter_mask)."""
    B, words = qkey.numel(), (int(n_local) + 31) // 32
    if out is None:
        out = torch.empty((B, words), dtype=torch.int32, device=qkey.device)
    if out.size(0) != B or out.size(1) < words or not out.is_contiguous():
        raise NativeError('filter_mask: out must be contiguous (%d, >= %d)' % (B, words))
    if ptr.numel() != keys.numel() + 1:
        raise NativeError('filter_mask: ptr must have len(keys) + 1 entries')
    _same_device(qkey, keys, ptr, tails, out)
    _check(lib().mgcn_filter_mask(B, _dev(qkey, torch.int64, 'qkey'), keys.numel(), _dev(keys, torch.int64, 'keys'),
                                  _dev(ptr, torch.int64, 'ptr'), _dev(tails, torch.int32, 'tails'), int(ent_row0),
                                  int(n_local), _dev(out, torch.int32, 'mask'), out.size(1), _stream(qkey)),
           'mgcn_filter_mask')
    return out


def smoothed_targets(lbl_smooth, num_entities):
    """(hot, cold) = (1 - eps) * y + 1/N for y = 1, 0, evaluated in f32 as numpy does (data_loader.py:41-43)."""
    import numpy as np
    y = np.array([1.0, 0.0], dtype=np.float32)
    if lbl_smooth != 0.0:
        y = (1.0 - lbl_smooth) * y + (1.0 / int(num_entities))
    return float(y[0]), float(y[1])


def score_bce_supported(x, ent):
    return (x.size(0) % 4 == 0 and x.size(1) % 4 == 0 and x.is_contiguous() and ent.is_contiguous()
            and x.data_ptr() % 16 == 0 and ent.data_ptr() % 16 == 0)


def score_bce_fwd(x, ent, bias, mask, hot, cold):
    """(N3) One launch: returns (loss [] f32, G [n, B] f32 = d loss / d logits, entity-major) for the mean BCE of
    sigmoid(x @ ent^T + bias) against targets `hot` at the mask's bits, `cold` elsewhere (see mgcn_score_bce_fwd)."""
    B, n, O = _score_args(x, ent, bias)
    if mask.dim() != 2 or mask.size(0) != B or mask.size(1) < (n + 31) // 32 or not mask.is_contiguous():
        raise NativeError('score_bce_fwd: mask must be contiguous (%d, >= %d)' % (B, (n + 31) // 32))
    _same_device(x, ent, bias, mask)
    g = torch.empty((n, B), dtype=torch.float32, device=x.device)
    parts = torch.empty(int(lib().mgcn_score_bce_partials(B, n)), dtype=torch.float32, device=x.device)
    inv = 1.0 / (float(B) * float(n))
    _check(lib().mgcn_score_bce_fwd(B, n, O, _dev(x, torch.float32, 'x'), _ld(x), _dev(ent, torch.float32, 'ent'), _ld(ent),
                                    _dev(bias, torch.float32, 'bias'), _dev(mask, torch.int32, 'mask'), mask.size(1),
                                    float(hot), float(cold), inv, _dev(g, torch.float32, 'grad_logit'), g.stride(0),
                                    _dev(parts, torch.float32, 'loss_partial'), _stream(x)), 'mgcn_score_bce_fwd')
    return parts.sum() * inv, g


def label_rows(qkey, keys, ptr, tails, n_local, lbl_smooth=0.0, num_entities=None, ent_row0=0, out=None):
    """Dense training targets [B, n_local] f32 for queries with keys `qkey` (see mgcn_label_rows): 1 at the known
    tails, 0 elsewhere, then (1 - eps) * y + 1/N when eps != 0 (data_loader.py:41-43, evaluated in f32 as numpy does)."""
    B = qkey.numel()
    y = smoothed_targets(lbl_smooth, int(n_local if num_entities is None else num_entities))
    if out is None:
        out = torch.empty((B, int(n_local)), dtype=torch.float32, device=qkey.device)
    if out.size(0) != B or out.size(1) < n_local or out.stride(1) != 1:
        raise NativeError('label_rows: out must be (%d, >= %d) with unit column stride' % (B, n_local))
    if ptr.numel() != keys.numel() + 1:
        raise NativeError('label_rows: ptr must have len(keys) + 1 entries')
    _same_device(qkey, keys, ptr, tails, out)
    _check(lib().mgcn_label_rows(B, _dev(qkey, torch.int64, 'qkey'), keys.numel(), _dev(keys, torch.int64, 'keys'),
                                 _dev(ptr, torch.int64, 'ptr'), _dev(tails, torch.int32, 'tails'), int(ent_row0),
                                 int(n_local), float(y[0]), float(y[1]), _dev(out, torch.float32, 'labels'),
                                 out.stride(0), _stream(qkey)), 'mgcn_label_rows')
    return out


class IngestUnsupported(NativeError):
    """The native reader declined the files (non-ASCII names): use the Python reader."""


def ingest(train_path, valid_path, test_path):
    """(0) Native reader of the three split files. Returns (entity_names, relation_names, {split: [n, 3] int64 ids}):
    names in id order (first-seen over train, valid, test; lower-cased), ids as data_loader.py:84-86 assigns them.
    Raises ValueError / KeyError where the reference's reader does, IngestUnsupported for non-ASCII names."""
    handle = _ptr()
    rc = lib().mgcn_ingest_open(os.fsencode(train_path), os.fsencode(valid_path), os.fsencode(test_path),
                                ctypes.byref(handle))
    if rc != 0:
        msg = (lib().mgcn_last_error() or b'').decode('utf-8', 'replace')
        if rc == 3:
            raise IngestUnsupported(msg)
        if 'KeyError' in msg:
            raise KeyError(msg.split("'")[1] if "'" in msg else msg)
        if 'ValueError' in msg:
            raise ValueError(msg)
        if 'cannot open' in msg:
            raise FileNotFoundError(msg)
        raise NativeError('mgcn_ingest_open failed (%d): %s' % (rc, msg))
    try:
        names = []
        for kind in (0, 1):
            n, nbytes = lib().mgcn_ingest_count(handle, kind), lib().mgcn_ingest_names_bytes(handle, kind)
            buf = ctypes.create_string_buffer(max(int(nbytes), 1))
            offs = torch.empty(n + 1, dtype=torch.int64)
            _check(lib().mgcn_ingest_names(handle, kind, ctypes.cast(buf, _ptr), offs.data_ptr()), 'mgcn_ingest_names')
            raw, o = buf.raw, offs.tolist()
            names.append([raw[o[i]:o[i + 1]].decode('ascii') for i in range(n)])
        ids = {}
        for k, split in enumerate(('train', 'valid', 'test')):
            t = torch.empty((int(lib().mgcn_ingest_count(handle, 2 + k)), 3), dtype=torch.int64)
            _check(lib().mgcn_ingest_triples(handle, k, t.data_ptr() if t.numel() else None), 'mgcn_ingest_triples')
            ids[split] = t
        return names[0], names[1], ids
    finally:
        lib().mgcn_ingest_close(handle)


def filter_index_build(triples, num_relations):
    """Known-answer index of [n, 3] int64 id triples, both directions: (keys [K] int64, ptr [K+1] int64, tails int32),
    all host tensors (see mgcn_filter_index_build)."""
    t = triples.detach().to('cpu', torch.int64).contiguous().reshape(-1, 3)
    nk, nt = _i64(0), _i64(0)
    tp = t.data_ptr() if t.numel() else None
    _check(lib().mgcn_filter_index_build(t.size(0), tp, int(num_relations), None, None, None, ctypes.byref(nk),
                                         ctypes.byref(nt)), 'mgcn_filter_index_build')
    keys = torch.empty(nk.value, dtype=torch.int64)
    ptr = torch.empty(nk.value + 1, dtype=torch.int64)
    tails = torch.empty(nt.value, dtype=torch.int32)
    _check(lib().mgcn_filter_index_build(t.size(0), tp, int(num_relations), keys.data_ptr(), ptr.data_ptr(),
                                         tails.data_ptr() if nt.value else None, ctypes.byref(nk), ctypes.byref(nt)),
           'mgcn_filter_index_build')
    return keys, ptr, tails
