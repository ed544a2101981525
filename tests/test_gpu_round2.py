"""Round-2 additions on the GPU: encoder-cache / hipGraph invalidation, the operator seam with an edge
list that is not mirror-symmetric (model.py:82-101), slot-order bookkeeping of the per-edge tables and of the optimizer
state, and the fall-back from the fused launch."""
import types

import numpy as np
import pytest
import torch

from .conftest import golden
from .test_gpu_parity import DEV, _model

pytestmark = pytest.mark.gpu


def _oracle_encode(oracle, sd, ei, ea):
    return oracle.layer_forward(sd, 'conv1.', sd['entity_embedding'], ei, ea[0], sd['edge_embeddings'], sd['relation_embedding'])


@pytest.mark.parametrize('case', ['syn_a', 'syn_b'])
def test_encoder_cache_and_replay_follow_every_parameter_change(pkg, oracle, case):
    """main.py:117-121 recomputes the encoder per batch; the build caches it in eval mode and replays a captured
    hipGraph. After an optimizer step, a load_state_dict, an in-place edit of a BN statistic or of a table, the next
    encode() must return the NEW values (cache stamp = parameter versions; the replay reads through stable pointers)."""
    g = golden(case)
    model, dl, params = _model(pkg, g)
    params.cache_encoder = True
    ei, ea = g.t('dl_edge_index'), g.t('dl_edge_attr')
    model.eval()

    def check(tag):
        with torch.no_grad():
            ent, rel = model.encode(dl.graph)
            again, _ = model.encode(dl.graph)
        assert again.data_ptr() == ent.data_ptr(), tag            # second call: served from the cache
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        want_ent, want_rel = _oracle_encode(oracle, sd, ei, ea)
        np.testing.assert_allclose(ent.cpu().numpy(), want_ent.numpy(), rtol=0, atol=5e-5, err_msg=tag)
        np.testing.assert_allclose(rel.cpu().numpy(), want_rel.numpy(), rtol=0, atol=2e-5, err_msg=tag)
        return ent.clone()

    base = check('initial')
    # (1) one optimizer step on the encoder's parameters
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    model.train()
    trip = g.t('dl_q_test_tail_triple').to(DEV)
    loss = model(trip[:, 0], trip[:, 1], dl.graph).mean()
    loss.backward()
    opt.step()
    model.eval()
    after_step = check('after optimizer step')
    assert float((after_step - base).abs().max()) > 1e-4          # the step really changed the output
    # (2) BN running statistics edited in place (what a train-mode forward does)
    with torch.no_grad():
        model.conv1.ent_bn.running_mean.add_(0.05)
        model.conv1.ent_bn.running_var.mul_(1.3)
    after_bn = check('after BN statistics change')
    assert float((after_bn - after_step).abs().max()) > 1e-4
    # (3) a table edited in place
    with torch.no_grad():
        model.edge_embeddings.mul_(1.1)
    after_table = check('after per-edge table change')
    assert float((after_table - after_bn).abs().max()) > 1e-5
    # (4) load_state_dict back to the golden parameters
    model.load_state_dict(g.state_dict(), strict=False)
    back = check('after load_state_dict')
    np.testing.assert_allclose(back.cpu().numpy(), g['eval_all_ent'], rtol=0, atol=3e-5)
    # (5) .train() then .eval(): no stale cache from the train-mode forward in between
    model.train()
    model(trip[:, 0], trip[:, 1], dl.graph)
    model.eval()
    check('after a train-mode forward')


def test_operator_seam_with_a_non_mirrored_edge_list(pkg, oracle):
    """MGCNConv.forward accepts any [2, 2E] list split in halves by position (model.py:88-90). For a list whose second
    half is NOT the first half reversed: the forward (eval) matches the oracle; under autograd w.r.t. x the seam raises
    instead of returning a wrong gradient (the HIP backward walks destination runs through the reverse-edge map);
    gradients that do not need that map (per-edge table, relations) still come out right when x needs none."""
    N, R, E, D, O = 211, 4, 900, 20, 40
    gen = torch.Generator().manual_seed(21)
    ei = torch.randint(0, N, (2, 2 * E), generator=gen)           # two independent halves: not mirror-symmetric
    et = torch.cat([torch.randint(0, R, (E,), generator=gen), torch.randint(R, 2 * R, (E,), generator=gen)])
    sd = oracle.init_layer_state('conv1.', D, O, gen)
    x, ee, rel = torch.randn(N, D, generator=gen), torch.randn(2 * E, D, generator=gen), torch.randn(2 * R, D, generator=gen)
    want_ent, want_rel = oracle.layer_forward(sd, 'conv1.', x, ei, et, ee, rel)
    conv = pkg.MGCNConv(D, O, 2 * R)
    conv.load_state_dict({k[6:]: v for k, v in sd.items()})
    conv.to(DEV).eval()
    with torch.no_grad():
        got_ent, got_rel = conv(x.to(DEV), ei.to(DEV), et.to(DEV), None, ee.to(DEV), rel.to(DEV))
    np.testing.assert_allclose(got_ent.cpu().numpy(), want_ent.numpy(), rtol=0, atol=5e-5)
    np.testing.assert_allclose(got_rel.cpu().numpy(), want_rel.numpy(), rtol=0, atol=1e-5)
    csr = pkg.graph.csr_for_tensors(N, 2 * R + 1, ei.to(DEV), et.to(DEV))
    assert not csr.mirrored and int(csr.mirror.max()) == -1
    conv.train()
    conv.drop.p = 0.0
    with pytest.raises(pkg._native.NativeError, match='reversed'):
        conv(x.to(DEV).requires_grad_(True), ei.to(DEV), et.to(DEV), None, ee.to(DEV), rel.to(DEV))
    # x without gradient: the training-mode forward and the table / relation gradients against oracle autograd
    eed, reld = ee.to(DEV).requires_grad_(True), rel.to(DEV).requires_grad_(True)
    out, _ = conv(x.to(DEV), ei.to(DEV), et.to(DEV), None, eed, reld)
    out.square().sum().backward()
    ee_c, rel_c = ee.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    sd_c = {k: v.clone() for k, v in sd.items()}
    ref, _ = oracle.layer_forward(sd_c, 'conv1.', x, ei, et, ee_c, rel_c, training=True, drop_p=0.0)
    ref.square().sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=5e-5)
    np.testing.assert_allclose(eed.grad.cpu().numpy(), ee_c.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(reld.grad.cpu().numpy(), rel_c.grad.numpy(), rtol=1e-4, atol=2e-4)


def test_mirrored_flag_of_the_loader_graph(pkg):
    g = golden('syn_a')
    model, dl, params = _model(pkg, g)
    assert dl.graph.csr(2 * dl.num_relation + 1).mirrored


def test_tables_return_to_reference_order_for_a_graph_with_permuted_edge_ids(pkg, oracle):
    """after the tables were laid out in one graph's slot order, encoding a graph whose edge ids are not the
    identity gathers rows by edge id: the tables must first go back to reference order."""
    g = golden('syn_a')
    model, dl, params = _model(pkg, g)
    model.eval()
    with torch.no_grad():
        first, _ = model.encode(dl.graph)
        first = first.clone()
    assert model._slot_csr is not None
    E2 = dl.graph.edge_attr.size(1)
    perm = torch.randperm(E2, generator=torch.Generator().manual_seed(5))
    # the same edges listed with shuffled ids: edge k now reads table row perm[k]
    g2 = pkg.Graph(edge_index=dl.graph.edge_index.clone(), edge_attr=torch.stack([dl.graph.edge_attr[0].cpu(), perm]).to(DEV))
    g2.entity, g2.num_nodes, g2.edge_norm = dl.graph.entity, dl.graph.num_nodes, None
    with torch.no_grad():
        second, _ = model.encode(g2)
    assert model._slot_csr is None                                 # reference order restored before the gather
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ei, ea = g.t('dl_edge_index'), g.t('dl_edge_attr')
    want, _ = oracle.layer_forward(sd, 'conv1.', sd['entity_embedding'], ei, ea[0], sd['edge_embeddings'].index_select(0, perm),
                                   sd['relation_embedding'])
    np.testing.assert_allclose(second.cpu().numpy(), want.numpy(), rtol=0, atol=5e-5)
    with torch.no_grad():
        third, _ = model.encode(dl.graph)                           # and back to the loader's graph
    np.testing.assert_allclose(third.cpu().numpy(), first.cpu().numpy(), rtol=0, atol=1e-6)


def test_partial_load_state_dict_keeps_untouched_tables_consistent(pkg):
    """load_state_dict(strict=False) without the per-edge table must not leave slot-ordered data behind
    that is later treated as reference-ordered."""
    g = golden('syn_b')
    model, dl, params = _model(pkg, g)
    model.eval()
    with torch.no_grad():
        before, _ = model.encode(dl.graph)
        before = before.clone()
    partial = {k: v for k, v in g.state_dict().items() if k != 'edge_embeddings'}
    res = model.load_state_dict(partial, strict=False)
    assert 'edge_embeddings' in res.missing_keys
    assert torch.equal(model.state_dict()['edge_embeddings'].cpu(), g.state_dict()['edge_embeddings'])
    with torch.no_grad():
        after, _ = model.encode(dl.graph)
    assert torch.equal(after, before)


def test_optimizer_state_travels_in_reference_order(pkg):
    """Adam's moments of the per-edge table follow the parameter's in-place slot order; the helpers convert
    them so that a checkpoint's optim_dict does not depend on the slot layout."""
    g = golden('syn_b')
    model, dl, params = _model(pkg, g)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    trip = g.t('dl_q_test_tail_triple').to(DEV)
    model.train()
    model(trip[:, 0], trip[:, 1], dl.graph).mean().backward()
    opt.step()
    assert model._slot_csr is not None
    csr = model._slot_csr
    raw = opt.state[model.edge_embeddings]['exp_avg'].clone()                     # slot order
    sd = model.optimizer_state_dict(opt)
    idx = [i for i, p in enumerate(opt.param_groups[0]['params']) if p is model.edge_embeddings][0]
    assert torch.equal(sd['state'][idx]['exp_avg'], raw.index_select(0, csr.inv_perm))   # reference order on the way out
    # a second model with ANOTHER slot layout (hubs split differently) loads the same optim_dict
    other, dl2, _ = _model(pkg, g)
    other.load_state_dict(model.state_dict())
    csr2 = pkg.GraphCSR(csr.num_nodes, csr.num_rel_rows, dl2.graph.edge_index, dl2.graph.edge_attr[0], DEV, hub_threshold=2, hub_chunk=2)
    other._use_slot_order(csr2)
    opt2 = torch.optim.Adam(other.parameters(), lr=1e-3)
    other.load_optimizer_state_dict(opt2, sd)
    got = opt2.state[other.edge_embeddings]['exp_avg']
    assert torch.equal(got.index_select(0, csr2.inv_perm), raw.index_select(0, csr.inv_perm))
    assert torch.equal(other.state_dict()['edge_embeddings'], model.state_dict()['edge_embeddings'])


def test_fused_launch_falls_back_when_an_operand_is_misaligned(pkg, oracle):
    """a layer input whose rows are not 16-byte aligned makes mgcn_layer_fwd_fused return MGCN_EUNSUPPORTED;
    MGCNConv.forward then takes the aggregation + dense launches instead of raising."""
    g = golden('syn_c')
    sd = g.state_dict()
    ei, ea = g.t('dl_edge_index').to(DEV), g.t('dl_edge_attr').to(DEV)
    D, O = sd['conv1.in_weight'].shape
    conv = pkg.MGCNConv(D, O, sd['relation_embedding'].size(0), bias='conv1.bias' in sd)
    conv.load_state_dict({k[6:]: v for k, v in sd.items() if k.startswith('conv1.')})
    conv.to(DEV).eval()
    csr = pkg.graph.csr_for_tensors(sd['entity_embedding'].size(0), sd['relation_embedding'].size(0) + 1, ei, ea[0])
    table = sd['edge_embeddings'].to(DEV).index_select(0, csr.perm)
    wide = torch.zeros((sd['entity_embedding'].size(0), D + 1), device=DEV)
    x_bad = wide[:, 1:]                                              # row stride D + 1 floats, base + 4 bytes
    x_bad.copy_(sd['entity_embedding'])
    with torch.no_grad():
        with pytest.raises(pkg._native.FusedUnsupported):
            _, wpack = conv.derived_weights()
            bn = conv.ent_bn
            pkg._native.layer_fwd_fused(csr, x_bad, sd['relation_embedding'].to(DEV), conv.loop_rel.reshape(-1), table, True,
                                        conv.loop_edge.reshape(-1), wpack, O, conv.bias, bn.running_mean, bn.running_var,
                                        bn.weight, bn.bias, bn.eps, torch.empty((wide.size(0), O), device=DEV))


@pytest.mark.parametrize('dims', [(64, 128), (64, 256)])          # fused layers / O = 256 > 208: aggregation + dense launches
def test_ranks_that_hold_only_their_table_shard(pkg, oracle, dims):
    """SURVEY §8e / BASELINE configs[4]: a rank allocates ONLY the rows of the per-edge tables that its destination range
    needs (params.edge_table_rows + dist.shard_model_tables; the rows come from a chunk-wise defined xavier table, so the
    [2E, D] table never exists on it), and the rows it computes are torch.equal to the single-GPU ones. Three 'ranks' are
    played in one process, layer by layer (what dist.encode_sharded does with an all-gather in between)."""
    D, O = dims
    N, R, E, W = 6000, 5, 150000, 3
    tri = oracle.synthetic_triples(N, R, E, seed=3, zipf=1.0)
    ei, ea = oracle.build_edge_list(tri, R)
    graph = pkg.Graph(edge_index=torch.from_numpy(ei), edge_attr=torch.from_numpy(ea))
    graph.entity, graph.num_nodes, graph.edge_norm = torch.arange(N), N, None
    graph.to(DEV)
    csr = graph.csr(2 * R + 1)
    base = dict(gcn_in_dim=D, gcn_out_dim=O, gcn_drop=0.3, hidden_drop=0.3, feat_drop=0.3, k_w=8, k_h=O // 8,
                num_filter=4, kernel_size=3, bias=False, lbl_smooth=0.1, gcn_layers=2)
    dims_l = [D, O]
    source = lambda li, ids: pkg.dist.xavier_rows(ids, 2 * E, dims_l[li], 100 + li, DEV, chunk=1 << 14)
    # the single-GPU model: the same tables, whole (reference order = xavier_rows of every edge id)
    torch.manual_seed(7)
    full = pkg.MGCN(N, R, E, types.SimpleNamespace(**base)).to(DEV).eval()
    with torch.no_grad():
        for li, t in enumerate([full.edge_embeddings] + list(full.edge_embeddings_extra)):
            t.copy_(source(li, torch.arange(2 * E)))
        want_ent, want_rel = full.encode(graph)
    table_bytes = sum(t.numel() * 4 for t in [full.edge_embeddings] + list(full.edge_embeddings_extra))
    shared = {k: v for k, v in full.state_dict().items() if not k.startswith('edge_embeddings')}
    b = csr.balanced_bounds(W)
    models, peaks = [], []
    for r in range(W):
        n0, n1 = b[r], b[r + 1]
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        before = torch.cuda.memory_allocated()
        torch.manual_seed(7)
        m = pkg.MGCN(N, R, E, types.SimpleNamespace(edge_table_rows=sum(csr.shard_slot_counts(n0, n1)), **base)).to(DEV).eval()
        m.load_state_dict(shared, strict=False)
        pkg.dist.shard_model_tables(m, csr, n0, n1, source)
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated() - before)
        held = sum(t.numel() * 4 for t in [m.edge_embeddings] + list(m.edge_embeddings_extra))
        assert abs(held / table_bytes - 1.0 / W) < 0.05                      # a third of the table bytes, not all of them
        models.append(m)
        with pytest.raises(pkg._native.NativeError, match='shard'):
            m.encode(graph)                                                   # a partial table cannot be encoded alone
    other = sum(p.numel() * 4 for k, p in full.state_dict().items() if not k.startswith('edge_embeddings'))
    # building a rank: its shard + one layer's rows in flight + one chunk of the generator — never the whole table
    assert max(peaks) < other + 0.75 * table_bytes, (peaks, other, table_bytes)
    x, rel = full.entity_embedding.detach(), full.relation_embedding.detach()
    with torch.no_grad():
        for li in range(2):
            rows = []
            for r, m in enumerate(models):
                n0, n1 = b[r], b[r + 1]
                layer = ([m.conv1] + list(m.conv1_extra))[li]
                table = ([m.edge_embeddings] + list(m.edge_embeddings_extra))[li]
                rows.append(pkg.dist.encode_layer_rows(layer, csr, x, rel, table.detach(), n0, n1, csr.shard_ee_sub(n0, n1)))
            x = torch.cat(rows, dim=0)
            rel = pkg._native.matmul(rel.contiguous(), ([full.conv1] + list(full.conv1_extra))[li].rels_weight)
    assert torch.equal(x, want_ent)
    assert torch.equal(rel, want_rel)


def test_training_layer_kernels_vs_torch_autograd(pkg):
    """model.py:103-106,116 under .train() on the HIP path (csrc/train_layer.hip + the MFMA products) against torch's own
    float64 autograd of the same expression: outputs, running statistics (momentum, unbiased variance), and every gradient."""
    torch.manual_seed(5)
    N, D, O = 5003, 100, 200
    agg = torch.randn(N, 2 * D, device=DEV) * 0.5
    a_loop = torch.randn(N, D, device=DEV) * 0.5
    ws = [torch.randn(D, O, device=DEV) * 0.1 for _ in range(3)]
    bias, gamma, beta = torch.randn(O, device=DEV) * 0.1, torch.rand(O, device=DEV) + 0.5, torch.randn(O, device=DEV) * 0.1
    rm, rv = torch.randn(O, device=DEV) * 0.05, torch.rand(O, device=DEV) + 0.5
    gy = torch.randn(N, O, device=DEV)
    leaves = [t.clone().requires_grad_(True) for t in [agg, a_loop] + ws + [bias, gamma, beta]]
    rm1, rv1 = rm.clone(), rv.clone()
    y = pkg.model._LayerTrainFn.apply(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], leaves[5], leaves[6], leaves[7], rm1, rv1,
                                      0.1, 1e-5, 0.0)
    y.backward(gy)
    ref = [t.double().clone().requires_grad_(True) for t in [agg, a_loop] + ws + [bias, gamma, beta]]
    rm2, rv2 = rm.double().clone(), rv.double().clone()
    out = (ref[0][:, :D] @ ref[2] + ref[0][:, D:] @ ref[3] + ref[1] @ ref[4]) / 3 + ref[5]
    yr = torch.tanh(torch.nn.functional.batch_norm(out, rm2, rv2, ref[6], ref[7], True, 0.1, 1e-5))
    yr.backward(gy.double())
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(rm1.cpu().numpy(), rm2.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rv1.cpu().numpy(), rv2.cpu().numpy(), rtol=1e-5, atol=1e-7)
    names = ['agg', 'a_loop', 'w_in', 'w_out', 'w_loop', 'bias', 'gamma', 'beta']
    for name, a, b in zip(names, leaves, ref):
        scale = float(b.grad.abs().max()) + 1e-30
        err = float((a.grad.double() - b.grad).abs().max())
        # d bias is analytically zero (BN removes the column mean): what is left is the f32 cancellation error of a sum
        # of N = 5003 terms of size ~1 (float64 autograd leaves 3e-13)
        tol = 2e-5 * scale if name != 'bias' else 2e-3
        assert err <= tol, (name, err, scale)


@pytest.mark.parametrize('shape', [(40943, 100, 200), (14541, 200, 200), (777, 36, 44)])
def test_weight_gradient_product_split_k(pkg, shape):
    """mgcn_matmul_tn_f32 (dW = aggregate^T g): exact-f32 MFMA, split over the N rows, against float64."""
    K, M, N = shape
    g = torch.Generator().manual_seed(K)
    a, b = torch.randn(K, M, generator=g).to(DEV), torch.randn(K, N, generator=g).to(DEV)
    got = pkg._native.matmul_tn(a, b)
    want = (a.double().t() @ b.double())
    assert float((got.double() - want).abs().max()) <= 1e-5 * (K ** 0.5) * 4
    wide = torch.randn(K, 2 * M, generator=g).to(DEV)                       # a column view with a row stride (the aggregate's halves)
    assert torch.equal(pkg._native.matmul_tn(wide[:, M:], b), pkg._native.matmul_tn(wide[:, M:].contiguous(), b))
    assert torch.equal(pkg._native.matmul_tn(a, b), got)                     # reproducible


def test_training_layer_with_dropout_is_a_scaled_bernoulli_mask(pkg):
    """p > 0: the keep-masks are Bernoulli(1 - p) scaled by 1 / (1 - p) (as F.dropout): with W_loop = 0 and one-hot
    aggregates the pre-BN output exposes them; the same torch seed reproduces the same masks."""
    N, D, O, p = 2048, 8, 16, 0.25
    agg = torch.zeros(N, 2 * D, device=DEV)
    agg[:, 0] = 3.0                                                          # u_in[:, c] = 3 * w_in[0, c], u_out = 0
    w_in = torch.ones(D, O, device=DEV)
    zeros = torch.zeros(D, O, device=DEV)
    args = (agg, torch.zeros(N, D, device=DEV), w_in, zeros, zeros, None, torch.ones(O, device=DEV), torch.zeros(O, device=DEV),
            torch.zeros(O, device=DEV), torch.ones(O, device=DEV), 0.1, 1e-5, p)
    torch.manual_seed(3)
    y1 = pkg.model._LayerTrainFn.apply(*args)
    torch.manual_seed(3)
    y2 = pkg.model._LayerTrainFn.apply(*args)
    assert torch.equal(y1, y2)
    # per column z takes two values (0 and 3 / (3 * 0.75)); after BN + tanh still two values, the dropped share is ~ p
    low = (y1 < y1.mean(0, keepdim=True)).float().mean().item()
    assert abs(low - p) < 0.03, low
