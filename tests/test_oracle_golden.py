"""The oracle (oracle/mgcn_oracle.py) against the golden vectors produced by running the reference
(tests/golden/gen/make_golden.py). CPU only. This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from .conftest import ALL_CASES, ENCODER_CASES, FULL_CASES, GOLDEN, golden


@pytest.mark.parametrize('case', ALL_CASES)
def test_ids_edges_queries_bit_exact(oracle, case):
    g = golden(case)
    ds = oracle.load_dataset(g.data_dir)
    assert ds['num_entity'] == int(g['dl_num_entity'])
    assert ds['num_relation'] == int(g['dl_num_relation'])
    assert ds['num_edge'] == int(g['dl_num_edge'])
    assert sorted(ds['entity2id'], key=ds['entity2id'].get) == list(g['dl_entity_names'])
    assert sorted(ds['relation2id'], key=ds['relation2id'].get) == list(g['dl_relation_names'])
    ei, ea = oracle.build_edge_list(ds['data']['train'], ds['num_relation'])
    assert np.array_equal(ei, g['dl_edge_index'])
    assert np.array_equal(ea, g['dl_edge_attr'])
    assert np.array_equal(oracle.edge_normal(ei, ds['num_entity']).numpy(), g['dl_edge_norm'])
    for split in ('train', 'valid_tail', 'valid_head', 'test_tail', 'test_head'):
        qs = ds['queries'][split]
        assert np.array_equal(np.array([q[0] for q in qs], dtype=np.int64).reshape(-1, 3), g['dl_q_%s_triple' % split])
        ptr = np.cumsum([0] + [len(q[1]) for q in qs])
        assert np.array_equal(ptr, g['dl_q_%s_label_ptr' % split])
        assert np.array_equal(np.array([e for q in qs for e in q[1]], dtype=np.int64), g['dl_q_%s_label_idx' % split])


def test_uppercase_token_raises_like_reference(oracle, tmp_path):
    # Q7: ids are built from lower-cased tokens but looked up raw (data_loader.py:67 vs 85-86)
    for split in ('train', 'valid', 'test'):
        (tmp_path / (split + '.txt')).write_text('A\tr\tb')
    with pytest.raises(KeyError):
        oracle.load_dataset(str(tmp_path))


@pytest.mark.parametrize('case', ALL_CASES)
def test_norms_bit_exact(oracle, case):
    g = golden(case)
    ei = g.t('dl_edge_index')
    E, N = int(g['dl_num_edge']), int(g['dl_num_entity'])
    assert np.array_equal(oracle.compute_norm(ei[:, :E], N).numpy(), g['norm_in'])
    assert np.array_equal(oracle.compute_norm(ei[:, E:], N).numpy(), g['norm_out'])


@pytest.mark.parametrize('case', ALL_CASES)
def test_layer_eval_bit_exact(oracle, case):
    g = golden(case)
    sd = g.state_dict()
    ei, ea = g.t('dl_edge_index'), g.t('dl_edge_attr')
    ee = sd['edge_embeddings'].index_select(0, ea[1])
    all_ent, all_rel = oracle.layer_forward(sd, 'conv1.', sd['entity_embedding'], ei, ea[0], ee, sd['relation_embedding'])
    assert np.array_equal(all_ent.numpy(), g['eval_all_ent'])
    assert np.array_equal(all_rel.numpy(), g['eval_all_rel'])


@pytest.mark.parametrize('case', ALL_CASES)
def test_build_order_within_rounding_of_reference_order(oracle, case):
    # W after the sum differs from W per edge only by f32 rounding
    g = golden(case)
    sd = g.state_dict()
    ei, ea = g.t('dl_edge_index'), g.t('dl_edge_attr')
    ee = sd['edge_embeddings'].index_select(0, ea[1])
    out, _ = oracle.aggregate_then_weight(sd, 'conv1.', sd['entity_embedding'], ei, ea[0], ee, sd['relation_embedding'])
    out = torch.nn.functional.batch_norm(out, sd['conv1.ent_bn.running_mean'], sd['conv1.ent_bn.running_var'],
                                         sd['conv1.ent_bn.weight'], sd['conv1.ent_bn.bias'], False, 0.1, 1e-5)
    np.testing.assert_allclose(torch.tanh(out).numpy(), g['eval_all_ent'], rtol=0, atol=2e-5)


@pytest.mark.parametrize('case', FULL_CASES)
def test_full_forward_and_ranks(oracle, case):
    g = golden(case)
    sd = g.state_dict()
    ei, ea = g.t('dl_edge_index'), g.t('dl_edge_attr')
    ds = oracle.load_dataset(g.data_dir)
    N = ds['num_entity']
    for split in ('valid_tail', 'valid_head', 'test_tail', 'test_head'):
        qs = ds['queries'][split]
        trip = torch.tensor([q[0] for q in qs], dtype=torch.long)
        label = torch.stack([oracle.label_row(q[1], N) for q in qs])
        pred = oracle.mgcn_forward(sd, g.hp, trip[:, 0], trip[:, 1], ei, ea)
        assert np.array_equal(pred.numpy(), g['eval_%s_score' % split])
        r = oracle.filtered_rank(pred, label, trip[:, 2])
        for k in ('ranks', 'gt', 'ties', 'ties_lower', 'target'):
            assert np.array_equal(r[k].numpy(), g['eval_%s_%s' % (split, k)]), k
        # tie-free rows: reference rank == 1 + gt (the definition the HIP path is held to)
        free = r['ties'] == 0
        assert torch.equal(r['ranks'][free], 1 + r['gt'][free])
        assert bool(((r['ranks'] >= 1 + r['gt']) & (r['ranks'] <= 1 + r['gt'] + r['ties'])).all())


@pytest.mark.parametrize('case', FULL_CASES)
def test_evaluate_matches_reference_evaluate(oracle, case):
    g = golden(case)
    sd = g.state_dict()
    ds = oracle.load_dataset(g.data_dir)
    for split in ('valid', 'test'):
        res, parts = oracle.evaluate(sd, g.hp, ds, g.t('dl_edge_index'), g.t('dl_edge_attr'), split,
                                     batch_size=g.hp['batch_size'])
        for k in ('mr', 'mrr', 'hits@1', 'hits@3', 'hits@10'):
            assert abs(float(res[k]) - float(g['evaluate_%s_%s' % (split, k)])) <= 2e-5, (split, k)
        assert parts['tail']['count'] == float(g['predict_%s_tail_count' % split])
        assert parts['tail']['mr'] == float(g['predict_%s_tail_mr' % split])
        for k in range(1, 11):
            assert parts['tail']['hits@%d' % k] == float(g['predict_%s_tail_hits@%d' % (split, k)])


def test_toy_untrained_published_probe(oracle):
    # untrained Toy, seed 2020 -> mr 3.66667, mrr 0.42996, hits 0.25 / 0.5 / 1.0
    with open(os.path.join(GOLDEN, 'toy_untrained_eval.json')) as f:
        res = json.load(f)
    assert res == {'mr': 3.66667, 'mrr': 0.42996, 'hits@1': 0.25, 'hits@3': 0.5, 'hits@10': 1.0}


@pytest.mark.parametrize('case', FULL_CASES)
def test_label_smoothing(oracle, case):
    g = golden(case)
    ds = oracle.load_dataset(g.data_dir)
    q0 = ds['queries']['train'][0]
    row = oracle.label_row(q0[1], ds['num_entity'], training=True, lbl_smooth=0.1)
    assert np.array_equal(row.numpy(), g['smooth_label0'])


@pytest.mark.parametrize('case', FULL_CASES)
def test_train_step_gradients(oracle, case):
    """One BCE step, dropout 0, lbl_smooth 0 (main.py:59-66): autograd through the oracle's forward
    reproduces the reference's gradients and BN running statistics."""
    g = golden(case)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in g.state_dict().items()}
    hp = dict(g.hp, gcn_drop=0.0, hidden_drop=0.0, feat_drop=0.0)
    trip, lab = g.t('train_triple'), g.t('train_label')
    pred = oracle.mgcn_forward(sd, hp, trip[:, 0], trip[:, 1], g.t('dl_edge_index'), g.t('dl_edge_attr'), training=True)
    assert np.array_equal(pred.detach().numpy(), g['train_score'])
    loss = torch.nn.functional.binary_cross_entropy(pred, lab)
    assert float(loss) == float(g['train_loss'])
    loss.backward()
    for k, ref in g.grads().items():
        got = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-8, err_msg=k)
    for k in g.z.files:
        if k.startswith('train_after_') and 'num_batches' not in k:
            np.testing.assert_allclose(sd[k[len('train_after_'):]].detach().numpy(), g[k], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('case', ENCODER_CASES)
def test_encoder_train_mode_gradients(oracle, case):
    g = golden(case)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in g.state_dict().items()}
    ei, ea = g.t('dl_edge_index'), g.t('dl_edge_attr')
    ee = sd['edge_embeddings'].index_select(0, ea[1])
    all_ent, all_rel = oracle.layer_forward(sd, 'conv1.', sd['entity_embedding'], ei, ea[0], ee, sd['relation_embedding'], training=True)
    assert np.array_equal(all_ent.detach().numpy(), g['train_all_ent'])
    ((all_ent * g.t('train_G')).sum() + (all_rel * g.t('train_H')).sum()).backward()
    for k, ref in g.grads().items():
        got = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
