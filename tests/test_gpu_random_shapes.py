"""Randomised parity of the three kernel families on shapes nobody picked by hand (`-m gpu`): short, seeded instances
of the stress tools under tools/ (which run longer sweeps by hand). Each trial draws a graph / shape / hub setting /
destination range from a seeded generator, so a failure names its trial and reproduces.

* fused layer launch (model.py:82-118 in one launch) against the two-launch path (exact-f32 MFMA) to 5e-5, the relation
  projection and a random destination range + table shard bit for bit, and the two-launch path against the oracle in
  the reference's operation order to 1e-4 (tools/stress_fused.py);
* score / target / filtered rank counts (model.py:177-179, main.py:122-126) with bit-mask and dense-label filters and an
  entity shard split, exact against a torch recount over the same scores (tools/stress_rank.py);
* aggregation forward + backward (autograd through model.py:99-101, 111-118) against a float64 restatement to 2e-5
  relative (tools/stress_backward.py)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _tool(name):
    spec = importlib.util.spec_from_file_location('tools_' + name, os.path.join(ROOT, 'tools', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('seed', [11, 12])
def test_fused_layer_random_shapes(seed):
    ok, worst = _tool('stress_fused').run(seed=seed, trials=12, keep_going=False)
    assert ok, 'fused layer differs from the two-launch path / the oracle (see the trial printed last)'
    assert worst < 1e-4


@pytest.mark.parametrize('seed', [13, 14])
def test_fused_layer_wide_shapes(seed):
    """input widths beyond one 256-column pass (D in 200..1024: one, two, four passes) and outputs beyond
    13 column tiles (O up to 512: the 32-column-tile instance), same checks as above."""
    ok, worst = _tool('stress_fused').run(seed=seed, trials=10, keep_going=False, wide=True)
    assert ok, 'fused layer (wide shapes) differs from the two-launch path / the oracle (see the trial printed last)'
    assert worst < 1e-4


def test_score_and_rank_random_shapes():
    assert _tool('stress_rank').run(seed=21, trials=16)


def test_aggregation_backward_random_shapes():
    ok, worst = _tool('stress_backward').run(seed=31, trials=14)
    assert ok, 'aggregation forward / backward differs from the float64 restatement by %.2e (relative)' % worst
