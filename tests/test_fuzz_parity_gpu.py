"""A small randomised parity sweep in the GPU suite (tools/fuzz_parity.py holds the generator; `python tools/fuzz_parity.py <seed> <n>` runs
thousands): random map sizes around every kernel boundary (hand-scheduled 16 / 32 / 64, compiled LDS loop, hybrid large-map kernel), obstacle
densities, cost kinds, g_ratio, budgets, with and without selection log and placement -- histories, paths, step counts and logs against the
oracle's state-machine restatement (reference differentiable_astar.py:150-267), bit-exact."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_sizes_costs_budgets_and_placements_match_the_oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    stats, bad = fuzz_parity.run(seed=5, N=120, big_frac=0.15, verbose=False)
    assert not bad, bad[:5]
    assert stats.get("lds", 0) >= 60 and stats.get("hybrid", 0) >= 8, stats


def test_random_small_maps_gradients_match_the_oracle_reverse_mode():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    n, bad = fuzz_parity.run_backward(seed=3, N=30, verbose=False)
    assert n >= 25 and not bad, bad[:5]


def test_module_forward_equals_the_literal_batch_loop_also_in_the_coupled_class():
    """forward() against the oracle's literal restatement of the reference's batch loop on random batches with g_ratio below, at and above 0.5,
    costs up to 10 and costs below -1: every checking mode, a third of the cases UNDER AUTOGRAD (gradients against the oracle's literal reverse
    mode, 1e-5), a tenth on maps whose state does not fit LDS (to 150x200), training budgets -- batches in which a finished map is not at a
    fixed point of the batch loop go through the exact pipeline (marks + lock-step re-run of the marked maps) and must come out exact"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    n, reruns, bad = fuzz_parity.run_module(seed=9, N=70, verbose=False, large_frac=0.08, large_hw=((112, 122), (112, 126)))  # (the sweep proper: 150x200)
    assert n >= 60 and not bad, bad[:5]
    assert reruns >= 4 and fuzz_parity.run_module.ngrad >= 8, (reruns, fuzz_parity.run_module.ngrad)  # the sweep does visit the class, also under autograd
