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
