"""A fixed-seed slice of the two differential fuzzers (tools/fuzz_trees.py, tools/fuzz_robots.py) in the GPU suite:
random tree shapes and the six robot models under random world parameters, one step from shared FP32 states against
the oracle, through the C-ABI.  The tools hold the bars (IntParity / FloatParity, see test_gpu_parity.py) and say
which sample classes they do not compare and why; their long runs are recorded under profiles/."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_tree_shapes_slice():
    import fuzz_trees
    r = fuzz_trees.run(n_cases=40, seed=11)
    assert r["cases"] >= 30 and r["contact_samples"] > 1000
    assert r["outliers"] <= 2


def test_robots_under_random_world_parameters_slice():
    import fuzz_robots
    r = fuzz_robots.run(n_cases=40, seed=11)
    assert r["cases"] == 40 and r["contact_samples"] > 2000
    assert r["outliers"] <= 2 and r["ill_conditioned"] <= 3
