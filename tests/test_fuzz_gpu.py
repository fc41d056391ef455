"""A fixed-seed slice of tools/fuzz_parity.py in the GPU suite: 30 random (plant, horizon, batch, dt, wrench, cost weights, rho, mu)
configurations, one SQP iteration at PCG's floor against the fp32 and float64 oracles + a default 3-iteration solve that must descend."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_configurations_against_the_oracles():
    import fuzz_parity
    bad, worst = fuzz_parity.run(30, 7, verbose=False)
    assert bad == 0, worst
    assert worst["merit"] < 1e-5
