"""gato_amd.bsqp.force_estimator.ForceEstimator against a fixture generated from the IMPORTED reference
(examples/force_estimator.py:4-155 driven by tools/make_golden.py:reference_force_estimator): every hypothesis batch, the radius
schedule and the reset, bit for bit.  No GPU."""
import os

import numpy as np
import pytest

from gato_amd.bsqp.force_estimator import ForceEstimator

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_force_estimator.npz")


@pytest.mark.parametrize("B", [4, 16, 128])
def test_force_estimator_reproduces_the_reference(B):
    g = np.load(GOLD)
    np.random.seed(1234 + B)
    est = ForceEstimator(batch_size=B, initial_radius=5.0, min_radius=2.0, max_radius=20.0, smoothing_factor=0.5)
    np.testing.assert_array_equal(est.sphere_dirs, g["B%d_sphere" % B])
    batches, radii = g["B%d_batches" % B], g["B%d_radius" % B]
    np.testing.assert_array_equal(est.generate_batch(), batches[0])
    assert np.all(batches[0][1] == 0)                                   # hypothesis 1 is always "no force"
    for step, row in enumerate(g["B%d_script" % B]):
        est.update(int(row[0]), row[1:], alpha=0.6, beta=0.5)
        b = est.generate_batch()
        assert b.dtype == np.float32 and b.shape == (B, 6)
        np.testing.assert_array_equal(b, batches[step + 1])
        assert est.radius == radii[step + 1]
    assert est.get_stats()["confidence"] == float(g["B%d_confidence" % B])
    est.reset()
    np.testing.assert_array_equal(est.generate_batch(), g["B%d_after_reset" % B])
    assert est.radius == 10.0                                           # reset() goes back to 10, not to initial_radius (force_estimator.py:137)


def test_batch_size_guard():
    with pytest.raises(AssertionError):
        ForceEstimator(3)
