"""Host-side mirror of the reference's Python surface (python/bsqp/interface.py, python/bindings.cu) -- no GPU needed."""
import importlib
import inspect
import sys

import pytest

from gato_amd.bsqp.config import STANDARD_BATCH_SIZES, SUPPORTED_KNOT_POINTS, SUPPORTED_PLANTS


@pytest.mark.parametrize("plant", SUPPORTED_PLANTS)
@pytest.mark.parametrize("N", SUPPORTED_KNOT_POINTS)
def test_modules_and_classes_exist(plant, N):
    m = importlib.import_module("gato_amd.bsqp.bsqpN%d_%s" % (N, plant))  # CMakeLists.txt:46-61
    assert m.KNOT_POINTS == N                                            # bindings.cu:241
    for b in STANDARD_BATCH_SIZES:                                       # bindings.cu:254-264
        cls = getattr(m, "BSQP_%d_float" % b)
        assert cls.BATCH_SIZE == b and cls.KNOT_POINTS == N and cls.PLANT == plant
        for meth in ("solve", "reset_dual", "set_f_ext_batch", "set_rho_penalty_batch", "set_drho_batch", "set_mu_batch", "set_pcg_tol_batch",
                     "sim_forward", "reset_rho", "set_rho_adaptation"):  # bindings.cu:224-237
            assert callable(getattr(cls, meth))
    assert "set_as_reset_default: bool = True" in m.BSQP_1_float.set_rho_penalty_batch.__doc__        # py::arg(...) = true, bindings.cu:229-230
    from gato_amd import _gato_ext                                      # the classes are the COMPILED pybind11 class with (plant, N, B) fixed
    assert issubclass(m.BSQP_1_float, _gato_ext.BSQP) and type(_gato_ext.BSQP).__name__ == "pybind11_type"
    assert not hasattr(m, "BSQP_0_float")
    dbl = m.BSQP_8_double                                               # a USE_DOUBLES build's classes (bindings.cu:244-252), batch <= 128
    assert dbl.BATCH_SIZE == 8 and dbl.__name__ == "BSQP_8_double" and not issubclass(dbl, _gato_ext.BSQP)
    assert not hasattr(m, "BSQP_256_double")


def test_facade_signature_and_errors():
    from gato_amd.bsqp.interface import BSQP
    sig = inspect.signature(BSQP.__init__)
    names = list(sig.parameters)[1:]
    assert names == ["model_path", "batch_size", "N", "dt", "max_sqp_iters", "kkt_tol", "max_pcg_iters", "pcg_tol", "solve_ratio", "mu", "q_cost",
                     "qd_cost", "u_cost", "N_cost", "q_lim_cost", "vel_lim_cost", "ctrl_lim_cost", "rho", "rho_batch", "mu_batch", "pcg_tol_batch",
                     "adapt_rho", "plant_type"]                           # interface.py:7-31
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["max_sqp_iters"], d["mu"], d["q_cost"], d["rho"], d["plant_type"]) == (10, 1.0, 2.0, 0.0, "indy7")
    with pytest.raises(ValueError, match="Number of knots 33 not supported"):   # interface.py:45-50
        BSQP("indy7.urdf", 4, 33, 0.01)
    for meth in ("solve", "ee_pos", "reset", "sim_forward", "set_f_ext_B", "reset_rho", "reset_dual", "get_stats"):
        assert callable(getattr(BSQP, meth))


def test_constructor_arity():
    import gato_amd.bsqp.bsqpN8_indy7 as m
    with pytest.raises(TypeError):
        m.BSQP_1_float(0.01, 5)


def test_install_as_bsqp():
    import gato_amd.bsqp as pkg
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "bsqp" or k.startswith("bsqp.")}
    try:
        pkg.install_as_bsqp()
        from bsqp.interface import BSQP  # the reference's import line (examples/benchmark_fig8.py)
        import bsqp.common
        assert BSQP.__module__ == "gato_amd.bsqp.interface" and hasattr(bsqp.common, "figure8")
    finally:
        for k in [k for k in sys.modules if k == "bsqp" or k.startswith("bsqp.")]:
            del sys.modules[k]
        sys.modules.update(saved)
