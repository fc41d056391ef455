"""tests/stage_bounds.py on the CPU: the forward-error bounds that the full-size stage-wise GPU test holds gamma and dz to are (1) the oracle's
own formulas -- the float64 build of the oracle reproduces `exact` to rounding, (2) satisfied by the fp32 oracle with a wide margin, on fig-8 data
and on the sweep (C5) where dz_u = -R^-1 (...) cancels seven digits, and (3) SHARP enough to be a test: a mis-indexed, mis-signed or
mis-laid-out evaluation exceeds them by orders of magnitude (schur_linsys.cuh:81,121-128,316-431)."""
import numpy as np
import pytest

import stage_bounds as SB
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem, hparam_problem
from oracle.oracle import OracleSolver

KEYS = ("A", "B", "Qinv", "Rinv", "q", "r")


def _stages(kind, f64, lam=None):
    if kind == "sweep":
        plant, N, B = "iiwa14", 16, 24
        pr = hparam_problem(plant, N, B, shard=3)
        pr["rho"] = 10.0 ** np.linspace(-8, 1, B).astype(np.float32)     # the whole range of the sweep in 24 rows
        p, dt = pr["params"], pr["dt"]
    else:
        plant, N, B = "indy7", 16, 8
        pr, p, dt = fig8_problem(plant, N, B, f_ext_std=3.0), dict(DEFAULT_SOLVER_PARAMS), 0.01
    o = OracleSolver(plant, N, B, dt=dt, f64=f64, **p)
    o.set_f_ext_batch(pr["f_ext"])
    if "rho" in pr:
        o.set_rho_penalty_batch(pr["rho"])
    o.setup_kkt(pr["xu"], pr["x_s"], pr["ref"], dt)
    t = {k: o.buf(k) for k in ("q", "r", "c")}
    o.form_schur()
    t.update({k: o.buf(k) for k in ("A", "B", "Qinv", "Rinv")})
    t["A"][:, N - 1] = 0
    t["B"][:, N - 1] = 0
    t["gamma"] = o.buf("gamma")
    if lam is None:
        o.pcg()
    else:
        o.set_lambda(lam)
    t["lambda"] = o.buf("lambda")
    o.compute_dz()
    t["dz"] = o.buf("dz")
    return t, o.nx


@pytest.mark.parametrize("kind", ["fig8", "sweep"])
def test_bounds_hold_for_the_oracle_and_catch_wrong_evaluations(kind):
    t, nx = _stages(kind, False)
    Kg, Kd = 2 * nx + 6, 2 * nx + 4
    g, bg = SB.gamma_exact_and_bound(*(t[k] for k in KEYS), t["c"])
    d, bd = SB.dz_exact_and_bound(*(t[k] for k in KEYS), t["lambda"])
    assert SB.ratio(t["gamma"], g, bg, Kg) <= 0.5 and SB.ratio(t["dz"], d, bd, Kd) <= 0.5
    # the float64 build of the oracle IS the exact formula (same inputs rounded to fp32 would differ: it runs its own float64 pipeline, so only
    # lambda is shared): its own tensors through the formulas give its own gamma / dz to float64 rounding
    t64, _ = _stages(kind, True, lam=t["lambda"])
    g64, _ = SB.gamma_exact_and_bound(*(t64[k] for k in KEYS), t64["c"])
    d64, _ = SB.dz_exact_and_bound(*(t64[k] for k in KEYS), t64["lambda"])
    assert np.abs(g64 - t64["gamma"]).max() <= 1e-9 * max(1.0, np.abs(g64).max())
    assert np.abs(d64 - t64["dz"]).max() <= 1e-9 * max(1.0, np.abs(d64).max())
    # wrong evaluations: lambda read one knot off, a transposed A, a sign, the control rows shifted by one entry
    lam_shift = np.roll(t["lambda"], 1, axis=1)
    wrong = {
        "lambda one knot off": SB.dz_exact_and_bound(*(t[k] for k in KEYS), lam_shift)[0],
        "A transposed": SB.dz_exact_and_bound(np.swapaxes(t["A"], -1, -2), *(t[k] for k in KEYS[1:]), t["lambda"])[0],
        "sign of the control step": d * np.where(np.arange(d.shape[1]) % (3 * nx // 2) >= nx, -1.0, 1.0),
        "layout shifted by one": np.roll(d, 1, axis=1),
    }
    for name, w in wrong.items():
        assert SB.ratio(w.astype(np.float32), d, bd, Kd) > 50.0, name
    for name, gw in {"sign of c": SB.gamma_exact_and_bound(*(t[k] for k in KEYS), -t["c"])[0], "A transposed": SB.gamma_exact_and_bound(
            np.swapaxes(t["A"], -1, -2), *(t[k] for k in KEYS[1:]), t["c"])[0]}.items():
        assert SB.ratio(gw.astype(np.float32), g, bg, Kg) > 50.0, name
