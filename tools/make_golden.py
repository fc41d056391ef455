#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Runs ONLY in the authoring container.

1. reference_python.npz -- outputs of the importable Python half of the reference (/root/reference/python/bsqp), imported here with
   `pinocchio` stubbed (it is only needed by callers of the path): `figure8(0.01, **FIG8_DEFAULT_PARAMS)`, `initialize_warm_start`,
   and the default parameter dictionaries.  tests/test_workloads.py checks gato_amd.bsqp.{common,config} against it.
2. oracle_*.npz -- small end-to-end inputs/outputs of the CPU oracle (inputs from gato_amd.bsqp.workloads, seed 0): regression
   pins for the oracle itself and fixtures for the `-m gpu` parity tests.  (The CUDA reference cannot run here: DESIGN.md (c).)
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def reference_python():
    ref = "/root/reference/python"
    if not os.path.isdir(ref):
        print("reference not present; skipping reference_python.npz")
        return
    sys.modules.setdefault("pinocchio", types.ModuleType("pinocchio"))
    sys.path.insert(0, ref)
    import importlib
    common = importlib.import_module("bsqp.common")
    config = importlib.import_module("bsqp.config")
    fig8 = common.figure8(0.01, **config.FIG8_DEFAULT_PARAMS)
    fig8_b = common.figure8(0.02, A_x=0.3, A_z=0.2, offset=[0.1, 0.4, 0.5], period=4, cycles=2, theta=0.3)
    ws = common.initialize_warm_start(np.arange(12, dtype=float), 5, 12, 6)
    np.savez_compressed(
        os.path.join(GOLD, "reference_python.npz"), fig8=fig8, fig8_b=fig8_b, warm_start=ws,
        indy7_ready=config.INDY7_START_CONFIGS["ready"],
        default_solver_params=json.dumps(config.DEFAULT_SOLVER_PARAMS), pickplace_solver_params=json.dumps(config.PICKPLACE_SOLVER_PARAMS),
        fig8_default_params=json.dumps({k: (list(v) if isinstance(v, (list, tuple)) else float(v)) for k, v in config.FIG8_DEFAULT_PARAMS.items()}),
        standard_batch_sizes=np.array(config.STANDARD_BATCH_SIZES))
    sys.path.remove(ref)
    for m in [m for m in sys.modules if m == "bsqp" or m.startswith("bsqp.")]:
        del sys.modules[m]
    print("wrote reference_python.npz")


def reference_force_estimator():
    """ForceEstimator of the reference (examples/force_estimator.py, pure numpy) driven through a fixed script: batches at reset, after
    every update and after a reset, with numpy's global stream seeded -> tests/golden/reference_force_estimator.npz"""
    ref = "/root/reference/examples"
    if not os.path.isdir(ref):
        print("reference not present; skipping reference_force_estimator.npz")
        return
    sys.path.insert(0, ref)
    import importlib
    fe = importlib.import_module("force_estimator")
    sys.path.remove(ref)
    out = {}
    for B in (4, 16, 128):
        np.random.seed(1234 + B)
        est = fe.ForceEstimator(batch_size=B, initial_radius=5.0, min_radius=2.0, max_radius=20.0, smoothing_factor=0.5)  # mpc_controller.py:126-132
        rng = np.random.default_rng(B)
        batches, radii, script = [est.generate_batch()], [est.radius], []
        for step in range(12):
            errors = rng.uniform(0.0, 1.0, B) * (0.02 if step in (6, 7, 8, 9, 10) else 1.0)
            best = int(np.argmin(errors)) if step % 3 else int(rng.integers(0, B))
            est.update(best, errors, alpha=0.6, beta=0.5)                                                                  # mpc_controller.py:307
            script.append(np.concatenate([[best], errors]))
            batches.append(est.generate_batch())
            radii.append(est.radius)
        st = est.get_stats()
        est.reset()
        out["B%d_batches" % B] = np.stack(batches)
        out["B%d_radius" % B] = np.array(radii, np.float64)
        out["B%d_script" % B] = np.stack(script)
        out["B%d_confidence" % B] = np.float64(st["confidence"])
        out["B%d_after_reset" % B] = est.generate_batch()
        out["B%d_sphere" % B] = est.sphere_dirs
    np.savez_compressed(os.path.join(GOLD, "reference_force_estimator.npz"), **out)
    del sys.modules["force_estimator"]
    print("wrote reference_force_estimator.npz")


def oracle_cases():
    from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
    from gato_amd.bsqp.workloads import fig8_problem
    from oracle.oracle import OracleSolver
    cases = [("indy7", 8, 1, 0.0, 3), ("indy7", 32, 4, 5.0, 3), ("iiwa14", 8, 2, 0.0, 3), ("iiwa14", 32, 2, 5.0, 3)]
    for plant, N, B, fstd, iters in cases:
        p = dict(DEFAULT_SOLVER_PARAMS)
        p["max_sqp_iters"] = iters
        pr = fig8_problem(plant, N, B, seed=0, f_ext_std=fstd)
        s = OracleSolver(plant, N, B, dt=0.01, **p)
        s.set_f_ext_batch(pr["f_ext"])
        # stage dump of the first iteration
        s.setup_kkt(pr["xu"], pr["x_s"], pr["ref"], 0.01)
        s.form_schur()
        st = {k: s.buf(k) for k in ("Q", "R", "q", "r", "A", "B", "c", "Qinv", "Rinv", "S", "Pinv", "gamma")}
        m1 = s.merit(pr["xu"], pr["x_s"], pr["ref"], 0.01, num_alphas=1, zero_dz=True)
        out = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
        np.savez_compressed(
            os.path.join(GOLD, "oracle_%s_N%d_B%d.npz" % (plant, N, B)), params=json.dumps(p), dt=0.01, **{"in_" + k: v for k, v in pr.items()},
            merit0=m1, **{"st_" + k: v for k, v in st.items()},
            **{"out_" + k: np.asarray(v) for k, v in out.items()})
        print("wrote oracle_%s_N%d_B%d.npz" % (plant, N, B))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    reference_python()
    reference_force_estimator()
    if "--no-oracle" not in sys.argv:
        oracle_cases()
