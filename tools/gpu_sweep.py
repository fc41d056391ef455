#!/usr/bin/env python3
"""Sweep of (plant, N, B): one-iteration agreement of the HIP solve with the oracle and sanity of a full solve (debug aid; needs a GPU)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver  # noqa: E402
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS  # noqa: E402
from gato_amd.bsqp.workloads import fig8_problem  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402

DT = 0.01
bad = 0
for plant, N, B, fstd in [("indy7", 4, 1, 0.0), ("indy7", 8, 3, 2.0), ("indy7", 16, 5, 0.0), ("indy7", 32, 7, 3.0), ("indy7", 64, 3, 0.0), ("indy7", 128, 2, 1.0),
                          ("indy7", 256, 1, 0.0), ("iiwa14", 4, 2, 0.0), ("iiwa14", 8, 1, 2.0), ("iiwa14", 32, 5, 0.0), ("iiwa14", 64, 3, 1.0),
                          ("iiwa14", 128, 2, 0.0), ("iiwa14", 256, 1, 0.0), ("indy7", 32, 1000, 2.0)]:
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1, pcg_tol=1e-8, max_pcg_iters=400)
    pr = fig8_problem(plant, N, B, f_ext_std=fstd)
    nat = NativeSolver(plant, N, B, dt=DT, **p)
    nat.set_f_ext_batch(pr["f_ext"])
    rg = nat.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    line = "%-6s N=%3d B=%4d" % (plant, N, B)
    if B <= 8:
        orc = OracleSolver(plant, N, B, dt=DT, **p)
        orc.set_f_ext_batch(pr["f_ext"])
        ro = orc.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
        err = np.abs(rg["XU"] - ro["XU"]).max() / max(1.0, np.abs(ro["XU"]).max())
        same = np.array_equal(rg["ls_step_size"], ro["ls_step_size"])
        line += "  1-iter XU rel err %.2e  steps equal %s" % (err, same)
        if not (err < 5e-4 and same):
            bad += 1
            line += "  <-- CHECK"
    p2 = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=5)
    nat2 = NativeSolver(plant, N, B, dt=DT, **p2)
    nat2.set_f_ext_batch(pr["f_ext"])
    r2 = nat2.solve(pr["xu"], DT, pr["x_s"], pr["ref"])
    ok = bool(np.all(np.isfinite(r2["XU"])) and np.all(r2["final_merit"] <= r2["initial_merit"] + 1e-3))
    line += "  5-iter merit %.3f -> %.3f finite/monotone %s  pcg mean %.1f" % (r2["initial_merit"].mean(), r2["final_merit"].mean(), ok,
                                                                          r2["pcg_iters_all"].mean())
    if not ok:
        bad += 1
        line += "  <-- CHECK"
    print(line, flush=True)
print("problems:", bad)
