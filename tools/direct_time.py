"""One-iteration solve times of the three linear solvers (PCG, direct sweep, direct block cyclic reduction) over (plant, N, B); runs on the MI355X box.
-> profiles/r03_direct_solver_times.txt, DESIGN.md 5d."""
import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
for plant, N in [("indy7", 128), ("iiwa14", 128), ("indy7", 64), ("indy7", 32)]:
    for B in (1, 8, 64, 256, 1024):
        pr = fig8_problem(plant, N, B)
        row = []
        for mode, cr in (("pcg", None), ("direct", "0"), ("direct", "1")):
            if cr is not None: os.environ["GATO_DIRECT_CR"] = cr
            s = NativeSolver(plant, N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1))
            s.set_linear_solver(mode)
            s.set_profiling(True)
            ts = []
            for rep in range(6):
                s.reset_dual(); s.reset_rho()
                o = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
                ts.append(o["sqp_time_us"])
            st = s.stage_times_us()
            row.append("%s%s: %.0f us (linsolve %.0f, schur %.0f)" % (mode, "" if cr is None else ("-cr" if cr == "1" else "-sweep"), min(ts), st["pcg"], st["schur"]))
        print(plant, N, "B", B, " | ".join(row), flush=True)
