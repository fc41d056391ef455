#!/usr/bin/env python3
"""The reference's published workload, measured directly: mean `sqp_time_us` per MPC step of the figure-8 tracking loop
(examples/benchmark_fig8.py:81-90 -> MPC_GATO.run_mpc_fig8, DEFAULT_SOLVER_PARAMS: ONE SQP iteration per solve) for a row of the
solve-time heat-map (plots/gato_solve_time_heatmap.png).  Runs on the MI355X box.

    python tools/mpc_heatmap.py --knots 32 --out gpurun_out/mpc_heatmap_N32.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# plots/gato_solve_time_heatmap.png as transcribed in BASELINE.md (ms, unstated NVIDIA GPU), batch 1..512
PUBLISHED = {8: [0.09, 0.10, 0.10, 0.10, 0.10, 0.10, 0.11, 0.15, 0.29, 0.58], 16: [0.10, 0.10, 0.10, 0.10, 0.10, 0.12, 0.16, 0.31, 0.63, 1.37],
             32: [0.10, 0.10, 0.10, 0.11, 0.12, 0.17, 0.33, 0.65, 1.41, 2.84], 64: [0.12, 0.12, 0.12, 0.14, 0.19, 0.37, 0.75, 1.48, 2.95, 7.76],
             128: [0.16, 0.17, 0.19, 0.25, 0.47, 0.93, 1.71, 3.15, 9.98, 19.98]}
BATCHES = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knots", type=int, nargs="+", default=[32])
    ap.add_argument("--sim-time", type=float, default=0.6)
    ap.add_argument("--linear-solver", default="pcg", choices=["pcg", "direct"], help="direct: the block-tridiagonal direct solve (the small-batch / long-horizon mode)")
    ap.add_argument("--batches", type=int, nargs="+", default=None)
    ap.add_argument("--solve-wall", action="store_true",
                    help="a SECOND run of every cell with the host wall clock around the solve alone (GATO_MPC_TIME_SOLVE: the reference's sqp_time_us semantics, "
                         "bsqp.cuh:109,185): the figure the published heat-map shows.  A run of its own because the two extra host waits slow the step down")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mpc_heatmap.json"))
    a = ap.parse_args()
    from gato_amd.bsqp.common import figure8
    from gato_amd.bsqp.config import FIG8_DEFAULT_PARAMS, INDY7_START_CONFIGS
    from gato_amd.bsqp.mpc_controller import MPC_GATO
    fig8 = figure8(0.01, **FIG8_DEFAULT_PARAMS)
    x0 = np.concatenate([INDY7_START_CONFIGS["ready"], np.zeros(6)])
    rows = []
    for N in a.knots:
        for i, B in enumerate(BATCHES):
            if a.batches and B not in a.batches:
                continue
            np.random.seed(0)
            mpc = MPC_GATO(None, None, N=N, dt=0.01, batch_size=B, plant_type="indy7")
            mpc.solver.solver.set_linear_solver(a.linear_solver)
            # B in {2} cannot build the force estimator (it needs > 3 hypotheses, force_estimator.py:8): like the reference's benchmark
            # (benchmark_fig8.py passes no disturbance) the batch then carries identical zero-force hypotheses
            _, st = mpc.run_mpc_fig8(x0, fig8, sim_dt=0.001, sim_time=a.sim_time, solve_time_override=0.002, verbose=False)
            t = np.asarray(st["solve_times"])[1:]          # device time of the SQP solve inside the session call (hipEvents); first step dropped like the walls
            w = 1e3 * np.asarray(mpc.step_wall_s[1:])      # host wall time of the WHOLE step call: transfers in, plant, prepare, solve, selection, read-back
            pub = PUBLISHED.get(N, [None] * 10)
            wall = {}
            if a.solve_wall:
                np.random.seed(0)
                m2 = MPC_GATO(None, None, N=N, dt=0.01, batch_size=B, plant_type="indy7")
                m2.solver.solver.set_linear_solver(a.linear_solver)
                m2.time_solve_wall = True
                m2.run_mpc_fig8(x0, fig8, sim_dt=0.001, sim_time=a.sim_time, solve_time_override=0.002, verbose=False)
                sw = 1e-3 * np.asarray(m2.solve_wall_us[1:])      # ms; the first step carries one-off costs (first launches of the kernels)
                wall = dict(solve_wall_mean_ms=float(sw.mean()), solve_wall_median_ms=float(np.median(sw)), solve_wall_p95_ms=float(np.percentile(sw, 95)),
                            solve_wall_steps=int(sw.size))
            r = dict(**wall, knots=N, batch=B, linear_solver=a.linear_solver, steps=int(t.size), mean_ms=float(t.mean()), median_ms=float(np.median(t)), p95_ms=float(np.percentile(t, 95)),
                     step_wall_mean_ms=float(w.mean()), step_wall_median_ms=float(np.median(w)),
                     mean_goal_dist=float(np.mean(st["goal_distances"])), published_ms=pub[i] if i < len(pub) else None)
            rows.append(r)
            print(json.dumps(r), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
