#!/usr/bin/env python3
"""The closed MPC loop over 30 s of plant time (MPC_GATO.run_mpc_fig8: the device-resident session, one gato_mpc_step per MPC step), indy7 N = 32,
dt = 0.01, sim_dt = 1 ms, figure-8 of FIG8_DEFAULT_PARAMS from the "ready" configuration: a soak of the session at the current kernels (hundreds of
thousands of steps: everything finite, the tracking error where it was).  Runs on the MI355X box -> profiles/r06_mpc_long_run.txt."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd.bsqp.common import figure8
from gato_amd.bsqp.config import FIG8_DEFAULT_PARAMS, INDY7_START_CONFIGS
from gato_amd.bsqp.mpc_controller import MPC_GATO

fig8 = figure8(0.01, **dict(FIG8_DEFAULT_PARAMS, cycles=6))
x0 = np.concatenate([INDY7_START_CONFIGS["ready"], np.zeros(6)])
sim_time = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
for B, fz in ((1, 0.0), (32, -30.0)):
    np.random.seed(0)
    mpc = MPC_GATO(None, None, N=32, dt=0.01, batch_size=B, plant_type="indy7", constant_f_ext=[0.0, 0.0, fz, 0.0, 0.0, 0.0] if fz else None)
    t0 = time.perf_counter()
    _, st = mpc.run_mpc_fig8(x0, fig8, sim_dt=0.001, sim_time=sim_time, verbose=False)
    wall = time.perf_counter() - t0
    q = np.asarray(st["joint_positions"])
    e = np.asarray(st["goal_distances"])
    print("B=%-3d f_ext_z=%5.1f N  %7d steps  %s  tracking error mean %.4f max %.3f m  solve mean %.3f ms  wall %.1f s"
          % (B, fz, len(e), "finite" if np.all(np.isfinite(q)) else "NOT FINITE", e.mean(), e.max(), float(np.mean(st["solve_times"])), wall), flush=True)
