"""Default parameter sets of the reference's benchmarks (values of python/bsqp/config.py:8-67), as plain data."""
import numpy as np

STANDARD_BATCH_SIZES = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]
SUPPORTED_KNOT_POINTS = [8, 16, 32, 64, 128]       # CMakeLists.txt:46
SUPPORTED_PLANTS = ["indy7", "iiwa14"]

FIG8_DEFAULT_PARAMS = {"A_x": 0.4, "A_z": 0.4, "offset": [0.0, 0.5, 0.6], "period": 6, "cycles": 5, "theta": np.pi / 4}

INDY7_START_CONFIGS = {
    "zero": np.zeros(6),
    "home": np.zeros(6),
    "ready": np.array([-1.096711, -0.09903229, 0.83125766, -0.10907673, 0.49704404, 0.01499449]),
}
IIWA14_START_CONFIGS = {"zero": np.zeros(7), "home": np.zeros(7)}

DEFAULT_SOLVER_PARAMS = {
    "max_sqp_iters": 1, "kkt_tol": 0.001, "max_pcg_iters": 200, "pcg_tol": 1e-4, "solve_ratio": 1.0, "mu": 10.0, "q_cost": 2.0,
    "qd_cost": 1e-2, "u_cost": 2e-6, "N_cost": 50.0, "q_lim_cost": 0.01, "vel_lim_cost": 0.0, "ctrl_lim_cost": 0.0, "rho": 0.01,
}
PICKPLACE_SOLVER_PARAMS = {
    "max_sqp_iters": 5, "kkt_tol": 0.0, "max_pcg_iters": 100, "pcg_tol": 1e-6, "solve_ratio": 1.0, "mu": 10.0, "q_cost": 5.0,
    "qd_cost": 1e-2, "u_cost": 5e-7, "N_cost": 50.0, "q_lim_cost": 0.0, "vel_lim_cost": 0.0, "ctrl_lim_cost": 0.0, "rho": 0.001,
}
