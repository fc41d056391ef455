"""`BSQP` facade with the reference's constructor, methods and statistics (python/bsqp/interface.py:6-237).

Differences from the reference, all on the caller side of the hot path: no pinocchio (nq/nx/nu come from the plant, `ee_pos` uses
the solver's own kinematics: gato_ee_pos), no torch import; `model_path` is accepted for signature compatibility and only used to
auto-detect the plant when `plant_type is None`.
"""
import importlib

import numpy as np

_NQ = {"indy7": 6, "iiwa14": 7}


class BSQP:
    def __init__(self, model_path, batch_size, N, dt, max_sqp_iters=10, kkt_tol=1e-4, max_pcg_iters=100, pcg_tol=1e-4, solve_ratio=1.0, mu=1.0,
                 q_cost=2.0, qd_cost=1e-4, u_cost=1e-6, N_cost=50.0, q_lim_cost=1e-3, vel_lim_cost=0.0, ctrl_lim_cost=0.0, rho=0.0, rho_batch=None,
                 mu_batch=None, pcg_tol_batch=None, adapt_rho=True, plant_type="indy7"):
        if plant_type is None:
            plant_type = "iiwa14" if (model_path and "iiwa" in str(model_path).lower()) else "indy7"
        module_name = f"{__package__}.bsqpN{N}_{plant_type}"
        try:
            base = importlib.import_module(module_name)
        except ImportError as e:
            raise ValueError(f"Number of knots {N} not supported (could not import {module_name}): {e}")
        class_name = f"BSQP_{batch_size}_float"
        if not hasattr(base, class_name):
            raise ValueError(f"Batch size {batch_size} not supported in module {module_name}")
        self.lib = base
        self.solver_class = getattr(base, class_name)
        self.plant_type = plant_type
        self.solver = self.solver_class(dt, max_sqp_iters, kkt_tol, max_pcg_iters, pcg_tol, solve_ratio, mu, q_cost, qd_cost, u_cost, N_cost,
                                        q_lim_cost, vel_lim_cost, ctrl_lim_cost, rho)
        self.batch_size = batch_size
        self.N = N
        self.dt = dt
        self.nq = self.nv = _NQ[plant_type]
        self.nx = 2 * self.nq
        self.nu = self.nq
        self.f_ext_B = np.zeros((self.batch_size, 6), dtype=np.float32)
        self.set_f_ext_B(self.f_ext_B)
        self.XU_B = np.zeros((self.batch_size, self.N * (self.nx + self.nu) - self.nu), dtype=np.float32)
        self.stats = {"sqp_time_us": np.array([]), "sqp_iters": np.array([]), "kkt_converged": np.array([]), "pcg_iters": np.array([]),
                      "pcg_times_us": np.array([]), "min_merit": np.array([]), "step_size": np.array([]), "initial_merit": np.array([]),
                      "best_initial_merit": np.array([])}
        if rho_batch is not None:
            self.solver.set_rho_penalty_batch(np.asarray(rho_batch, dtype=np.float32).reshape(self.batch_size), True)
        self.solver.set_rho_adaptation(bool(adapt_rho))
        if mu_batch is not None:
            self.solver.set_mu_batch(np.asarray(mu_batch, dtype=np.float32).reshape(self.batch_size))
        if pcg_tol_batch is not None:
            self.solver.set_pcg_tol_batch(np.asarray(pcg_tol_batch, dtype=np.float32).reshape(self.batch_size))

    def set_cost_weights_B(self, weights_B):
        """Extension (not in the reference facade): weights_B[B,7] = q_cost, qd_cost, u_cost, N_cost, q_lim_cost, vel_lim_cost,
        ctrl_lim_cost per trajectory, so that a hyper-parameter sweep is one batch."""
        self.solver.set_cost_weights_batch(np.asarray(weights_B, dtype=np.float32).reshape(self.batch_size, 7))

    def solve(self, xcur_B, eepos_goals_B, XU_B=None):
        xcur_B = np.asarray(xcur_B, dtype=np.float32)
        eepos_goals_B = np.asarray(eepos_goals_B, dtype=np.float32)
        XU_B = self.XU_B if XU_B is None else np.asarray(XU_B, dtype=np.float32)
        XU_B[:, : self.nx] = xcur_B
        result = self.solver.solve(XU_B, self.dt, xcur_B, eepos_goals_B)

        B = self.batch_size
        st = self.stats
        self.XU_B = np.asarray(result["XU"], dtype=np.float32)
        st["sqp_time_us"] = int(result["sqp_time_us"])
        st["sqp_iters"] = np.asarray(result["sqp_iters"], dtype=np.int32).reshape(B)
        st["kkt_converged"] = np.asarray(result["kkt_converged"], dtype=np.int32).reshape(B)
        st["final_merit"] = np.asarray(result["final_merit"], dtype=np.float32).reshape(B)
        st["initial_merit"] = np.asarray(result["initial_merit"], dtype=np.float32).reshape(B)
        st["best_initial_merit"] = float(np.min(st["initial_merit"])) if st["initial_merit"].size else np.array([], dtype=np.float32)
        n = st["ls_num_iters"] = int(result.get("ls_num_iters", 0))
        st["pcg_iters"] = np.asarray(result["pcg_iters"], dtype=np.int32).reshape(n, B) if n else np.zeros((0, B), np.int32)
        st["pcg_times_us"] = np.asarray(result["pcg_times_us"], dtype=np.float32)
        st["min_merit"] = np.asarray(result["ls_min_merit"], dtype=np.float32).reshape(n, B) if n else np.zeros((0, B), np.float32)
        st["step_size"] = np.asarray(result["ls_step_size"], dtype=np.float32).reshape(n, B) if n else np.zeros((0, B), np.float32)
        if n:
            best = np.min(st["min_merit"], axis=1)
            st["best_merit_per_iter"] = best
            st["best_merit_iter1"] = float(best[0])
        else:
            st["best_merit_per_iter"] = np.array([], dtype=np.float32)
            st["best_merit_iter1"] = float("nan")
        denom = st["best_initial_merit"] if np.size(st["best_initial_merit"]) else None
        if denom and st["best_merit_per_iter"].size:
            st["best_merit_per_iter_normalized"] = st["best_merit_per_iter"] / denom
        else:
            st["best_merit_per_iter_normalized"] = st["best_merit_per_iter"]
        return self.XU_B, result["sqp_time_us"]

    def ee_pos(self, q):
        return np.asarray(self.solver.ee_pos(np.asarray(q, np.float32).reshape(1, self.nq))[0], dtype=np.float64)

    def reset(self):
        self.reset_dual()
        self.set_f_ext_B(np.zeros((self.batch_size, 6)))
        self.XU_B = np.zeros((self.batch_size, self.N * (self.nx + self.nu) - self.nu))

    def sim_forward(self, xk, uk, sim_dt):
        return self.solver.sim_forward(np.asarray(xk, dtype=np.float32), np.asarray(uk, dtype=np.float32), sim_dt)

    # ---- extensions used by the MPC loop (not in the reference facade) ----
    def select_best(self, x_last, u_last, x_meas, sim_dt):
        """(best index, errors[B]): sim_forward + per-hypothesis error + arg-min in ONE device launch (gato_select_best); what
        MPC_GATO.evaluate_best_trajectory assembles from sim_forward and numpy in the reference (mpc_controller.py:294-309)"""
        best, err = self.solver.select_best(np.asarray(x_last, np.float32), np.asarray(u_last, np.float32), np.asarray(x_meas, np.float32), sim_dt)
        return int(best), np.asarray(err)

    def plant_rk4(self, x, u_seq, f_ext6, sim_dt):
        """len(u_seq) RK4 steps of the library's own forward dynamics under a constant spatial wrench: the plant of the closed loop
        (the reference integrates pinocchio's aba, common.py:49-91)"""
        return np.asarray(self.solver.plant_rk4(np.asarray(x, np.float32), np.asarray(u_seq, np.float32).reshape(-1, self.nu),
                                                np.asarray(f_ext6, np.float32), sim_dt))

    def set_f_ext_B(self, f_ext_B):
        self.f_ext_B = np.asarray(f_ext_B, dtype=np.float32)
        self.solver.set_f_ext_batch(self.f_ext_B)

    def reset_rho(self):
        self.solver.reset_rho()

    def reset_dual(self):
        self.solver.reset_dual()

    def get_stats(self):
        return self.stats
