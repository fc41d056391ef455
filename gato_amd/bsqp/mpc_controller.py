"""Closed-loop MPC around the batched SQP solver: the host-side mirror of the reference's `python/bsqp/mpc_controller.py`
(`MPC_GATO`, :18-599) WITHOUT pinocchio.

What the reference does with pinocchio is done here with the library's own rigid-body code:
  * the plant simulator -- RK4 over the forward dynamics with a wrench on the last link, 1 kHz (`common.rk4`, common.py:49-91) -- is
    `gato_plant_rk4` (one device launch per simulated interval; the plant therefore has the solver's model, incl. the indy7 table's
    COM-less inertias);
  * `transform_force_to_gato_frame` (mpc_controller.py:311-338) uses `gato_fk_placements` for the joint placements data.oMi;
  * `evaluate_best_trajectory` (:294-309) is ONE launch (`gato_select_best`: sim_forward, per-hypothesis error, arg-min);
  * `BSQP.ee_pos` is `gato_ee_pos`.
Same loop structure, statistics keys and defaults as the reference.  The pendulum payload (`pendulum_config`, a spherical joint added
to the pinocchio model) is not available: the library's plants are the two serial arms.  `solve_time_override` (an extension) fixes
the simulated duration of every solve so that a run is reproducible; by default the measured wall time of the solve is used, as in the
reference (mpc_controller.py:236-238)."""
import time

import numpy as np

from .config import DEFAULT_SOLVER_PARAMS
from .force_estimator import ForceEstimator
from .interface import BSQP


def _act_inv(R, p, lin, ang):
    """pinocchio SE3.actInv on a Force(linear, angular): the same wrench expressed in the frame (R, p)"""
    return R.T @ lin, R.T @ (ang - np.cross(p, lin))


class MPC_GATO:
    def __init__(self, model=None, model_path=None, N=32, dt=0.03125, batch_size=1, constant_f_ext=None, track_full_stats=False, plant_type="indy7",
                 pendulum_config=None, solver_params=None):
        if pendulum_config is not None:
            raise NotImplementedError("pendulum_config needs a pinocchio model with a spherical joint; the MI355X library simulates the arm itself")
        solver_cfg = DEFAULT_SOLVER_PARAMS.copy()
        if solver_params is not None:
            solver_cfg.update(solver_params)
        self.solver = BSQP(model_path=model_path, batch_size=batch_size, N=N, dt=dt, plant_type=plant_type, **solver_cfg)
        self.solver_params = solver_cfg
        self.plant_type = plant_type
        self.has_pendulum = False
        self.nq = self.nv = self.nq_robot = self.nv_robot = self.solver.nq
        self.nx, self.nu = self.solver.nx, self.solver.nu
        self.N, self.dt, self.batch_size = N, dt, batch_size
        self.track_full_stats = track_full_stats
        self.setup_external_forces(constant_f_ext)
        self.setup_force_estimator()

    # ---- plant ----
    def setup_external_forces(self, constant_f_ext):
        """mpc_controller.py:107-119: the disturbance acts on the last joint as pin.Force(f[:3] linear, f[3:] angular) in that joint's
        frame; the library's dynamics take spatial vectors [angular; linear]"""
        self.constant_f_ext_world = np.zeros(6) if constant_f_ext is None else np.asarray(constant_f_ext, dtype=np.float64)
        f = self.constant_f_ext_world
        self.actual_f_ext = np.concatenate([f[3:], f[:3]]).astype(np.float32)

    def setup_force_estimator(self):
        # the estimator needs > 3 hypotheses (force_estimator.py:8); the reference's benchmark runs batch 2 without one (its import of
        # examples/force_estimator.py is optional, mpc_controller.py:10-14): smaller batches carry identical zero-force hypotheses
        if self.batch_size > 3:
            self.force_estimator = ForceEstimator(batch_size=self.batch_size, initial_radius=5.0, min_radius=2.0, max_radius=20.0, smoothing_factor=0.5)
        else:
            self.force_estimator = None

    def _simulate(self, q, dq, XU_best, timestep, sim_dt, state):
        """the plant between two solves (mpc_controller.py:199-218): int(timestep / sim_dt) RK4 steps, the control of knot
        min(int(i / (dt / sim_dt)), N-1) at step i, plus one step whenever the accumulated remainders reach sim_dt"""
        nsteps = int(timestep / sim_dt)
        idx = [min(int(i / (self.dt / sim_dt)), self.N - 1) for i in range(nsteps)]
        if timestep % sim_dt > 1e-5:
            state["accumulated"] += timestep % sim_dt
            if state["accumulated"] >= sim_dt:
                state["accumulated"] = 0.0
                idx.append(min(int(nsteps / (self.dt / sim_dt)), self.N - 1))
        if idx:
            ks = self.nx + self.nu
            u_seq = np.stack([XU_best[self.nx + ks * k: self.nx + ks * k + self.nu] for k in idx])
            x = self.solver.plant_rk4(np.concatenate([q, dq]), u_seq, self.actual_f_ext, sim_dt)
            q, dq = x[: self.nq].astype(np.float64), x[self.nq:].astype(np.float64)
        return q, dq, len(idx) * sim_dt

    # ---- force hypotheses ----
    def update_force_batch(self, q):
        if self.batch_size == 1 or self.force_estimator is None:
            return
        force_batch = self.force_estimator.generate_batch()
        placements = self._placements(q)                       # one forward-kinematics pass serves every hypothesis
        transformed = np.zeros_like(force_batch)
        for i in range(self.batch_size):
            transformed[i, :] = self.transform_force_to_gato_frame(q, force_batch[i, :], placements)
        self.solver.set_f_ext_B(transformed)

    def evaluate_best_trajectory(self, x_last, u_last, x_curr, dt):
        if self.batch_size == 1 or self.force_estimator is None:
            return 0
        best_id, errors = self.solver.select_best(x_last, u_last, x_curr, dt)
        self.force_estimator.update(best_id, errors, alpha=0.6, beta=0.5)
        return best_id

    def _placements(self, q):
        from .. import _gato_ext
        return _gato_ext.fk_placements(self.plant_type, np.asarray(q[: self.nq], np.float32))

    def transform_force_to_gato_frame(self, q, f_world, placements=None):
        """mpc_controller.py:311-338 with the library's joint placements: the world wrench (linear f[:3], angular f[3:]) expressed in
        the last joint's frame, then `actInv` of the placement of that frame in its parent joint's frame; returned as
        [linear, angular] like the reference does."""
        R, p = self._placements(q) if placements is None else placements
        R_ee, p_ee, R_pj, p_pj = R[-1], p[-1], R[-2], p[-2]
        f_world = np.asarray(f_world, dtype=np.float64)
        lin, ang = _act_inv(R_ee, p_ee, f_world[:3], f_world[3:])
        R_rel, p_rel = R_pj.T @ R_ee, R_pj.T @ (p_ee - p_pj)          # oMi[parent].inverse() * oMi[ee]
        lin, ang = _act_inv(R_rel, p_rel, lin, ang)
        return np.concatenate([lin, ang])

    # ---- loops ----
    def _warm_start(self, x_curr):
        XU = np.zeros(self.N * (self.nx + self.nu) - self.nu)
        for i in range(self.N):
            XU[i * (self.nx + self.nu): i * (self.nx + self.nu) + self.nx] = x_curr
        return np.tile(XU, (self.batch_size, 1))

    def run_mpc_fig8(self, x_start, fig8_traj, sim_dt=0.001, sim_time=5.0, solve_time_override=None, verbose=True):
        """mpc_controller.py:136-277: track a figure-8; returns (None, stats) with the reference's statistics keys"""
        stats = {"timestamps": [], "solve_times": [], "goal_distances": [], "ee_actual": [], "joint_positions": [], "joint_velocities": []}
        if self.track_full_stats:
            stats["sqp_iters"] = []
        total_sim_time = 0.0
        sim_state = {"accumulated": 0.0}
        x_curr = np.asarray(x_start, dtype=np.float64)
        q, dq = x_curr[: self.nq].copy(), x_curr[self.nq: self.nx].copy()
        x_curr_batch = np.tile(x_curr, (self.batch_size, 1))
        ee_g_batch = np.tile(fig8_traj[: 6 * self.N], (self.batch_size, 1))
        XU_batch = self._warm_start(x_curr)
        self.solver.reset_dual()
        self.update_force_batch(q)
        XU_batch, _ = self.solver.solve(x_curr_batch, ee_g_batch, XU_batch)          # warm-up solve
        XU_batch = np.array(XU_batch, dtype=np.float64)
        XU_best = XU_batch[0, :].copy()
        if verbose:
            print(f"\nRunning MPC: N={self.N}, batch={self.batch_size}, time={sim_time}s")
        solve_time = self.dt
        while total_sim_time < sim_time:
            x_last, u_last = x_curr, XU_best[self.nx: self.nx + self.nu].copy()
            timestep = solve_time
            q, dq, advanced = self._simulate(q, dq, XU_best, timestep, sim_dt, sim_state)
            total_sim_time += advanced
            x_curr = np.concatenate([q, dq])
            eepos_offset = int(total_sim_time / self.dt)
            if eepos_offset >= len(fig8_traj) / 6 - 6 * self.N:
                break
            x_curr_batch = np.tile(x_curr, (self.batch_size, 1))
            ee_g = fig8_traj[6 * eepos_offset: 6 * (eepos_offset + self.N)]
            ee_g_batch[:, :] = ee_g
            XU_batch[:, : self.nx] = x_curr
            self.update_force_batch(q)
            self.solver.reset_rho()
            start = time.time()
            XU_batch_new, gpu_solve_time = self.solver.solve(x_curr_batch, ee_g_batch, XU_batch)
            solve_time = time.time() - start if solve_time_override is None else float(solve_time_override)
            best_id = self.evaluate_best_trajectory(x_last, u_last, x_curr, max(sim_dt, round(timestep / sim_dt) * sim_dt))
            XU_best = np.array(XU_batch_new[best_id, :], dtype=np.float64)
            XU_batch[:, :] = XU_best
            ee_pos = self.solver.ee_pos(q)
            stats["timestamps"].append(total_sim_time)
            stats["solve_times"].append(gpu_solve_time / 1000.0)
            stats["goal_distances"].append(np.linalg.norm(ee_pos[:3] - ee_g[6:9]))
            stats["ee_actual"].append(ee_pos.copy())
            stats["joint_positions"].append(q.copy())
            stats["joint_velocities"].append(dq.copy())
            if self.track_full_stats:
                stats["sqp_iters"].append(int(np.atleast_1d(self.solver.get_stats()["sqp_iters"])[0]))
        for key in stats:
            if stats[key]:
                stats[key] = np.array(stats[key])
        if verbose and len(stats["goal_distances"]):
            print(f"Avg error: {np.mean(stats['goal_distances']):.4f}m")
            print(f"Avg solve time: {np.mean(stats['solve_times']):.3f}ms")
        return None, stats

    def run_mpc_goals(self, x_start, goals, sim_dt=0.001, goal_timeout=5.0, goal_threshold=0.05, velocity_threshold=1.0, solve_time_override=None,
                      verbose=True):
        """mpc_controller.py:361-599: drive the end effector through discrete goals (reached = within goal_threshold with
        |qd|_1 < velocity_threshold; a goal is abandoned after goal_timeout seconds)"""
        stats = {"timestamps": [], "solve_times": [], "goal_distances": [], "ee_actual": [], "joint_positions": [], "joint_velocities": [],
                 "best_trajectory_id": []}
        if self.track_full_stats:
            stats["sqp_iters"], stats["pcg_iters"] = [], []
        stats["goal_outcomes"] = ["not_reached"] * len(goals)
        stats["goal_reached_times"] = [None] * len(goals)
        stats["time_to_all_reached"] = None
        total_sim_time = 0.0
        sim_state = {"accumulated": 0.0}
        x_curr = np.asarray(x_start, dtype=np.float64)
        q, dq = x_curr[: self.nq].copy(), x_curr[self.nq: self.nx].copy()
        x_curr_batch = np.tile(x_curr, (self.batch_size, 1))
        current_goal_idx = 0
        current_goal = np.asarray(goals[current_goal_idx], dtype=np.float64)
        ee_g = np.tile(np.concatenate([current_goal, np.zeros(3)]), self.N)
        ee_g_batch = np.tile(ee_g, (self.batch_size, 1))
        self.solver.reset_dual()
        XU_batch = self._warm_start(x_curr)
        self.update_force_batch(q)
        XU_batch, _ = self.solver.solve(x_curr_batch, ee_g_batch, XU_batch)
        XU_batch = np.array(XU_batch, dtype=np.float64)
        XU_best = XU_batch[0, :].copy()
        if verbose:
            print(f"\nRunning MPC: N={self.N}, batch={self.batch_size}, {len(goals)} goals")
        goal_start_time = total_sim_time
        solve_time = self.dt
        while total_sim_time < goal_timeout * len(goals):
            x_last, u_last = x_curr, XU_best[self.nx: self.nx + self.nu].copy()
            timestep = solve_time
            q, dq, advanced = self._simulate(q, dq, XU_best, timestep, sim_dt, sim_state)
            total_sim_time += advanced
            x_curr = np.concatenate([q, dq])
            ee_pos = self.solver.ee_pos(q)
            current_dist = np.linalg.norm(ee_pos - current_goal)
            reached = (current_dist < goal_threshold) and (np.linalg.norm(dq, ord=1) < velocity_threshold)
            timeout = (total_sim_time - goal_start_time) >= goal_timeout
            if reached or timeout:
                if reached:
                    stats["goal_outcomes"][current_goal_idx] = "reached"
                    stats["goal_reached_times"][current_goal_idx] = total_sim_time
                else:
                    stats["goal_outcomes"][current_goal_idx] = "timeout"
                current_goal_idx += 1
                if current_goal_idx >= len(goals):
                    break
                current_goal = np.asarray(goals[current_goal_idx], dtype=np.float64)
                ee_g = np.tile(np.concatenate([current_goal, np.zeros(3)]), self.N)
                goal_start_time = total_sim_time
                self.solver.reset_rho()
            x_curr_batch = np.tile(x_curr, (self.batch_size, 1))
            ee_g_batch[:, :] = ee_g
            XU_batch[:, : self.nx] = x_curr
            self.update_force_batch(q)
            self.solver.reset_rho()
            start = time.time()
            XU_batch_new, gpu_solve_time = self.solver.solve(x_curr_batch, ee_g_batch, XU_batch)
            solve_time = time.time() - start if solve_time_override is None else float(solve_time_override)
            best_id = self.evaluate_best_trajectory(x_last, u_last, x_curr, max(sim_dt, round(timestep / sim_dt) * sim_dt))
            XU_best = np.array(XU_batch_new[best_id, :], dtype=np.float64)
            XU_batch[:, :] = XU_best
            stats["timestamps"].append(total_sim_time)
            stats["solve_times"].append(gpu_solve_time / 1000.0)
            stats["goal_distances"].append(current_dist)
            stats["ee_actual"].append(ee_pos.copy())
            stats["joint_positions"].append(q.copy())
            stats["joint_velocities"].append(dq.copy())
            stats["best_trajectory_id"].append(best_id)
            if self.track_full_stats:
                st = self.solver.get_stats()
                stats["sqp_iters"].append(int(np.atleast_1d(st["sqp_iters"])[0]))
                pcg = st.get("pcg_iters", [])
                stats["pcg_iters"].append(int(np.asarray(pcg).reshape(-1)[0]) if np.size(pcg) else 0)
        for key in stats:
            if isinstance(stats[key], list) and stats[key] and key not in ("goal_outcomes", "goal_reached_times", "time_to_all_reached"):
                stats[key] = np.array(stats[key])
        if all(o == "reached" for o in stats["goal_outcomes"]):
            stats["time_to_all_reached"] = float(np.max([t for t in stats["goal_reached_times"] if t is not None]))
        if verbose:
            print(f"Goals reached: {sum(1 for o in stats['goal_outcomes'] if o == 'reached')}/{len(goals)}")
            if len(stats["solve_times"]) > 0:
                print(f"Avg solve time: {np.mean(stats['solve_times']):.3f}ms")
        return None, stats
