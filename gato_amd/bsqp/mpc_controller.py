"""`MPC_GATO`: closed-loop MPC on the MI355X library -- a thin driver of the device-resident MPC session (`gato_mpc_begin` /
`gato_mpc_step`, include/gato_abi.h).

The contract taken from the reference (`python/bsqp/mpc_controller.py:18-599`) is the constructor signature, the method names and the keys of
the statistics dictionaries.  The loop itself is this library's own design: the reference's MPC step is host numpy between four solver calls
(pinocchio RK4 plant, window slide, `reset_rho`, `solve`, `sim_forward` + `argmin`, broadcast of the best row); here the measured state, the
best trajectory and the batch iterates stay on the device and ONE call per MPC step enqueues, on one stream,

    plant RK4 over the measured interval  ->  every row := best trajectory, first state := measured state, reference window,
    wrench hypotheses world frame -> last joint frame  ->  reset_rho  ->  SQP solve  ->  hypothesis selection  ->  best row

and returns {state, end effector, selected hypothesis, selection errors, device time of the solve}.  What stays on the host is what needs no
device: the clock that turns measured latencies into plant steps, the reference lookup, the force estimator (numpy) and the bookkeeping.

Semantics kept from the reference (cited where they are implemented): the plant advances by the latency of the previous step, one control per
1-kHz step taken from knot min(int(i / (dt / sim_dt)), N-1), remainders accumulate into an extra step (:199-218); every solve starts from the
best row of the previous one with its first state replaced by the measurement (:222-242); hypotheses are scored by one integrator step from the
previous state (:294-309).  The plant is the library's own rigid-body model (no pinocchio).  `pendulum_config` (:44-60, 340-360) hangs a
swinging payload from the SIMULATED arm only: the session's plant integrates arm + spherical joint + bob (`gato_mpc_set_payload`; the joint is
eliminated articulated-body fashion, kernels.hpp payload_dynamics) with the joint torque -damping x velocity formed once per plant step
(:472-478).  One deliberate difference: `initial_angle` is taken as the axis-angle vector its sampler produces (common.py:121-136); the reference
writes it into the vector part of the joint's quaternion with w = 0 (:412-418), which pinocchio's integrate then pulls towards a half turn.
`solve_time_override` (extension) fixes the simulated latency of every step so that a run is reproducible.
"""
import time

import numpy as np

from .config import DEFAULT_SOLVER_PARAMS
from .force_estimator import ForceEstimator
from .interface import BSQP


class _PlantClock:
    """Measured latencies -> whole plant steps.  mpc_controller.py:199-218: int(latency / sim_dt) steps, and the remainders (when above 10 us)
    pile up until they are worth one more step."""

    def __init__(self, sim_dt):
        self.sim_dt = float(sim_dt)
        self.now = 0.0          # simulated time
        self._carry = 0.0

    def advance(self, latency):
        """number of plant steps for this latency; moves `now` forward by them"""
        n = int(latency / self.sim_dt)
        rest = latency % self.sim_dt
        if rest > 1e-5:
            self._carry += rest
            if self._carry >= self.sim_dt:
                self._carry = 0.0
                n += 1
        self.now += n * self.sim_dt
        return n


class _Log:
    """per-step records -> the reference's statistics dictionary (lists become arrays at the end, empty ones stay lists)"""

    def __init__(self, keys):
        self.d = {k: [] for k in keys}

    def add(self, **kw):
        for k, v in kw.items():
            self.d[k].append(v)

    def finish(self, keep_lists=()):
        for k, v in self.d.items():
            if isinstance(v, list) and v and k not in keep_lists:
                self.d[k] = np.array(v)
        return self.d


class MPC_GATO:
    def __init__(self, model=None, model_path=None, N=32, dt=0.03125, batch_size=1, constant_f_ext=None, track_full_stats=False, plant_type="indy7",
                 pendulum_config=None, solver_params=None):
        cfg = dict(DEFAULT_SOLVER_PARAMS)
        cfg.update(solver_params or {})
        self.solver = BSQP(model_path=model_path, batch_size=batch_size, N=N, dt=dt, plant_type=plant_type, **cfg)
        self.solver_params, self.plant_type = cfg, plant_type
        self.pendulum_config, self.has_pendulum = pendulum_config, pendulum_config is not None
        self.pendulum_state = None      # [quat x y z w | angular velocity] of the payload after the last run
        self.nq_robot = self.nv_robot = self.solver.nq
        # dimensions of the simulated model as the reference reports them (a spherical joint adds 4 configuration and 3 velocity entries, :94-95)
        self.nq, self.nv = self.nq_robot + (4 if self.has_pendulum else 0), self.nv_robot + (3 if self.has_pendulum else 0)
        self.nx, self.nu = self.solver.nx, self.solver.nu
        self.N, self.dt, self.batch_size, self.track_full_stats = N, dt, batch_size, track_full_stats
        self.step_wall_s = []
        self.time_solve_wall = False   # True: every planning step also records the host wall clock around its solve alone (self.solve_wall_us; slower steps)
        self.solve_wall_us = []
        self.setup_external_forces(constant_f_ext)
        self.setup_force_estimator()

    # ---- disturbance and hypotheses -------------------------------------------------------------------------------------------------
    def setup_external_forces(self, constant_f_ext):
        """The disturbance is given as (linear[3], angular[3]) on the last joint in that joint's frame (mpc_controller.py:107-119: a pin.Force);
        the plant kernel takes the spatial vector [angular; linear]."""
        w = np.zeros(6) if constant_f_ext is None else np.asarray(constant_f_ext, dtype=np.float64)
        self.constant_f_ext_world = w
        self.actual_f_ext = np.concatenate([w[3:], w[:3]]).astype(np.float32)

    def setup_force_estimator(self):
        """The estimator explores a sphere of hypotheses and needs more than three of them (force_estimator.py:8); smaller batches run with the
        stored (zero) wrenches and no selection, like the reference's benchmark does when its optional estimator import is absent."""
        self.force_estimator = None
        if self.batch_size > 3:
            self.force_estimator = ForceEstimator(batch_size=self.batch_size, initial_radius=5.0, min_radius=2.0, max_radius=20.0, smoothing_factor=0.5)

    def _hypotheses(self):
        """this step's world-frame wrench hypotheses [B, 6] (None: nothing to explore)"""
        return None if self.force_estimator is None else np.asarray(self.force_estimator.generate_batch(), np.float32)

    def _placements(self, q):
        from .. import _gato_ext
        return _gato_ext.fk_placements(self.plant_type, np.asarray(q[: self.nq_robot], np.float32))

    def transform_force_to_gato_frame(self, q, f_world, placements=None):
        """Host form of what the session does per hypothesis on the device (kernels.hpp:force_to_gato_frame; mpc_controller.py:311-338): the
        world wrench (linear f[:3], angular f[3:]) moved into the last joint's frame -- SE3.actInv with that joint's world placement -- and
        then through actInv of the joint's placement in its parent; returned as [linear, angular]."""
        R, p = self._placements(q) if placements is None else placements

        def act_inv(Rf, pf, lin, ang):
            return Rf.T @ lin, Rf.T @ (ang - np.cross(pf, lin))
        f = np.asarray(f_world, dtype=np.float64)
        lin, ang = act_inv(R[-1], p[-1], f[:3], f[3:])
        lin, ang = act_inv(R[-2].T @ R[-1], R[-2].T @ (p[-1] - p[-2]), lin, ang)
        return np.concatenate([lin, ang])

    def update_force_batch(self, q):
        """Outside a session: draw the estimator's hypotheses, move them into the last joint's frame at q and hand them to the solver."""
        hyp = self._hypotheses()
        if hyp is None:
            return
        pl = self._placements(q)
        self.solver.set_f_ext_B(np.stack([self.transform_force_to_gato_frame(q, h, pl) for h in hyp]))

    def evaluate_best_trajectory(self, x_last, u_last, x_curr, dt):
        """Outside a session: score the stored hypotheses by one integrator step from (x_last, u_last) against x_curr (one device launch) and
        feed the estimator; returns the winner (0 without an estimator)."""
        if self.force_estimator is None:
            return 0
        best, errors = self.solver.select_best(x_last, u_last, x_curr, dt)
        self.force_estimator.update(best, errors, alpha=0.6, beta=0.5)
        return best

    # ---- the session ------------------------------------------------------------------------------------------------------------------
    def _begin(self, x_start, window):
        """device state := x_start, warm start, duals cleared; one solve on the first window before the clock starts (mpc_controller.py:170-176)"""
        dev = self.solver.solver
        dev.mpc_begin(np.asarray(x_start, np.float32))
        dev.mpc_set_payload(self._payload_at_rest())
        return dev.mpc_step(advance=False, plan=True, plant_steps=0, sim_dt=0.0, steps_per_knot=1.0, plant_wrench=None, ref_window=window,
                            hyp_world=self._hypotheses(), select=False, select_dt=0.0)

    def _payload_at_rest(self):
        """[quat x y z w | w = 0 | mass, length, damping, inertia] from pendulum_config (defaults of mpc_controller.py:342-343, 417, 474; the bob's
        own inertia 0.001 of :354), None without one"""
        if not self.has_pendulum:
            return None
        c = self.pendulum_config
        aa = np.asarray(c.get("initial_angle", [0.3, 0.0, 0.0]), np.float64).reshape(3)
        ang = float(np.linalg.norm(aa))
        quat = np.array([0.0, 0.0, 0.0, 1.0]) if ang < 1e-12 else np.concatenate([np.sin(ang / 2) * aa / ang, [np.cos(ang / 2)]])
        return np.concatenate([quat, np.zeros(3), [c.get("mass", 15.0), c.get("length", 0.3), c.get("damping", 0.4), 0.001]]).astype(np.float32)

    def _end(self):
        if self.has_pendulum:
            self.pendulum_state = np.asarray(self.solver.solver.mpc_payload(), np.float64)

    def _step(self, advance, plan, nsteps, sim_dt, window, latency):
        """one call into the session.  The selection integrates over the latency rounded to whole plant steps (mpc_controller.py:239)."""
        dev = self.solver.solver
        hyp = self._hypotheses() if plan else None
        t0 = time.perf_counter()
        out = dev.mpc_step(advance=advance, plan=plan, plant_steps=nsteps, sim_dt=sim_dt, steps_per_knot=self.dt / sim_dt, plant_wrench=self.actual_f_ext,
                           ref_window=window, hyp_world=hyp, select=plan and hyp is not None, select_dt=max(sim_dt, round(latency / sim_dt) * sim_dt),
                           time_solve_wall=bool(getattr(self, "time_solve_wall", False)))
        out["wall_s"] = time.perf_counter() - t0
        if getattr(self, "time_solve_wall", False) and plan:
            # measurement mode (tools/mpc_heatmap.py --solve-wall): the host wall clock around the solve alone, device-synchronised on both sides --
            # the reference's own `sqp_time_us` (bsqp.cuh:109,185), which is what its published solve-time heat-map shows
            self.solve_wall_us.append(out["solve_wall_us"])
        self.step_wall_s.append(out["wall_s"])   # host wall time of every session call (not a statistics key of the reference)
        # The CONTROLLER's latency: the reference charges the time around solver.solve only (mpc_controller.py:234-236).  Simulating the plant
        # is not controller time -- fed back as latency it adds plant steps, which add wall time (the payload plant is several times as
        # expensive per step) -- so the plant launch's device time is taken out of the call's wall time, and an advance-only call counts nothing.
        out["latency_s"] = max(0.0, out["wall_s"] - 1e-6 * out.get("plant_us", 0.0)) if plan else 0.0
        if plan and hyp is not None:
            self.force_estimator.update(out["best"], np.asarray(out["errors"]), alpha=0.6, beta=0.5)
        return out

    def _iteration_counts(self):
        st = self.solver.solver.last_stats()
        pcg = np.asarray(st["pcg_iters_all"]).reshape(-1)
        return int(np.asarray(st["sqp_iters"]).reshape(-1)[0]), (int(pcg[0]) if pcg.size else 0)

    # ---- figure-8 tracking ------------------------------------------------------------------------------------------------------------
    def run_mpc_fig8(self, x_start, fig8_traj, sim_dt=0.001, sim_time=5.0, solve_time_override=None, verbose=True):
        """Track a sampled figure-8 (6 floats per sample, one sample per dt).  Returns (None, stats) with the statistics keys of
        mpc_controller.py:136-277.  Every MPC step is ONE session call: the plant catches up with the previous step's latency, the window of the
        N samples from the plant's new time is solved for, the best hypothesis becomes the next plan."""
        fig8 = np.asarray(fig8_traj, np.float32).reshape(-1, 6)
        log = _Log(["timestamps", "solve_times", "goal_distances", "ee_actual", "joint_positions", "joint_velocities"] + (["sqp_iters"] if self.track_full_stats else []))
        clock = _PlantClock(sim_dt)
        self._begin(x_start, fig8[: self.N])
        last_sample = len(fig8) - 6 * self.N          # the reference stops 6 N samples before the end of the track (:221)
        latency = self.dt
        if verbose:
            print(f"MPC session: figure-8, {self.plant_type} N={self.N} batch={self.batch_size}, {sim_time} s of plant time")
        while clock.now < sim_time:
            nsteps = clock.advance(latency)
            k0 = int(clock.now / self.dt)
            if k0 >= last_sample:
                break
            window = fig8[k0: k0 + self.N]
            out = self._step(True, True, nsteps, sim_dt, window, latency)
            latency = out["latency_s"] if solve_time_override is None else float(solve_time_override)
            x = np.asarray(out["x"], np.float64)
            ee = np.asarray(out["ee"], np.float64)
            log.add(timestamps=clock.now, solve_times=out["solve_us"] / 1000.0, goal_distances=float(np.linalg.norm(ee - window[1, :3])), ee_actual=ee,
                    joint_positions=x[: self.nq_robot], joint_velocities=x[self.nq_robot:])
            if self.track_full_stats:
                log.add(sqp_iters=self._iteration_counts()[0])
        self._end()
        stats = log.finish()
        if verbose and len(stats["goal_distances"]):
            print(f"  mean tracking error {np.mean(stats['goal_distances']) * 1e3:.1f} mm, mean solve {np.mean(stats['solve_times']):.3f} ms over {len(stats['timestamps'])} steps")
        return None, stats

    # ---- goal sequence ----------------------------------------------------------------------------------------------------------------
    def run_mpc_goals(self, x_start, goals, sim_dt=0.001, goal_timeout=5.0, goal_threshold=0.05, velocity_threshold=1.0, solve_time_override=None,
                      verbose=True):
        """Drive the end effector through a list of goals (statistics keys of mpc_controller.py:361-599).  A goal counts as reached within
        goal_threshold metres at |qd|_1 < velocity_threshold and is given up after goal_timeout seconds.  The decision needs the state the
        plant has just reached, so a step is two session calls: advance (state and end effector come back), decide, plan."""
        goals = [np.asarray(g, np.float64) for g in goals]
        log = _Log(["timestamps", "solve_times", "goal_distances", "ee_actual", "joint_positions", "joint_velocities", "best_trajectory_id"] +
                   (["sqp_iters", "pcg_iters"] if self.track_full_stats else []))
        outcome, reached_at = ["not_reached"] * len(goals), [None] * len(goals)

        def window_of(goal):
            w = np.zeros((self.N, 6), np.float32)
            w[:, :3] = goal
            return w
        clock = _PlantClock(sim_dt)
        cur, since = 0, 0.0
        window = window_of(goals[cur])
        self._begin(x_start, window)
        latency = self.dt
        if verbose:
            print(f"MPC session: {len(goals)} goals, {self.plant_type} N={self.N} batch={self.batch_size}")
        while clock.now < goal_timeout * len(goals):
            nsteps = clock.advance(latency)
            at = self._step(True, False, nsteps, sim_dt, None, latency)
            x, ee = np.asarray(at["x"], np.float64), np.asarray(at["ee"], np.float64)
            dist = float(np.linalg.norm(ee - goals[cur]))
            arrived = dist < goal_threshold and np.abs(x[self.nq_robot:]).sum() < velocity_threshold
            if arrived or clock.now - since >= goal_timeout:
                outcome[cur] = "reached" if arrived else "timeout"
                if arrived:
                    reached_at[cur] = clock.now
                cur += 1
                if cur == len(goals):
                    break
                window, since = window_of(goals[cur]), clock.now
            out = self._step(False, True, 0, sim_dt, window, latency)
            latency = out["latency_s"] if solve_time_override is None else float(solve_time_override)   # the advance-only call is plant simulation
            log.add(timestamps=clock.now, solve_times=out["solve_us"] / 1000.0, goal_distances=dist, ee_actual=ee, joint_positions=x[: self.nq_robot],
                    joint_velocities=x[self.nq_robot:], best_trajectory_id=out["best"])
            if self.track_full_stats:
                sqp, pcg = self._iteration_counts()
                log.add(sqp_iters=sqp, pcg_iters=pcg)
        self._end()
        stats = log.finish()
        stats["goal_outcomes"], stats["goal_reached_times"] = outcome, reached_at
        stats["time_to_all_reached"] = float(max(reached_at)) if all(o == "reached" for o in outcome) else None
        if verbose:
            print(f"  {outcome.count('reached')} of {len(goals)} goals reached" + (f", mean solve {np.mean(stats['solve_times']):.3f} ms" if len(stats["solve_times"]) else ""))
        return None, stats
