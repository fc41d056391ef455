"""The callers either side of the hot path: plant simulator, force transformation, hypothesis selection and the
closed MPC loop of python/bsqp/mpc_controller.py on the library's own rigid-body code (no pinocchio)."""
import numpy as np
import pytest

from gato_amd.bsqp.common import figure8
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS, FIG8_DEFAULT_PARAMS, INDY7_START_CONFIGS


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


@pytest.mark.parametrize("plant", ["indy7", "iiwa14"])
def test_fk_placements_are_a_consistent_revolute_chain(plant):
    """gato_fk_placements (host code): orthonormal frames, the last origin is the end effector of the oracle's kinematics, and
    turning joint k moves every later origin on a circle about joint k's own z axis (what data.oMi means in pinocchio)"""
    from gato_amd._lib import fk_placements
    from oracle import oracle as O
    nq = 6 if plant == "indy7" else 7
    rng = np.random.default_rng(2)
    q = rng.uniform(-1.5, 1.5, nq).astype(np.float32)
    R, p = fk_placements(plant, q)
    for k in range(nq):
        assert np.abs(R[k] @ R[k].T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R[k]) - 1) < 1e-12
    np.testing.assert_allclose(p[-1], O.ee(plant, q)[0], atol=2e-6)
    for k in range(nq - 1):
        d = 0.37
        q2 = q.copy(); q2[k] += d
        R2, p2 = fk_placements(plant, q2)
        np.testing.assert_allclose(p2[:k + 1], p[:k + 1], atol=1e-6)                       # joints up to k do not move
        rot = _rot(R[k][:, 2], float(q2[k]) - float(q[k]))
        for j in range(k + 1, nq):
            np.testing.assert_allclose(p2[j] - p[k], rot @ (p[j] - p[k]), atol=2e-6)
            np.testing.assert_allclose(R2[j], rot @ R[j], atol=2e-6)


def test_transform_force_to_gato_frame_formula():
    """MPC_GATO.transform_force_to_gato_frame (mpc_controller.py:311-338) = two SE3.actInv's; checked against homogeneous
    adjoint algebra written independently (wrench row-vector convention)"""
    from gato_amd._lib import fk_placements
    from gato_amd.bsqp.mpc_controller import MPC_GATO
    q = np.array([0.3, -0.7, 1.1, 0.2, -0.4, 0.9])
    f = np.array([3.0, -2.0, 5.0, 0.4, 0.1, -0.3])
    R, p = fk_placements("indy7", q.astype(np.float32))

    def wrench_in_frame(Rf, pf, lin, ang):
        # moment about the frame's origin, both rotated into the frame
        return Rf.T @ lin, Rf.T @ (ang + np.cross(lin, pf))
    lin, ang = wrench_in_frame(R[-1], p[-1], f[:3], f[3:])
    Rr, pr = R[-2].T @ R[-1], R[-2].T @ (p[-1] - p[-2])
    lin, ang = wrench_in_frame(Rr, pr, lin, ang)
    got = MPC_GATO.transform_force_to_gato_frame(None, q, f, placements=(R, p))
    np.testing.assert_allclose(got, np.concatenate([lin, ang]), atol=1e-12)
    assert abs(np.linalg.norm(got[:3]) - np.linalg.norm(f[:3])) < 1e-12                     # the force part is only rotated


@pytest.mark.gpu
@pytest.mark.parametrize("plant", ["indy7", "iiwa14"])
def test_plant_rk4_against_numpy_rk4_over_the_oracle_dynamics(plant):
    """gato_plant_rk4 (the MPC loop's plant, common.py:49-91 semantics) vs the same RK4 scheme in numpy over the oracle's forward dynamics"""
    from gato_amd._lib import NativeSolver
    from oracle import oracle as O
    s = NativeSolver(plant, 8, 1)
    nq = s.nq
    rng = np.random.default_rng(4)
    q, v = rng.uniform(-0.8, 0.8, nq), rng.uniform(-0.5, 0.5, nq)
    fe = rng.normal(0, 4.0, 6)
    useq = rng.uniform(-6, 6, (7, nq))
    h = 0.001
    x = s.plant_rk4(np.concatenate([q, v]), useq, fe, h)
    for u in useq:
        fd = lambda qq, vv: O.fd(plant, qq, vv, u, fe).astype(np.float64)  # noqa: E731
        k1v = fd(q, v)
        k2q = v + k1v * h / 2; k2v = fd(q + v * h / 2, k2q)
        k3q = v + k2v * h / 2; k3v = fd(q + k2q * h / 2, k3q)
        k4q = v + k3v * h; k4v = fd(q + k3q * h, k4q)
        q = q + h * (v + 2 * k2q + 2 * k3q + k4q) / 6
        v = v + (h / 6) * (k1v + 2 * k2v + 2 * k3v + k4v)
    np.testing.assert_allclose(x, np.concatenate([q, v]), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("plant,N,B", [("indy7", 16, 8), ("iiwa14", 8, 5)])
def test_mpc_session_step_equals_the_separate_calls(plant, N, B):
    """gato_mpc_step (plant -> prepare -> reset_rho -> solve -> selection -> best row, device-resident, one call) against the same step
    assembled from the separate entry points the way the reference's loop does it (mpc_controller.py:196-242): plant_rk4 with the controls
    picked by knot index, rows := best with the measured first state, reset_rho, solve, select_best, best row -- bit for bit over three steps."""
    from gato_amd._lib import NativeSolver
    from oracle import oracle as O
    dt, sim_dt = 0.01, 0.001
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=2)
    nq = 6 if plant == "indy7" else 7
    nx, nu, ks = 2 * nq, nq, 3 * nq
    rng = np.random.default_rng(5)
    x0 = np.concatenate([(INDY7_START_CONFIGS["ready"] if plant == "indy7" else rng.uniform(-0.4, 0.4, nq)), np.zeros(nq)]).astype(np.float32)
    fext = rng.normal(0, 3.0, (B, 6)).astype(np.float32)            # stored wrench hypotheses (hyp_world = None keeps them)
    wrench = np.array([0.2, -0.1, 0.3, 2.0, -4.0, 6.0], np.float32)  # what really acts on the plant, [angular; linear]
    ses, man = NativeSolver(plant, N, B, dt=dt, **p), NativeSolver(plant, N, B, dt=dt, **p)
    for s in (ses, man):
        s.set_f_ext_batch(fext)
    e0 = O.ee(plant, x0[:nq])[0]
    windows = [np.tile(np.concatenate([e0 + np.array([0.02 * (i + 1), -0.01 * i, 0.015 * i], np.float32), np.zeros(3, np.float32)]), (N, 1)) for i in range(4)]
    # session: warm-up plan, then three fused steps
    ses.mpc_begin(x0)
    ses.mpc_step(advance=False, plan=True, ref_window=windows[0])
    # by hand
    xu = np.zeros((B, ks * N - nu), np.float32)
    for k in range(N):
        xu[:, k * ks: k * ks + nx] = x0
    man.reset_dual()
    r = man.solve(xu, dt, np.tile(x0, (B, 1)), np.tile(windows[0].reshape(-1), (B, 1)))
    best_row = r["XU"][0].copy()
    np.testing.assert_array_equal(ses.mpc_best(), best_row)
    x = x0.copy()
    for i, nsteps in enumerate((10, 7, 23)):
        out = ses.mpc_step(advance=True, plan=True, plant_steps=nsteps, sim_dt=sim_dt, steps_per_knot=dt / sim_dt, plant_wrench=wrench,
                           ref_window=windows[i + 1], select=True, select_dt=nsteps * sim_dt)
        idx = [min(int(j / (dt / sim_dt)), N - 1) for j in range(nsteps)]
        useq = np.stack([best_row[nx + ks * k: nx + ks * k + nu] for k in idx])
        x_last, u_last = x.copy(), best_row[nx: nx + nu].copy()
        x = man.plant_rk4(x, useq, wrench, sim_dt)
        rows = np.tile(best_row, (B, 1))
        rows[:, :nx] = x
        man.reset_rho()
        r = man.solve(rows, dt, np.tile(x, (B, 1)), np.tile(windows[i + 1].reshape(-1), (B, 1)))
        best, err = man.select_best(x_last, u_last, x, nsteps * sim_dt)
        best_row = r["XU"][best].copy()
        np.testing.assert_array_equal(out["x"], x)
        assert out["best"] == best
        np.testing.assert_array_equal(out["errors"], err)
        np.testing.assert_array_equal(ses.mpc_best(), best_row)
        np.testing.assert_allclose(out["ee"], O.ee(plant, x[:nq])[0], atol=2e-6)
        assert out["solve_us"] > 0
    # the two-call form of a step (advance, decide on the host, plan) is the same step
    two = NativeSolver(plant, N, B, dt=dt, **p)
    two.set_f_ext_batch(fext)
    two.mpc_begin(x0)
    two.mpc_step(advance=False, plan=True, ref_window=windows[0])
    a = two.mpc_step(advance=True, plan=False, plant_steps=10, sim_dt=sim_dt, steps_per_knot=dt / sim_dt, plant_wrench=wrench)
    b = two.mpc_step(advance=False, plan=True, ref_window=windows[1], select=True, select_dt=10 * sim_dt)
    one = NativeSolver(plant, N, B, dt=dt, **p)
    one.set_f_ext_batch(fext)
    one.mpc_begin(x0)
    one.mpc_step(advance=False, plan=True, ref_window=windows[0])
    c = one.mpc_step(advance=True, plan=True, plant_steps=10, sim_dt=sim_dt, steps_per_knot=dt / sim_dt, plant_wrench=wrench, ref_window=windows[1], select=True,
                     select_dt=10 * sim_dt)
    np.testing.assert_array_equal(a["x"], c["x"])
    assert b["best"] == c["best"]
    np.testing.assert_array_equal(two.mpc_best(), one.mpc_best())


@pytest.mark.gpu
@pytest.mark.parametrize("plant", ["indy7", "iiwa14"])
def test_session_moves_the_hypotheses_into_the_last_joint_frame(plant):
    """the device form of transform_force_to_gato_frame (kernels.hpp:force_to_gato_frame, fp32) against the host formula over float64
    placements (mpc_controller.py:311-338): the wrenches the session stores for the solve"""
    from gato_amd._lib import NativeSolver, fk_placements
    from gato_amd.bsqp.mpc_controller import MPC_GATO
    N, B = 8, 6
    nq = 6 if plant == "indy7" else 7
    rng = np.random.default_rng(8)
    x0 = np.concatenate([rng.uniform(-1.2, 1.2, nq), np.zeros(nq)]).astype(np.float32)
    hyp = rng.normal(0, 8.0, (B, 6)).astype(np.float32)
    s = NativeSolver(plant, N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1))
    s.mpc_begin(x0)
    s.mpc_step(advance=False, plan=True, ref_window=np.zeros((N, 6), np.float32), hyp_world=hyp)
    got = s.read("f_ext").reshape(B, 6)
    pl = fk_placements(plant, x0[:nq])
    want = np.stack([MPC_GATO.transform_force_to_gato_frame(None, x0[:nq], h, placements=pl) for h in hyp])
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * np.abs(want).max())
    with pytest.raises(Exception):
        NativeSolver(plant, N, B, dt=0.01).mpc_step(advance=True, plan=False, plant_steps=1)   # no session begun on this handle


def _x_start():
    return np.concatenate([INDY7_START_CONFIGS["ready"], np.zeros(6)])


@pytest.mark.gpu
def test_closed_loop_fig8_single_trajectory():
    """run_mpc_fig8 (mpc_controller.py:136-277), B = 1, one SQP iteration per step, a fixed 2 ms per solve: the arm converges onto the
    figure-8 and follows it; the run is reproducible bit for bit"""
    from gato_amd.bsqp.mpc_controller import MPC_GATO
    fig8 = figure8(0.01, **FIG8_DEFAULT_PARAMS)
    runs = []
    for _ in range(2):
        mpc = MPC_GATO(None, None, N=32, dt=0.01, batch_size=1, plant_type="indy7", track_full_stats=True)
        _, st = mpc.run_mpc_fig8(_x_start(), fig8, sim_dt=0.001, sim_time=1.0, solve_time_override=0.002, verbose=False)
        runs.append(st)
    a, b = runs
    assert len(a["timestamps"]) >= 400 and np.all(np.isfinite(a["joint_positions"]))
    for k in ("goal_distances", "joint_positions", "joint_velocities", "ee_actual"):
        np.testing.assert_array_equal(a[k], b[k])
    d = a["goal_distances"]
    assert d[-100:].mean() < 0.5 * d[:20].mean() and d[-100:].max() < 0.05, (d[:20].mean(), d[-100:].mean(), d[-100:].max())
    assert np.all(a["solve_times"] > 0) and np.all(a["sqp_iters"] == DEFAULT_SOLVER_PARAMS["max_sqp_iters"])
    assert set(a) == {"timestamps", "solve_times", "goal_distances", "ee_actual", "joint_positions", "joint_velocities", "sqp_iters"}


@pytest.mark.gpu
def test_closed_loop_with_force_hypotheses():
    """A constant 15 N disturbance on the last link, B = 16 force hypotheses (ForceEstimator + transform + device-side selection):
    the batched controller runs the whole hypothesis machinery every step, tracks at least as well as the single-hypothesis controller
    that knows nothing about the force (within 10 %: over 1.2 s the distance is dominated by the approach), and its estimate moves"""
    from gato_amd.bsqp.mpc_controller import MPC_GATO
    fig8 = figure8(0.01, **FIG8_DEFAULT_PARAMS)
    f = np.array([0.0, 0.0, -15.0, 0.0, 0.0, 0.0])
    res = {}
    for B in (1, 16):
        np.random.seed(0)
        mpc = MPC_GATO(None, None, N=32, dt=0.01, batch_size=B, constant_f_ext=f, plant_type="indy7")
        _, st = mpc.run_mpc_fig8(_x_start(), fig8, sim_dt=0.001, sim_time=1.2, solve_time_override=0.002, verbose=False)
        res[B] = (st, mpc)
    d1, d16 = res[1][0]["goal_distances"], res[16][0]["goal_distances"]
    assert np.all(np.isfinite(d16)) and d16[-200:].mean() < 1.1 * d1[-200:].mean(), (d1[-200:].mean(), d16[-200:].mean())
    est = res[16][1].force_estimator
    assert len(est.error_history) == len(d16) and np.all(np.isfinite(est.estimate)) and 2.0 <= est.radius <= 20.0   # updated once per step
    assert res[1][1].force_estimator is None


@pytest.mark.gpu
def test_closed_loop_goals():
    """run_mpc_goals (mpc_controller.py:361-599): two reachable goals, statistics keys of the reference"""
    from gato_amd.bsqp.config import PICKPLACE_SOLVER_PARAMS
    from gato_amd.bsqp.mpc_controller import MPC_GATO
    mpc = MPC_GATO(None, None, N=16, dt=0.03, batch_size=1, plant_type="indy7", solver_params=PICKPLACE_SOLVER_PARAMS, track_full_stats=True)
    e0 = mpc.solver.ee_pos(INDY7_START_CONFIGS["ready"])
    goals = [e0 + np.array([0.10, 0.05, -0.10]), e0 + np.array([-0.05, 0.10, 0.05])]
    _, st = mpc.run_mpc_goals(_x_start(), goals, sim_dt=0.001, goal_timeout=4.0, solve_time_override=0.004, verbose=False)
    assert st["goal_outcomes"] == ["reached", "reached"], (st["goal_outcomes"], st["goal_distances"][-5:])
    assert st["time_to_all_reached"] is not None and st["time_to_all_reached"] < 8.0
    assert {"timestamps", "solve_times", "goal_distances", "ee_actual", "joint_positions", "joint_velocities", "best_trajectory_id", "goal_outcomes",
            "goal_reached_times", "time_to_all_reached", "sqp_iters", "pcg_iters"} == set(st)
    assert mpc.pendulum_state is None and not mpc.has_pendulum      # the payload is tests/test_pendulum.py


@pytest.mark.gpu
def test_session_selection_without_a_plant_step_scores_against_the_current_state():
    """mpc_controller.py:196-197 takes x_last = x_curr at the top of EVERY loop iteration.  A session step whose latency rounds to no plant
    step (the measured 0.15 ms step against sim_dt = 1 ms: most steps), and a selection issued before any plant step at all, must score the
    hypotheses from the CURRENT state -- round 3 left x_last stale (never written without a plant step, uninitialised before the first one)."""
    from gato_amd._lib import NativeSolver
    plant, N, B, dt, sim_dt = "indy7", 8, 6, 0.01, 0.001
    p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1)
    rng = np.random.default_rng(9)
    x0 = np.concatenate([INDY7_START_CONFIGS["ready"], np.zeros(6)]).astype(np.float32)
    fext = rng.normal(0, 3.0, (B, 6)).astype(np.float32)
    wrench = np.array([0.0, 0.0, 0.0, 1.0, -2.0, 3.0], np.float32)
    win = np.tile(np.array([0.3, 0.2, 0.7, 0, 0, 0], np.float32), (N, 1))
    ses, man = NativeSolver(plant, N, B, dt=dt, **p), NativeSolver(plant, N, B, dt=dt, **p)
    for s in (ses, man):
        s.set_f_ext_batch(fext)
    ses.mpc_begin(x0)
    # (1) a selecting plan before ANY advance: x_last is the session's start state, u_last the warm start's zero control
    out = ses.mpc_step(advance=False, plan=True, ref_window=win, select=True, select_dt=sim_dt)
    best, err = man.select_best(x0, np.zeros(6, np.float32), x0, sim_dt)
    np.testing.assert_array_equal(out["errors"], err)
    assert out["best"] == best and np.all(np.isfinite(out["errors"]))
    # (2) move the plant, then an ADVANCE with zero plant steps + a selection: x_last must be the state the plant is in NOW
    a = ses.mpc_step(advance=True, plan=False, plant_steps=12, sim_dt=sim_dt, steps_per_knot=dt / sim_dt, plant_wrench=wrench)
    x1 = a["x"].copy()
    assert np.abs(x1 - x0).max() > 1e-4 and a["plant_us"] > 0
    best_row = ses.mpc_best()
    out = ses.mpc_step(advance=True, plan=True, plant_steps=0, sim_dt=sim_dt, steps_per_knot=dt / sim_dt, plant_wrench=wrench, ref_window=win, select=True,
                       select_dt=sim_dt)
    np.testing.assert_array_equal(out["x"], x1)
    best, err = man.select_best(x1, best_row[12:18], x1, sim_dt)
    np.testing.assert_array_equal(out["errors"], err)
    assert out["best"] == best


@pytest.mark.gpu
def test_mpc_step_refuses_a_struct_of_another_size():
    """GatoMpcStep carries sizeof(GatoMpcStep) as its client was compiled (advisor, round 4: the struct gained a field and the library wrote it
    past an older client's struct): any other size is refused before anything is read or written."""
    import ctypes as C
    from gato_amd._lib import GatoError, NativeSolver
    s = NativeSolver("indy7", 8, 2, dt=0.01, max_sqp_iters=1)
    s.mpc_begin(np.zeros(12, np.float32))
    io = s.L._MPC()
    io.phases = 2
    for bad in (0, C.sizeof(io) - 8):
        io.struct_size = bad
        io.best = -7
        assert s.L.gato_mpc_step(s.h, C.byref(io)) != 0 and b"struct_size" in s.L.gato_last_error()
        assert io.best == -7
    out = s.mpc_step(advance=False, plan=True, ref_window=np.zeros(48, np.float32))     # the binding's own size is accepted
    assert np.all(np.isfinite(out["x"]))
