"""The swinging payload of the MPC plant (MPC_GATO's pendulum_config, python/bsqp/mpc_controller.py:44-60, 340-360, 472-478).

pinocchio is not in this image, so the checker is tests/pendulum_ref.py: the arm + spherical joint + bob as ONE tree in float64, inverse
dynamics and a dense solve.  It is itself checked here -- against the oracle's forward dynamics of the arm alone and by conservation of the
total energy of the undamped system -- and then the library's plant (which eliminates the joint articulated-body fashion, in fp32 and in the
float64 build) is compared with it step for step."""
import numpy as np
import pytest

import pendulum_ref as P

NQ = {"indy7": 6, "iiwa14": 7}
PAYLOAD = (15.0, 0.3, 0.001)     # mass, length, the bob's own inertia (mpc_controller.py:342-343, 354)


def _state(plant, seed):
    rng = np.random.default_rng(seed)
    nq = NQ[plant]
    return dict(q=rng.uniform(-1, 1, nq), qd=rng.uniform(-0.4, 0.4, nq), u=rng.uniform(-4, 4, nq), fe=rng.uniform(-3, 3, 6),
                quat=P.quat_exp(rng.uniform(-0.5, 0.5, 3)), w=rng.uniform(-0.5, 0.5, 3))


@pytest.mark.parametrize("plant", ["indy7", "iiwa14"])
def test_reference_tree_equals_the_oracle_on_the_arm_alone(plant):
    from oracle import oracle as O
    for seed in range(4):
        s = _state(plant, seed)
        a = P.forward_dynamics(plant, s["q"], s["qd"], s["u"], f_ext=s["fe"])
        f32 = [np.asarray(s[k], np.float32) for k in ("q", "qd", "u", "fe")]
        b = np.asarray(O.fd(plant, *f32), np.float64)
        assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(a).max())


@pytest.mark.parametrize("plant", ["indy7", "iiwa14"])
def test_reference_tree_conserves_energy(plant):
    """no damping, no torque, no wrench: kinetic + potential energy of arm and bob stays put over 150 RK4 steps of 1 ms while everything moves"""
    s = _state(plant, 7)
    q, qd, quat, w = s["q"], s["qd"], s["quat"], s["w"]
    e0 = P.total_energy(plant, q, qd, quat, w, PAYLOAD)
    q0, quat0 = q.copy(), quat.copy()
    for _ in range(150):
        q, qd, quat, w = P.rk4_step(plant, q, qd, quat, w, np.zeros(NQ[plant]), 1e-3, PAYLOAD, 0.0)
    e1 = P.total_energy(plant, q, qd, quat, w, PAYLOAD)
    assert np.abs(q - q0).max() > 0.05 and np.abs(quat - quat0).max() > 0.02
    assert abs(e1 - e0) < 1e-6 * abs(e0)
    # and with damping the energy only goes down
    q, qd, quat, w = s["q"], s["qd"], s["quat"], s["w"]
    for _ in range(150):
        q, qd, quat, w = P.rk4_step(plant, q, qd, quat, w, np.zeros(NQ[plant]), 1e-3, PAYLOAD, 0.4)
    assert P.total_energy(plant, q, qd, quat, w, PAYLOAD) < e0


def test_pendulum_config_samplers():
    from gato_amd.bsqp.common import sample_axis_angle, sample_pendulum_params
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = sample_axis_angle((0.1, 0.6), rng)
        assert a.shape == (3,) and 0.1 <= np.linalg.norm(a) <= 0.6 + 1e-9
    c = sample_pendulum_params(rng=rng)
    assert c["mass"] == 15.0 and 0.3 <= c["length"] <= 0.7 and 0.1 <= c["damping"] <= 0.6 and np.linalg.norm(c["initial_angle"]) <= 0.6 + 1e-9


def _pend11(s, damping):
    return np.concatenate([s["quat"], s["w"], [PAYLOAD[0], PAYLOAD[1], damping, PAYLOAD[2]]])


@pytest.mark.gpu
@pytest.mark.parametrize("plant", ["indy7", "iiwa14"])
@pytest.mark.parametrize("f64", [False, True])
def test_payload_plant_equals_the_reference_tree(plant, f64):
    """40 RK4 steps of 1 ms with a new control every step, a wrench on the last link and joint damping: arm state, quaternion and angular
    velocity against the float64 tree.  The float64 build agrees to rounding; fp32 to what 40 steps of fp32 forward dynamics allow."""
    from gato_amd._lib import NativeSolver
    nat = NativeSolver(plant, 8, 1, f64=f64, dt=0.01)
    nq = NQ[plant]
    s = _state(plant, 11)
    rng = np.random.default_rng(5)
    useq = rng.uniform(-4, 4, (40, nq))
    x, pend = nat.plant_payload_rk4(np.concatenate([s["q"], s["qd"]]), _pend11(s, 0.4), useq, s["fe"], 1e-3)
    q, qd, quat, w = s["q"], s["qd"], s["quat"], s["w"]
    for i in range(40):
        q, qd, quat, w = P.rk4_step(plant, q, qd, quat, w, useq[i], 1e-3, PAYLOAD, 0.4, f_ext=s["fe"])
    tol = 1e-9 if f64 else 2e-4
    np.testing.assert_allclose(np.asarray(x, np.float64), np.concatenate([q, qd]), atol=tol * max(1.0, np.abs(qd).max()))
    np.testing.assert_allclose(np.asarray(pend[:4], np.float64), quat, atol=tol)
    np.testing.assert_allclose(np.asarray(pend[4:7], np.float64), w, atol=tol * max(1.0, np.abs(w).max()))
    assert np.abs(quat - s["quat"]).max() > 1e-3                      # it swings
    np.testing.assert_array_equal(pend[7:], np.asarray(_pend11(s, 0.4)[7:], pend.dtype))


@pytest.mark.gpu
def test_a_light_payload_leaves_the_arm_alone():
    """mass -> 0: the arm moves as without a payload (the two plant kernels' arm paths agree)"""
    from gato_amd._lib import NativeSolver
    nat = NativeSolver("indy7", 8, 1, f64=True, dt=0.01)
    s = _state("indy7", 3)
    useq = np.tile(s["u"], (20, 1))
    x0 = np.concatenate([s["q"], s["qd"]])
    bare = nat.plant_rk4(x0, useq, s["fe"], 1e-3)
    p = _pend11(s, 0.0); p[7] = 1e-9; p[10] = 1e-12
    with_p, _ = nat.plant_payload_rk4(x0, p, useq, s["fe"], 1e-3)
    np.testing.assert_allclose(with_p, bare, atol=1e-6)


@pytest.mark.gpu
def test_session_plant_carries_the_payload():
    """gato_mpc_set_payload: the session's ADVANCE phase is gato_plant_payload_rk4 with the controls of the best trajectory, bit for bit;
    taking the payload away restores the bare arm"""
    from gato_amd._lib import NativeSolver
    N, nq = 8, 6
    nat = NativeSolver("indy7", N, 2, dt=0.01, max_sqp_iters=2)
    s = _state("indy7", 2)
    x0 = np.concatenate([s["q"], np.zeros(nq)]).astype(np.float32)
    ref = np.zeros((N, 6), np.float32); ref[:, :3] = (0.3, 0.3, 0.5)
    p11 = _pend11(s, 0.4).astype(np.float32)
    nat.mpc_begin(x0)
    nat.mpc_set_payload(p11)
    nat.mpc_step(advance=False, plan=True, ref_window=ref)
    best = nat.mpc_best().reshape(-1)
    out = nat.mpc_step(advance=True, plan=False, plant_steps=25, sim_dt=1e-3, steps_per_knot=10.0, plant_wrench=s["fe"])
    useq = np.stack([best[min(int(i / 10.0), N - 1) * 18 + 12: min(int(i / 10.0), N - 1) * 18 + 18] for i in range(25)])
    x, pend = nat.plant_payload_rk4(x0, p11, useq, s["fe"], 1e-3)
    np.testing.assert_array_equal(out["x"], x)
    np.testing.assert_array_equal(nat.mpc_payload(), pend[:7])
    nat.mpc_set_payload(None)
    out2 = nat.mpc_step(advance=True, plan=False, plant_steps=5, sim_dt=1e-3, steps_per_knot=10.0, plant_wrench=s["fe"])
    np.testing.assert_array_equal(out2["x"], nat.plant_rk4(x, useq[:5] * 0 + best[12:18], s["fe"], 1e-3))
    with pytest.raises(Exception):
        nat.mpc_payload()


@pytest.mark.gpu
def test_mpc_goals_with_a_swinging_payload():
    """MPC_GATO(pendulum_config=...): the loop runs on the arm's state only (mpc_controller.py:505-507) while the plant swings the bob"""
    from gato_amd.bsqp.config import INDY7_START_CONFIGS
    from gato_amd.bsqp.mpc_controller import MPC_GATO
    cfg = {"mass": 2.0, "length": 0.3, "damping": 0.4, "initial_angle": np.array([0.3, 0.0, 0.0])}
    mpc = MPC_GATO(N=8, dt=0.03125, batch_size=1, plant_type="indy7", pendulum_config=cfg, solver_params={"max_sqp_iters": 2})
    assert mpc.has_pendulum and (mpc.nq, mpc.nv, mpc.nq_robot) == (10, 9, 6)
    x0 = np.concatenate([INDY7_START_CONFIGS["ready"], np.zeros(6)])
    goal = np.asarray(mpc.solver.ee_pos(x0[:6]), np.float64).reshape(3) + np.array([0.05, 0.0, 0.03])
    _, st = mpc.run_mpc_goals(x0, [goal], goal_timeout=0.3, solve_time_override=0.004, verbose=False)
    assert len(st["timestamps"]) > 20 and st["joint_positions"].shape[1] == 6 and np.isfinite(st["joint_positions"]).all()
    ps = mpc.pendulum_state
    assert ps.shape == (7,) and abs(np.linalg.norm(ps[:4]) - 1) < 1e-5 and np.abs(ps[4:]).max() > 1e-3
    # the same run without the payload ends elsewhere: the bob does pull on the arm
    bare = MPC_GATO(N=8, dt=0.03125, batch_size=1, plant_type="indy7", solver_params={"max_sqp_iters": 2})
    _, st0 = bare.run_mpc_goals(x0, [goal], goal_timeout=0.3, solve_time_override=0.004, verbose=False)
    assert np.abs(st0["joint_positions"][-1] - st["joint_positions"][-1]).max() > 1e-4


@pytest.mark.gpu
def test_compiled_binding_and_fig8_loop_carry_the_payload():
    """the pybind11 class's plant_payload_rk4 is the ctypes path's, bit for bit; the figure-8 loop runs with a payload as the goal loop does"""
    from gato_amd._lib import NativeSolver
    from gato_amd.bsqp.bsqpN8_indy7 import BSQP_1_float
    from gato_amd.bsqp.common import figure8
    from gato_amd.bsqp.config import INDY7_START_CONFIGS
    from gato_amd.bsqp.mpc_controller import MPC_GATO
    s = _state("indy7", 4)
    x0 = np.concatenate([s["q"], s["qd"]]).astype(np.float32)
    useq = np.tile(s["u"], (10, 1)).astype(np.float32)
    p11 = _pend11(s, 0.4).astype(np.float32)
    a_x, a_p = NativeSolver("indy7", 8, 1, dt=0.01).plant_payload_rk4(x0, p11, useq, s["fe"], 1e-3)
    b_x, b_p = BSQP_1_float().plant_payload_rk4(x0, p11, useq, np.asarray(s["fe"], np.float32), 1e-3)
    np.testing.assert_array_equal(np.asarray(b_x), a_x)
    np.testing.assert_array_equal(np.asarray(b_p), a_p)
    mpc = MPC_GATO(N=8, dt=0.03125, batch_size=1, plant_type="indy7", pendulum_config={"mass": 1.0, "length": 0.2}, solver_params={"max_sqp_iters": 2})
    x_start = np.concatenate([INDY7_START_CONFIGS["ready"], np.zeros(6)])
    _, st = mpc.run_mpc_fig8(x_start, figure8(0.03125), sim_time=0.2, solve_time_override=0.004, verbose=False)
    assert len(st["timestamps"]) > 20 and np.isfinite(st["joint_positions"]).all() and abs(np.linalg.norm(mpc.pendulum_state[:4]) - 1) < 1e-5
