"""Float64 reference of the MPC plant that carries a swinging payload (test infrastructure, like oracle/).

What the reference simulates with pinocchio when MPC_GATO gets a `pendulum_config` (python/bsqp/mpc_controller.py:44-60, _add_pendulum_to_model
:340-360, the control augmentation :472-478, `rk4` in python/bsqp/common.py:49-91): the arm plus one body on a spherical joint at the last joint
frame.  pinocchio is not in this image, so the system is restated here from the textbook recursions (Featherstone, Rigid Body Dynamics
Algorithms, ch. 5-6) in the UN-ELIMINATED form -- inverse dynamics of the whole tree, the joint-space inertia matrix column by column from it,
one dense solve for [qdd; wd] -- while the library eliminates the spherical joint articulated-body fashion (kernels.hpp payload_dynamics).  The
arm's tables are the ones tools/gen_robot_models.py builds from the URDF parameters.  Checked in tests/test_pendulum.py against the oracle's
arm-only forward dynamics, and by conservation of the total energy of the undamped, unforced system.

Conventions: spatial vectors [angular; linear]; X_k = [E 0; -E r~ E], E = Ez(q_k) E0_k; the spherical joint's configuration is the unit
quaternion (x, y, z, w) of the rotation pendulum -> last-link coordinates and its velocity the relative angular velocity in pendulum
coordinates (pinocchio's JointModelSpherical)."""
import numpy as np

from tools import gen_robot_models as G

GRAV = 9.81
_MODELS = {}


def model(plant):
    if plant not in _MODELS:
        _MODELS[plant] = G.build(G.INDY7 if plant == "indy7" else G.IIWA14)
    return _MODELS[plant]


def skew(c):
    return np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0.0]])


def crm(v):
    """motion cross product matrix: crm(v) m = v x m"""
    o = np.zeros((6, 6))
    o[:3, :3] = skew(v[:3]); o[3:, :3] = skew(v[3:]); o[3:, 3:] = skew(v[:3])
    return o


def crf(v):
    """force cross product matrix: crf(v) f = v x* f"""
    return -crm(v).T


def joint_X(m, k, qk):
    c, s = np.cos(qk), np.sin(qk)
    Ez = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1.0]])
    E = Ez @ m["E0"][k]
    X = np.zeros((6, 6))
    X[:3, :3] = E; X[3:, 3:] = E; X[3:, :3] = -E @ skew(m["r"][k])
    return X, E


def quat_R(qt):
    x, y, z, w = qt
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_exp(t):
    a = np.linalg.norm(t)
    if a < 1e-12:
        return np.array([0.5 * t[0], 0.5 * t[1], 0.5 * t[2], 1.0])
    return np.concatenate([np.sin(a / 2) * t / a, [np.cos(a / 2)]])


def quat_mul(a, b):
    av, bv = a[:3], b[:3]
    return np.concatenate([a[3] * bv + b[3] * av + np.cross(av, bv), [a[3] * b[3] - av @ bv]])


def quat_integrate(qt, w, h):
    o = quat_mul(qt, quat_exp(np.asarray(w) * h))
    return o / np.linalg.norm(o)


def payload_inertia(mass, length, inertia):
    c = np.array([0.0, 0.0, -length])
    C = skew(c)
    I6 = np.zeros((6, 6))
    I6[:3, :3] = inertia * np.eye(3) + mass * (C @ C.T)
    I6[:3, 3:] = mass * C
    I6[3:, :3] = mass * C.T
    I6[3:, 3:] = mass * np.eye(3)
    return I6


def inverse_dynamics(plant, q, qd, qdd, quat=None, w=None, wd=None, payload=None, f_ext=None):
    """generalised forces [tau_arm (nq); tau_pendulum (3)] of the whole tree (the last three only with a payload = (mass, length, inertia))"""
    m = model(plant)
    nq = m["nq"]
    v, a, X = [None] * nq, [None] * nq, [None] * nq
    vp, ap = np.zeros(6), np.array([0, 0, 0, 0, 0, GRAV])     # the base accelerates upwards: gravity
    S = np.array([0, 0, 1.0, 0, 0, 0])
    for k in range(nq):
        X[k], _ = joint_X(m, k, q[k])
        v[k] = X[k] @ vp + S * qd[k]
        a[k] = X[k] @ ap + S * qdd[k] + crm(v[k]) @ (S * qd[k])
        vp, ap = v[k], a[k]
    f = [m["I"][k] @ a[k] + crf(v[k]) @ (m["I"][k] @ v[k]) for k in range(nq)]
    if f_ext is not None:
        f[nq - 1] = f[nq - 1] - np.asarray(f_ext, float)
    tau = np.zeros(nq + (3 if payload is not None else 0))
    if payload is not None:
        E = quat_R(quat).T
        Xp = np.zeros((6, 6)); Xp[:3, :3] = E; Xp[3:, 3:] = E
        Ip = payload_inertia(*payload)
        sw = np.concatenate([w, np.zeros(3)])
        v_p = Xp @ v[nq - 1] + sw
        a_p = Xp @ a[nq - 1] + np.concatenate([wd, np.zeros(3)]) + crm(v_p) @ sw
        f_p = Ip @ a_p + crf(v_p) @ (Ip @ v_p)
        tau[nq:] = f_p[:3]
        f[nq - 1] = f[nq - 1] + Xp.T @ f_p
    for k in range(nq - 1, -1, -1):
        tau[k] = f[k][2]
        if k > 0:
            f[k - 1] = f[k - 1] + X[k].T @ f[k]
    return tau


def mass_matrix(plant, q, quat=None, payload=None):
    nq = model(plant)["nq"]
    n = nq + (3 if payload is not None else 0)
    z = np.zeros(nq)
    kw = dict(quat=quat, payload=payload) if payload is not None else {}

    def idyn(qdd, wd):
        return inverse_dynamics(plant, q, z, qdd, w=np.zeros(3), wd=wd, **kw) if payload is not None else inverse_dynamics(plant, q, z, qdd)
    b = idyn(z, np.zeros(3))
    Mm = np.zeros((n, n))
    for i in range(n):
        e = np.zeros(n); e[i] = 1.0
        Mm[:, i] = idyn(e[:nq], e[nq:]) - b
    return Mm


def forward_dynamics(plant, q, qd, u, quat=None, w=None, taup=None, payload=None, f_ext=None):
    """[qdd; wd] from one dense solve of the whole tree's equations of motion"""
    nq = model(plant)["nq"]
    if payload is None:
        b = inverse_dynamics(plant, q, qd, np.zeros(nq), f_ext=f_ext)
        return np.linalg.solve(mass_matrix(plant, q), np.asarray(u, float) - b)
    b = inverse_dynamics(plant, q, qd, np.zeros(nq), quat=quat, w=w, wd=np.zeros(3), payload=payload, f_ext=f_ext)
    rhs = np.concatenate([np.asarray(u, float), np.asarray(taup, float)]) - b
    return np.linalg.solve(mass_matrix(plant, q, quat=quat, payload=payload), rhs)


def rk4_step(plant, q, qd, quat, w, u, h, payload, damping, f_ext=None):
    """python/bsqp/common.py:49-91 on the arm + payload tree; payload = (mass, length, inertia)"""
    taup = -damping * np.asarray(w, float)

    def acc(q_, quat_, qd_, w_):
        o = forward_dynamics(plant, q_, qd_, u, quat=quat_, w=w_, taup=taup, payload=payload, f_ext=f_ext)
        return o[:len(q_)], o[len(q_):]
    k1q, k1o = qd, w
    k1v, k1w = acc(q, quat, qd, w)
    k2q, k2o = qd + k1v * h / 2, w + k1w * h / 2
    k2v, k2w = acc(q + k1q * h / 2, quat_integrate(quat, k1o, h / 2), k2q, k2o)
    k3q, k3o = qd + k2v * h / 2, w + k2w * h / 2
    k3v, k3w = acc(q + k2q * h / 2, quat_integrate(quat, k2o, h / 2), k3q, k3o)
    k4q, k4o = qd + k3v * h, w + k3w * h
    k4v, k4w = acc(q + k3q * h, quat_integrate(quat, k3o, h), k4q, k4o)
    qd_n = qd + h / 6 * (k1v + 2 * k2v + 2 * k3v + k4v)
    w_n = w + h / 6 * (k1w + 2 * k2w + 2 * k3w + k4w)
    q_n = q + h * (k1q + 2 * k2q + 2 * k3q + k4q) / 6
    quat_n = quat_integrate(quat, (k1o + 2 * k2o + 2 * k3o + k4o) / 6, h)
    return q_n, qd_n, quat_n, w_n


def total_energy(plant, q, qd, quat, w, payload):
    """kinetic + potential energy of arm and payload (potential from the link and bob centres of mass in the world frame)"""
    m = model(plant)
    nq = m["nq"]
    Mm = mass_matrix(plant, q, quat=quat, payload=payload)
    vel = np.concatenate([qd, w])
    ke = 0.5 * vel @ Mm @ vel
    R, p, pe = np.eye(3), np.zeros(3), 0.0
    for k in range(nq):
        _, E = joint_X(m, k, q[k])
        p = p + R @ m["r"][k]
        R = R @ E.T
        mass = m["I"][k][3, 3]
        hc = m["I"][k][:3, 3:]          # m c~
        com = np.array([hc[2, 1], hc[0, 2], hc[1, 0]]) / mass
        pe += mass * GRAV * (p + R @ com)[2]
    bob = p + R @ quat_R(quat) @ np.array([0, 0, -payload[1]])
    pe += payload[0] * GRAV * bob[2]
    return ke + pe
