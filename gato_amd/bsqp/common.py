"""Workload helpers with the reference's names and semantics (python/bsqp/common.py:10-44, 93-99), numpy only."""
import numpy as np


def figure8(dt, A_x=0.4, A_z=0.4, offset=(0.0, 0.5, 0.6), period=6, cycles=5, theta=np.pi / 4):
    """Figure-8 end-effector reference: flat float64 array of [x, y, z, 0, 0, 0] per timestep, `cycles` repetitions.

    Same sampling as the reference (python/bsqp/common.py:10-44): int(period/dt) points of t in linspace(0, 2pi), the planar
    curve (A_x sin t, 0, A_z sin(2t)/2 + A_z/2) + offset rotated about z by theta.
    """
    t = np.linspace(0, 2 * np.pi, int(period / dt))
    unrot = np.stack([offset[0] + A_x * np.sin(t), np.full_like(t, offset[1]), offset[2] + A_z * np.sin(2 * t) / 2 + A_z / 2])
    R = np.array([[np.cos(theta), -np.sin(theta), 0.0], [np.sin(theta), np.cos(theta), 0.0], [0.0, 0.0, 1.0]])
    rot = np.stack([R[i, 0] * unrot[0] + R[i, 1] * unrot[1] + R[i, 2] * unrot[2] for i in range(3)])   # the float sequence of the reference's loop
    pts = np.zeros((t.size, 6))
    pts[:, :3] = rot.T
    return np.tile(pts.reshape(-1), int(cycles))


def initialize_warm_start(x_start, N, nx, nu):
    """x_start repeated over the horizon, zero controls (python/bsqp/common.py:93-99)."""
    XU = np.zeros(N * (nx + nu) - nu)
    for i in range(N):
        XU[i * (nx + nu): i * (nx + nu) + nx] = x_start
    return XU


def sample_axis_angle(mag_range=(0.0, 0.6), rng=None):
    """Axis-angle vector of the payload's initial rotation: a uniformly random direction times a magnitude drawn uniformly from
    mag_range in radians (python/bsqp/common.py:121-136).  `rng` (a numpy Generator, extension) makes the draw reproducible."""
    rng = np.random.default_rng() if rng is None else rng
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction) + 1e-12
    return direction * rng.uniform(*mag_range)


def sample_pendulum_params(length_range=(0.3, 0.7), damping_range=(0.1, 0.6), angle_range=(0.0, 0.6), mass=15.0, rng=None):
    """A random `pendulum_config` for MPC_GATO sweeps (python/bsqp/common.py:139-158): fixed mass, length in metres, damping in N m s / rad."""
    rng = np.random.default_rng() if rng is None else rng
    return {"mass": mass, "length": float(rng.uniform(*length_range)), "damping": float(rng.uniform(*damping_range)),
            "initial_angle": sample_axis_angle(angle_range, rng)}
