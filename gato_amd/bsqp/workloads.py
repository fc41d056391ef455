"""Synthetic benchmark / parity inputs: fig-8 tracking windows with distinct phases per trajectory."""
import numpy as np

from .common import figure8, initialize_warm_start
from .config import FIG8_DEFAULT_PARAMS, INDY7_START_CONFIGS

NQ = {"indy7": 6, "iiwa14": 7}


def fig8_problem(plant, N, B, seed=0, dt=0.01, f_ext_std=0.0, batch_offset=0):
    """Returns dict(xu[B,TRAJ], x_s[B,nx], ref[B,6N], f_ext[B,6]) float32.

    ref_b = fig8[6 o_b : 6 (o_b + N)], o_b = (37 b) mod (600 - N); x_s,b = [q0 + U(-0.1,0.1), 0] with q0 = indy7 'ready' or, for
    iiwa14, U(-0.3,0.3) about the zero configuration; warm start = x_s repeated, u = 0.  `batch_offset` shifts b so that
    rank r of a sharded run generates rows [r*B, (r+1)*B) of the global problem.
    """
    nq = NQ[plant]
    nx, nu = 2 * nq, nq
    fig8 = figure8(dt, **FIG8_DEFAULT_PARAMS).reshape(-1, 6)
    npts = int(FIG8_DEFAULT_PARAMS["period"] / dt)
    xu = np.zeros((B, (nx + nu) * N - nu), np.float32)
    x_s = np.zeros((B, nx), np.float32)
    ref = np.zeros((B, 6 * N), np.float32)
    f_ext = np.zeros((B, 6), np.float32)
    for i in range(B):
        b = i + batch_offset
        rng = np.random.default_rng([seed, b])
        o = (37 * b) % (npts - N)
        ref[i] = fig8[o:o + N].reshape(-1)
        if plant == "indy7":
            q0 = INDY7_START_CONFIGS["ready"] + rng.uniform(-0.1, 0.1, nq)
        else:
            q0 = rng.uniform(-0.3, 0.3, nq)
        x_s[i, :nq] = q0
        xu[i] = initialize_warm_start(x_s[i], N, nx, nu)
        if f_ext_std > 0:
            f_ext[i] = rng.normal(0.0, f_ext_std, 6)
    return dict(xu=xu, x_s=x_s, ref=ref, f_ext=f_ext)


# the first 8 tuples of the hyper-parameter notebook's grid Q x QD x U x Ncost (configuration C5)
HPARAM_COST_GRID = [dict(q_cost=q, qd_cost=qd, u_cost=u, N_cost=nc) for q in (10.0, 1.0) for qd in (1e-1, 1e-3, 1e-5) for u in (1e-6, 1e-7)
                    for nc in (100.0, 10.0)]


def hparam_problem(plant, N, B, shard=0, seed=0):
    """Configuration C5: the hyper-parameter sweep.  Shard g (one GPU) uses cost tuple g of HPARAM_COST_GRID for all its
    trajectories; trajectory j of the shard has rho_j = 10^(-8 + 9 (j mod 512 + 1) / 513); one random goal per trajectory,
    U([-.8,.8]^2 x [.2,.8]), repeated over the horizon; x_s = 0; mu = 1, pcg_tol = 1e-3, dt = 0.05.

    Returns dict(xu, x_s, ref, f_ext, rho[B], params(dict of solver parameters incl. the shard's cost tuple), dt)."""
    from .config import DEFAULT_SOLVER_PARAMS
    nq = NQ[plant]
    nx, nu = 2 * nq, nq
    xu = np.zeros((B, (nx + nu) * N - nu), np.float32)
    x_s = np.zeros((B, nx), np.float32)
    ref = np.zeros((B, N, 6), np.float32)
    rho = np.zeros(B, np.float32)
    for i in range(B):
        rng = np.random.default_rng([seed, shard, i])
        goal = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0.2, 0.8)])
        ref[i, :, :3] = goal
        xu[i] = initialize_warm_start(x_s[i], N, nx, nu)
        rho[i] = 10.0 ** (-8.0 + 9.0 * ((i % 512) + 1) / 513.0)
    params = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10, mu=1.0, pcg_tol=1e-3)
    params.update(HPARAM_COST_GRID[shard % len(HPARAM_COST_GRID)])
    return dict(xu=xu, x_s=x_s, ref=ref.reshape(B, 6 * N), f_ext=np.zeros((B, 6), np.float32), rho=rho, params=params, dt=0.05)
