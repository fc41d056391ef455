import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
plant, N, B = "indy7", 8, 1
p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1)
pr = fig8_problem(plant, N, B, f_ext_std=1.0)
d = NativeSolver(plant, N, B, dt=0.01, **p)
d.set_f_ext_batch(pr["f_ext"])
for st in ("kkt", "schur", "direct"):
    d.stage(st, pr["xu"], 0.01, pr["x_s"], pr["ref"])
nx = d.nx
lam = d.read("lambda").reshape(N + 2, nx)
S = d.read("S").reshape(N, nx, 3 * nx); gam = d.read("gamma").reshape(N + 2, nx)
P = d.read("Pinv").reshape(N, nx, 3 * nx)
print("lam nan rows", np.isnan(lam).any(axis=1))
print("Dinv0 dev", P[0][:, nx:2*nx][0][:4], "expected", np.linalg.inv(S[0][:, nx:2*nx].astype(np.float64))[0][:4])
print("Dinv nan per block", [bool(np.isnan(P[k][:, nx:2*nx]).any()) for k in range(N)])
D1 = S[1][:, nx:2*nx] - S[1][:, :nx] @ np.linalg.inv(S[0][:, nx:2*nx].astype(np.float64)) @ S[1][:, :nx].T
print("Dinv1 dev", P[1][:, nx:2*nx][0][:4], "expected", np.linalg.inv(D1)[0][:4])
print(lam[1][:4], lam[N][:4])
