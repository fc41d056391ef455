import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["GATO_PERSIST"] = "1"; os.environ["GATO_PAIR_TRACE"] = "1"
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
N, B, dt = 32, int(os.environ.get("B", 1024)), 0.01
pr = fig8_problem("indy7", N, B)
s = NativeSolver("indy7", N, B, dt=dt, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10))
for _ in range(3):
    s.reset_dual(); s.reset_rho()
    order = s.read("order").astype(int)          # the pairing the NEXT solve uses
    r = s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
tr = s.read("pair_trace").reshape(-1, 32)
it = r["pcg_iters_all"]                        # [10, B]
nW = len(tr)
b0 = order[:nW]; b1 = np.array([order[B - 1 - j] if j < B // 2 else -1 for j in range(nW)])
pm = np.where(b1 >= 0, np.maximum(it[:, b0], it[:, np.maximum(b1, 0)]), it[:, b0])      # lock-step iterations per workgroup and SQP iteration
start, end = tr[:, 0], tr[:, 20]
print("B", B, "workgroups", nW, "sqp_time_us", r.get("sqp_time_us"))
print("start: max %.1f us; end: min %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (start.max(), end.min(), np.median(end), np.quantile(end, .9), np.quantile(end, .99), end.max()))
dur = end - start
tot = pm.sum(axis=0)
A = np.stack([tot, np.ones_like(tot)], 1).astype(float); coef = np.linalg.lstsq(A, dur, rcond=None)[0]
print("duration ~ %.3f us x (sum of lock-step PCG iterations) + %.1f us" % (coef[0], coef[1]))
pcg = tr[:, 1:20:2] - np.concatenate([tr[:, :1], tr[:, 2:19:2]], 1)   # kkt + pcg of iteration i
stp = tr[:, 2:21:2] - tr[:, 1:20:2]                                    # exit rule + two steps
for j in np.argsort(-dur)[:6]:
    print(" workgroup %d: trajectories %d (%d its) + %d (%d its), lock-step %d, ends %.1f; assembly+PCG per iteration %s ; steps %s" % (
        j, b0[j], it[:, b0[j]].sum(), b1[j], it[:, b1[j]].sum() if b1[j] >= 0 else 0, tot[j], end[j], np.round(pcg[j]).astype(int), np.round(stp[j]).astype(int)))
print(" lock-step iterations of these per SQP iteration:", [list(pm[:, j]) for j in np.argsort(-dur)[:3]])
for i in (0, 4, 9):
    a = np.stack([pm[i], np.ones_like(pm[i])], 1).astype(float); c = np.linalg.lstsq(a, pcg[:, i], rcond=None)[0]
    print(" it %d: assembly+PCG median %.1f us = %.3f us/iteration x it + %.1f ; steps median %.1f us (min %.1f max %.1f)" % (i, np.median(pcg[:, i]), c[0], c[1], np.median(stp[:, i]), stp[:, i].min(), stp[:, i].max()))
