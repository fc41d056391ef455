"""Static equilibria held by a wrench on the last link -- inputs of the convergence / solve_ratio parity tests (tests/mixed_batch.py).

A trajectory that rests at (q, 0) with u = 0 under the wrench f, tracks ee(q) and carries no joint-limit barrier is a KKT point of
the tracking problem: every cost gradient and every dynamics defect vanishes, so gamma = 0 up to rounding and the reference's PCG takes
0 iterations -- the ONLY convergence rule of the reference's driver (bsqp.cuh:153).  Such rows, next to ordinary fig-8 rows, give a
batch in which a strict subset is converged at entry.

(q, f) solve RNEA(q, 0, 0; f) = 0 in the float64 build of the oracle's dynamics: indy7 (6 joints, 6 wrench components) for a given q,
iiwa14 (7 joints) with joints 2, 4, 6 free as well.  Writes tests/golden/equilibria.npz (data only)."""
import ctypes as C
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_amd.bsqp.config import INDY7_START_CONFIGS  # noqa: E402
from oracle import oracle as O  # noqa: E402


def rnea64(plant, q, f):
    L = O.lib(True)
    nq = O.NQ[plant]
    out, z = np.zeros(nq), np.zeros(nq)
    fp = C.POINTER(C.c_double)
    q, f = np.ascontiguousarray(q, np.float64), np.ascontiguousarray(f, np.float64)
    L.orc_rnea(O.PLANTS[plant], q.ctypes.data_as(fp), z.ctypes.data_as(fp), z.ctypes.data_as(fp), f.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out


def equilibrium(plant, seed):
    nq = O.NQ[plant]
    rng = np.random.default_rng([7, seed])
    if plant == "indy7":
        q0 = INDY7_START_CONFIGS["ready"] + rng.uniform(-0.2, 0.2, nq)
        free = []
    else:
        q0 = rng.uniform(-0.6, 0.6, nq)
        free = [1, 3, 5]
    q0 = q0.astype(np.float32).astype(np.float64)

    def res(z):
        q = q0.copy()
        q[free] = z[:len(free)]
        return rnea64(plant, q, z[len(free):])
    r = least_squares(res, np.concatenate([q0[free], np.zeros(6)]), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    q = q0.copy()
    q[free] = r.x[:len(free)]
    # q is stored in float32 (what the solvers receive); the wrench is re-solved for the ROUNDED q where that is exact (indy7)
    q = q.astype(np.float32).astype(np.float64)
    f = r.x[len(free):]
    if not free:
        f = least_squares(lambda ff: rnea64(plant, q, ff), f, xtol=1e-15, ftol=1e-15, gtol=1e-15).x
    return q.astype(np.float32), f.astype(np.float32), float(np.abs(rnea64(plant, q, f.astype(np.float32))).max())


if __name__ == "__main__":
    out = {}
    for plant in ("indy7", "iiwa14"):
        qs, fs = [], []
        seed = 0
        while len(qs) < 8:
            q, f, resid = equilibrium(plant, seed)
            seed += 1
            if resid > 2e-5 or np.abs(f).max() > 100:
                continue
            qs.append(q); fs.append(f)
            print(plant, seed - 1, "max |tau| at the float32 (q, f): %.2e" % resid, "f", np.round(f, 2))
        out[plant + "_q"] = np.array(qs, np.float32)
        out[plant + "_f"] = np.array(fs, np.float32)
    np.savez(os.path.join(ROOT, "tests", "golden", "equilibria.npz"), **out)
