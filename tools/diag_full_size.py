import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_full_size_oracle_gpu as T
for case, over in [("C5", dict(max_sqp_iters=1, pcg_tol=1e-9, max_pcg_iters=1000)), ("C3", dict(max_sqp_iters=1)), ("C5", dict(max_sqp_iters=1))]:
    plant, N, B, dt, pr, nat, orc = T._setup(case, **over)
    rg = nat.solve(pr["xu"], dt, pr["x_s"], pr["ref"]); ro = orc.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
    mg = nat.read("merit").reshape(B, 8)
    same = rg["ls_step_size"][0] == ro["ls_step_size"][0]
    e = np.abs(rg["XU"].astype(np.float64) - ro["XU"]).max(axis=1) / np.maximum(1.0, np.abs(ro["XU"]).max(axis=1))
    print(case, over, "same", same.sum(), "of", B)
    lam_g = nat.read("lambda").reshape(B, -1); lam_o = orc.buf("lambda").reshape(B, -1)
    le = np.abs(lam_g - lam_o).max(axis=1) / np.maximum(1e-30, np.abs(lam_o).max(axis=1))
    for b in np.nonzero(~same)[0][:12]:
        print(" row", b, "rho", pr.get("rho", np.full(B, 0.01))[b], "pcg", rg["pcg_iters"][0][b], ro["pcg_iters"][0][b], "steps", rg["ls_step_size"][0][b], ro["ls_step_size"][0][b], "lam relerr %.2e" % le[b])
        print("   hip merits", np.array2string(mg[b], precision=6), "\n   orc merits", np.array2string(ro["ls_merits"][0, b], precision=6), "before", ro["ls_merit_before"][0, b])
    fl = (ro["pcg_iters"][0] < 1000) & (rg["pcg_iters"][0] < 1000)
    print(" rows both below cap", fl.sum(), " xu err of same&fl: max %.2e p99 %.2e med %.2e" % (e[same & fl].max(), np.quantile(e[same & fl], .99), np.median(e[same & fl])))
    print(" lam relerr of fl rows: max %.2e med %.2e ; pcg diff hist" % (le[fl].max(), np.median(le[fl])), np.bincount(np.minimum(np.abs(rg["pcg_iters"][0].astype(int) - ro["pcg_iters"][0].astype(int)), 5)))
