import sys, numpy as np
sys.path.insert(0, "/root/repo")
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.workloads import hparam_problem
plant, N, B = "iiwa14", 64, 512
pr = hparam_problem(plant, N, B, shard=0)
for cap in (None, 1):
    p = dict(pr["params"], max_sqp_iters=10)
    if cap: p["max_pcg_iters"] = cap
    s = NativeSolver(plant, N, B, dt=pr["dt"], **p)
    s.set_rho_penalty_batch(pr["rho"], True)
    s.set_profiling(True)
    for rep in range(3):
        s.reset_dual(); s.reset_rho()
        r = s.solve(pr["xu"], pr["dt"], pr["x_s"], pr["ref"])
    st = s.stage_times_us()
    it = np.asarray(r["pcg_iters_all"])
    print("cap", cap, "stage us per solve", {k: round(v, 1) for k, v in st.items()}, "pcg iters per launch: mean %.1f max %s" % (it.mean(), it.max(axis=1).tolist()))
