import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
N, B, dt = 32, int(os.environ.get("B", 1024)), 0.01
pr = fig8_problem("indy7", N, B)
p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10)
dev = torch.device("cuda", 0)
xu0 = torch.from_numpy(pr["xu"]).to(dev); xs = torch.from_numpy(pr["x_s"]).to(dev); ref = torch.from_numpy(pr["ref"]).to(dev)
for persist in ("0", "1"):
    os.environ["GATO_PERSIST"] = persist
    s = NativeSolver("indy7", N, B, dt=dt, **p)
    xu = xu0.clone(); st = torch.cuda.current_stream().cuda_stream
    def step():
        s.reset_async(True, True, st); xu.copy_(xu0); s.solve_device(xu.data_ptr(), dt, xs.data_ptr(), ref.data_ptr(), st)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    r = s.stats()
    print("persist", persist, "B", B, "ms/solve %.4f" % (t * 1e3), "traj-iter/s %.3e" % (B * 10 / t), "merit sum", float(r["final_merit"].sum()), "pcg iters", int(r["pcg_iters_all"].sum()))
