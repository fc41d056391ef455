"""Per-solve device time of the first 120 C2 solves of a fresh handle (hipEvents between the solves): how long the device takes to settle -- the reason a
20-solve bench line behind three warm-up solves reads ~2 % below a 200-solve one.  Runs on the MI355X box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
pr = fig8_problem("indy7", 32, 1024)
s = NativeSolver("indy7", 32, 1024, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10))
dev = torch.device("cuda", 0)
xu0 = torch.from_numpy(pr["xu"]).to(dev); xs = torch.from_numpy(pr["x_s"]).to(dev); ref = torch.from_numpy(pr["ref"]).to(dev)
xu = xu0.clone(); st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
n = 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    s.reset_async(True, True, st); xu.copy_(xu0); s.solve_device(xu.data_ptr(), 0.01, xs.data_ptr(), ref.data_ptr(), st)
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("per-solve ms, solves 1..120 from a cold handle:", " ".join("%.3f" % x for x in t[:12]), "...", " ".join("%.3f" % x for x in t[20:26]), "...", " ".join("%.3f" % x for x in t[-6:]))
print("mean of solves 4-23: %.4f  24-43: %.4f  100-119: %.4f" % (np.mean(t[3:23]), np.mean(t[23:43]), np.mean(t[100:120])))
