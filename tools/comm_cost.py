import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
dev = torch.device("cuda", 0)
N, B = 32, 1024
pr = fig8_problem("indy7", N, B)
p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10)
xu0 = torch.from_numpy(pr["xu"]).to(dev); xs = torch.from_numpy(pr["x_s"]).to(dev); ref = torch.from_numpy(pr["ref"]).to(dev)
def run(s, reps=100):
    x = xu0.clone()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        s.reset_async(True, True, st); x.copy_(xu0); s.solve_device(x.data_ptr(), 0.01, xs.data_ptr(), ref.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        s.reset_async(True, True, st); x.copy_(xu0); s.solve_device(x.data_ptr(), 0.01, xs.data_ptr(), ref.data_ptr(), st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
a = NativeSolver("indy7", N, B, dt=0.01, **p)
print("no communicator: %.1f us per solve" % run(a))
b = NativeSolver("indy7", N, B, dt=0.01, **p)
b.comm_init(NativeSolver.comm_unique_id(), 1, 0)
print("one-rank communicator (10 ncclAllReduce of 4 bytes per solve): %.1f us per solve" % run(b))
