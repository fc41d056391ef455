#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, numpy as np
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
from gato_amd._lib import NativeSolver
KEYS = ("XU", "final_merit", "pcg_iters_all", "ls_step_size", "ls_min_merit", "kkt_converged")
def run(n, N, B, fstd=0.0):
    os.environ["GATO_PCG_DUAL"] = str(n)
    pr = fig8_problem("indy7", N, B, f_ext_std=fstd)
    s = NativeSolver("indy7", N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10))
    s.set_f_ext_batch(pr["f_ext"])
    r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
    r2 = s.solve(r["XU"], 0.01, pr["x_s"], pr["ref"])
    return [r[k] for k in KEYS] + [r2[k] for k in KEYS] + [s.read("lambda"), s.read("rho")]
for (N, B, fstd) in ((32, 1024, 0.0), (32, 700, 2.0), (16, 600, 0.0), (32, 300, 0.0)):
    ref = run(0, N, B, fstd)
    for n in (1, 64, 128, 10000):
        out = run(n, N, B, fstd)
        ok = all(np.asarray(a).tobytes() == np.asarray(b).tobytes() for a, b in zip(ref, out))
        print(N, B, fstd, "n_pair", n, "SAME" if ok else "DIFF", flush=True)
PY
bench() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['stage_us_per_solve'], d['solution_ok'])"; }
for rep in 1 2; do
for n in 0 16 32 64 96 128 192 256; do
  export GATO_PCG_DUAL=$n
  echo "== n_pair $n"; bench --steps 100 --warmup 5
done; done
