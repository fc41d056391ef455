"""Per-wavefront cycle accounting of one PCG iteration (throw-away build -DGATO_PCG_TRACE: s_memtime stamps around every phase).
   GATO_HIP_LIB=tools/exp/libgato_hip_trace.so python tools/exp/pcg_phase_trace.py c5|c3 [out]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gato_amd import _lib
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem, hparam_problem

which = sys.argv[1]
if which == "c5":
    plant, N, B = "iiwa14", 64, 512
    pr = hparam_problem(plant, N, B, shard=3); p = dict(pr["params"]); dt = pr["dt"]
    names = {0: "store p", 1: "barrier A", 2: "window + S p + p.Ap", 3: "wave sum + partial", 4: "barrier B", 5: "alpha, x, r, store r", 6: "barrier C",
             7: "window + P^-1 r + r.z", 8: "wave sum + partial", 9: "barrier D", 10: "test, beta, p"}
else:
    plant, N, B = "iiwa14", 128, 256
    pr = fig8_problem(plant, N, B); p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10); dt = 0.01
    names = {0: "store p", 1: "barrier A", 2: "S: window, half blocks, tbuf/rowbuf, p.Ap, wave sum", 3: "barrier B (carries the sum)", 4: "gather rows, alpha, x, r, store r",
             5: "barrier C", 6: "P^-1: window, half blocks, r.z, wave sum", 7: "barrier D (carries the sum)", 8: "gather rows, test, beta, p"}
s = NativeSolver(plant, N, B, dt=dt, **p)
if "rho" in pr: s.set_rho_penalty_batch(pr["rho"])
L = s.L
traced = hasattr(L, "gato_debug_pcg_trace")
s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
s.set_profiling(True)
if traced:
    L.gato_debug_pcg_trace.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.gato_debug_pcg_trace(None, 0, 1)
s.reset_dual(); s.reset_rho()
r = s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
st = s.stage_times_us()
it = r["pcg_iters_all"]
print("%s: %s N=%d B=%d; PCG launches of the solve: %.1f us each (stage clock); iterations per launch: mean %.1f, max per launch %s" % (
    "TRACED build" if traced else "product build", plant, N, B, st["pcg"] / it.shape[0], it.mean(), it.max(axis=1).tolist()))
if traced:
    n = 1024 * 16 * 16
    buf = np.zeros(n, np.uint64)
    L.gato_debug_pcg_trace(buf.ctypes.data_as(C.c_void_p), n, 0)
    tr = buf.reshape(1024, 16, 16).astype(np.float64)
    iters = tr[:, :, 15]
    nw = int((iters[0] > 0).sum())
    full = np.nonzero(iters[:, 0] >= 0.9 * iters[:, 0].max())[0]       # the workgroups of the LAST launch that ran (close to) the longest
    print("last launch: %d workgroups traced with >= %d iterations, %d wavefronts each; cycles per iteration (s_memtime, 100 MHz x ratio -> see clock), mean over those workgroups" % (
        len(full), int(0.9 * iters[:, 0].max()), nw))
    per = tr[full][:, :nw, :11] / iters[full][:, :nw, None]          # [wg][wave][slot]
    m = per.mean(axis=0)
    hdr = "%-52s" % "phase" + "".join("   wf%-2d" % w for w in range(nw)) + "    mean"
    print(hdr)
    for i in sorted(names):
        print("%-52s" % names[i] + "".join(" %6.0f" % m[w, i] for w in range(nw)) + "  %6.0f" % m[:, i].mean())
    print("%-52s" % "sum" + "".join(" %6.0f" % m[w].sum() for w in range(nw)) + "  %6.0f" % m.sum(axis=1).mean())
    print("spread over the workgroups of the per-iteration total: min %.0f median %.0f max %.0f" % (per.sum(axis=2).mean(axis=1).min(), np.median(per.sum(axis=2).mean(axis=1)), per.sum(axis=2).mean(axis=1).max()))
