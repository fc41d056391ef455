#!/usr/bin/env python3
"""Randomised sweep of the HIP path against the oracle (runs on the MI355X box): plant, horizon, batch, time step, wrench, cost weights,
rho and mu drawn at random (conditioning of the Schur system from benign to 1e12).  One SQP iteration with PCG at its floor must take the
fp32 oracle's steps and be no further from the FLOAT64 oracle than max(5e-4, 4 x the fp32 oracle's own distance to it) -- the criterion of
tests/test_gpu_parity.py::test_three_iterations_against_float64; a default-tolerance 3-iteration solve must stay finite and descend.
When that end-to-end bound fails the case is re-run stage by stage (KKT blocks, Q^-1, S, P^-1, gamma, dz from the float64 lambda): if every stage
tensor of the HIP path is within 4x of the fp32 oracle's own error the case is counted as ill-conditioned (equal stage errors amplified by
cond(S) -- the fp32 oracle's luck, not its accuracy), otherwise as a VIOLATION.  A line-search step that differs from the float64 step
must be a near tie in the float64 merits -- unless the fp32 ORACLE fails that same test on a trajectory of the case (then the case is beyond what
fp32 resolves: counted, not a violation; seed 403 case 76, indy7 N=128: the fp32 oracle's final merits are up to 4x the float64 ones).
Exits non-zero on a violation."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_amd._lib import NativeSolver  # noqa: E402
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS  # noqa: E402
from gato_amd.bsqp.workloads import fig8_problem  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402



def stage_report(plant, N, B, dt, p, pr, verbose=True):
    """Where the HIP path and the fp32 oracle each sit relative to the float64 oracle, stage by stage (same inputs; lambda and dz teacher-forced
    from float64 so that every row measures ONE stage)."""
    nat, o32, o64 = NativeSolver(plant, N, B, dt=dt, **p), OracleSolver(plant, N, B, dt=dt, **p), OracleSolver(plant, N, B, dt=dt, f64=True, **p)
    for s in (nat, o32, o64):
        s.set_f_ext_batch(pr["f_ext"])
    xu, xs, ref = pr["xu"], pr["x_s"], pr["ref"]

    out = {}

    def rs(a, b):
        a, b = np.asarray(a, np.float64).reshape(B, -1), np.asarray(b, np.float64).reshape(B, -1)
        return (np.abs(a - b).max(axis=1) / np.maximum(1e-30, np.abs(b).max(axis=1))).max()

    def row(n, h, o, note=""):
        out[n] = (float(h), float(o))
        if verbose:
            print("    %-6s HIP %.2e  fp32 %.2e   %s" % (n, h, o, note))
    nat.stage("kkt", xu, dt, xs, ref)
    o32.setup_kkt(xu, xs, ref, dt)
    o64.setup_kkt(xu, xs, ref, dt)
    dk = nat.dense_kkt(dt)
    for n in ("A", "B", "c", "Q", "q", "R", "r"):
        sl = slice(0, N - 1) if n in ("A", "B", "R", "r") else slice(None)
        row(n, rs(dk[n][:, sl], o64.buf(n)[:, sl]), rs(o32.buf(n)[:, sl], o64.buf(n)[:, sl]))
    nat.stage("schur", xu, dt, xs, ref)
    o32.form_schur()
    o64.form_schur()
    dk = nat.dense_kkt(dt)
    row("Qinv", rs(dk["Qinv"], o64.buf("Qinv")), rs(o32.buf("Qinv"), o64.buf("Qinv")))
    for n in ("S", "Pinv", "gamma"):
        row(n, rs(nat.read(n), o64.buf(n)), rs(o32.buf(n), o64.buf(n)))
    o64.pcg()
    lam = o64.buf("lambda")
    nat.write("lambda", lam.astype(np.float32))
    o32.set_lambda(lam)
    nat.stage("dz", xu, dt, xs, ref)
    o32.compute_dz()
    o64.compute_dz()
    row("dz", rs(nat.read("dz"), o64.buf("dz")), rs(o32.buf("dz"), o64.buf("dz")), "(from the float64 lambda)")
    return out, all(h <= max(4.0 * o, 2e-6) for h, o in out.values())


def run(cases, seed, verbose=True, only=None, maxb=6):
  rng = np.random.default_rng(seed)
  worst = dict(xu=0.0, merit=0.0)
  bad = amplified = beyond = nstalled = stall_hip = stall_orc = ntraj = 0
  for case in range(cases):
      plant = rng.choice(["indy7", "iiwa14"])
      N = int(rng.choice([4, 8, 16, 32, 64, 128]))
      B = int(rng.integers(1, maxb + 1))
      dt = float(rng.choice([0.005, 0.01, 0.02, 0.05]))
      p = dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=1, pcg_tol=1e-9, max_pcg_iters=1000, rho=float(10 ** rng.uniform(-4, -1)),
               mu=float(rng.choice([1.0, 10.0, 50.0])), q_cost=float(rng.choice([0.5, 2.0, 10.0])), qd_cost=float(10 ** rng.uniform(-4, -1)),
               u_cost=float(10 ** rng.uniform(-7, -5)), N_cost=float(rng.choice([10.0, 50.0, 100.0])), q_lim_cost=float(rng.choice([0.0, 0.01])),
               vel_lim_cost=float(rng.choice([0.0, 1e-3])), ctrl_lim_cost=float(rng.choice([0.0, 1e-4])))
      pr = fig8_problem(plant, N, B, seed=int(rng.integers(0, 1000)), dt=0.01, f_ext_std=float(rng.choice([0.0, 3.0])))
      if only is not None and case not in only:
          continue
      nat = NativeSolver(plant, N, B, dt=dt, **p)
      orc = OracleSolver(plant, N, B, dt=dt, **p)
      o64 = OracleSolver(plant, N, B, dt=dt, f64=True, **p)
      for s in (nat, orc, o64):
          s.set_f_ext_batch(pr["f_ext"])
      rg = nat.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
      ro = orc.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
      r6 = o64.solve(pr["xu"], dt, pr["x_s"], pr["ref"])

      def terr(x, y):
          return np.abs(np.asarray(x, np.float64) - y).reshape(B, -1).max(axis=1) / np.maximum(1.0, np.abs(y).reshape(B, -1).max(axis=1))
      e, eo = terr(rg["XU"], r6["XU"]), terr(ro["XU"], r6["XU"])
      m = float(np.abs(rg["initial_merit"] - ro["initial_merit"]).max() / max(1e-30, np.abs(ro["initial_merit"]).max()))
      clear = np.array_equal(ro["ls_step_size"], r6["ls_step_size"].astype(np.float32))     # the fp32 oracle itself keeps the float64 steps
      same = np.array_equal(rg["ls_step_size"], ro["ls_step_size"]) or not clear
      worst["xu"], worst["merit"] = max(worst["xu"], float(e.max())), max(worst["merit"], m)
      worst["ratio"] = max(worst.get("ratio", 0.0), float(e.max() / max(eo.max(), 1.25e-4)))
      worst["o32"] = max(worst.get("o32", 0.0), float(eo.max()))
      if only is not None:
          natd = NativeSolver(plant, N, B, dt=dt, **p)
          natd.set_f_ext_batch(pr["f_ext"])
          natd.set_linear_solver("direct")
          rd = natd.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
          print("case %d %s N=%d B=%d dt=%g rho=%.3e mu=%g" % (case, plant, N, B, dt, p["rho"], p["mu"]))
          print("  pcg iters HIP", rg["pcg_iters"].ravel(), "fp32", ro["pcg_iters"].ravel(), "f64", r6["pcg_iters"].ravel())
          print("  steps HIP", rg["ls_step_size"].ravel(), "fp32", ro["ls_step_size"].ravel(), "f64", r6["ls_step_size"].ravel(), "direct", rd["ls_step_size"].ravel())
          print("  err HIP", e, "\n  err fp32", eo, "\n  err direct", terr(rd["XU"], r6["XU"]))
          print("  final merit HIP", rg["final_merit"].ravel(), "\n  fp32", ro["final_merit"].ravel(), "\n  f64", r6["final_merit"].ravel(), "\n  direct", rd["final_merit"].ravel())
      # trajectories on which an fp32 PCG never meets the floor tolerance (|r.z| stalls above 1e-6 + 1e-9 rho0 and the recurrence drifts
      # for all 1000 iterations -- either fp32 implementation does this on extreme configurations, float64 converges in ~10) are fp32's
      # limit, not a comparison: they are left out and counted
      stalled = (rg["pcg_iters"][0] >= p["max_pcg_iters"]) | (ro["pcg_iters"][0] >= p["max_pcg_iters"])
      nstalled += int(stalled.sum())
      stall_hip += int((rg["pcg_iters"][0] >= p["max_pcg_iters"]).sum())
      stall_orc += int((ro["pcg_iters"][0] >= p["max_pcg_iters"]).sum())
      ntraj += B
      live = ~stalled
      e_l, eo_l = (e[live] if live.any() else np.zeros(1)), (eo[live] if live.any() else np.zeros(1))
      same_l = np.array_equal(rg["ls_step_size"][:, live], ro["ls_step_size"][:, live]) or not clear
      ok = same_l and e_l.max() <= max(5e-4, 4.0 * eo_l.max()) and m < 1e-5 and np.all(np.isfinite(rg["XU"][live]))
      nat2 = NativeSolver(plant, N, B, dt=dt, **dict(p, max_sqp_iters=3, pcg_tol=1e-4, max_pcg_iters=200))
      nat2.set_f_ext_batch(pr["f_ext"])
      r2 = nat2.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
      ok = ok and bool(np.all(np.isfinite(r2["XU"])) and np.all(r2["final_merit"] <= r2["initial_merit"]))
      # a line search that differs from the float64 one must be a near tie THERE: the float64 merits of the two choices within 1e-3
      tie = True
      # the same question put to the fp32 ORACLE: does it keep the float64 steps up to a tie on every trajectory of this case?  Where the
      # reference's own arithmetic does not, the case is beyond what fp32 resolves and is counted as such, not as a violation of the HIP path
      oracle_tie = True
      for b_ in range(B):
          if stalled[b_]:
              continue
          so, s6 = float(ro["ls_step_size"][0, b_]), float(r6["ls_step_size"][0, b_])
          if so != np.float32(s6):
              cand = {**{float(2.0 ** -i): float(r6["ls_merits"][0, b_, i]) for i in range(8)}, -1.0: float(r6["ls_merit_before"][0, b_])}
              # a FIXED 1e-3 here: the oracle's own candidate errors cannot be the yardstick of the oracle's own choice
              if not abs(cand.get(so, np.inf) - cand[s6]) <= 1e-3 * max(1.0, abs(cand[s6])):
                  oracle_tie = False
      for b_ in range(B):
          if stalled[b_]:
              continue
          sg, s6 = float(rg["ls_step_size"][0, b_]), float(r6["ls_step_size"][0, b_])
          if sg != np.float32(s6):
              cand = {**{float(2.0 ** -i): float(r6["ls_merits"][0, b_, i]) for i in range(8)}, -1.0: float(r6["ls_merit_before"][0, b_])}
              mg, m6 = cand.get(sg, np.inf), cand[s6]
              # ... within what fp32 can resolve of these merits: the fp32 oracle's own error on the candidates (4 x, floor 1e-3)
              c32 = np.concatenate([ro["ls_merits"][0, b_].astype(np.float64), [float(ro["ls_merit_before"][0, b_])]])
              c64 = np.concatenate([r6["ls_merits"][0, b_], [float(r6["ls_merit_before"][0, b_])]])
              res = max(1e-3, 4.0 * float(np.max(np.abs(c32 - c64) / np.maximum(1.0, np.abs(c64)))))
              if not abs(mg - m6) <= res * max(1.0, abs(m6)):
                  tie = False
                  if verbose:
                      print("   pcg iterations HIP %s fp32 oracle %s float64 %s" % (rg["pcg_iters"].ravel(), ro["pcg_iters"].ravel(), r6["pcg_iters"].ravel()))
                      print("   trajectory %d: HIP step %g (float64 merit %.6g), float64 step %g (%.6g), fp32 oracle step %g; fp32 resolution of these merits %.1e" % (
                          b_, sg, mg, s6, m6, float(ro["ls_step_size"][0, b_]), res))
      if not tie and not oracle_tie:
          beyond += 1
          print("beyond fp32 case %d: %s N=%d B=%d dt=%g rho=%.2e  the fp32 oracle itself leaves the float64 steps without a tie on this case" % (
              case, plant, N, B, dt, p["rho"]), flush=True)
          continue
      if not tie:
          # the same second look the end-to-end bound gets: stage by stage against the float64 oracle.  Every stage tensor as accurate as the fp32
          # oracle's means the merits of this trajectory amplify a stage-level difference of rounding size into a different argmin
          # (seed 501 --maxb 48 case 43: iiwa14 N=16 dt=0.02, PCG counts identical on all 35 rows, iterate errors of BOTH fp32 paths 2e-4 .. 6e-3)
          st, stages_ok = stage_report(plant, N, B, dt, p, pr, verbose=only is not None)
          if stages_ok:
              amplified += 1
              print("ill-conditioned case %d: %s N=%d B=%d dt=%g rho=%.2e  a step differs from the float64 step beyond the tie tolerance, every stage within 4x of the "
                    "fp32 oracle's error (worst stage: %s)" % (case, plant, N, B, dt, p["rho"], max(st, key=lambda k: st[k][0] / max(st[k][1], 5e-7))), flush=True)
              continue
          bad += 1
          print("VIOLATION case %d: %s N=%d B=%d dt=%g rho=%.2e  a step differs from the float64 step without a tie in the float64 merits" % (
              case, plant, N, B, dt, p["rho"]), flush=True)
          continue
      if not ok and np.all(np.isfinite(rg["XU"])) and m < 1e-5 and np.all(np.isfinite(r2["XU"])) and np.all(r2["final_merit"] <= r2["initial_merit"]):
          # the end-to-end bound failed: is any single stage of the HIP path less accurate than the fp32 oracle's, or is this the
          # amplification of equally small stage errors by an ill-conditioned Schur system?
          st, stages_ok = stage_report(plant, N, B, dt, p, pr, verbose=only is not None)
          if stages_ok:
              amplified += 1
              print("ill-conditioned case %d: %s N=%d B=%d dt=%g rho=%.2e  every stage within 4x of the fp32 oracle's error (worst stage: %s), "
                    "end to end HIP-f64 %.2e  fp32oracle-f64 %.2e  steps equal %s" % (
                        case, plant, N, B, dt, p["rho"], max(st, key=lambda k: st[k][0] / max(st[k][1], 5e-7)), e.max(), eo.max(), same), flush=True)
              continue
      if not ok:
          bad += 1
          print("VIOLATION case %d: %s N=%d B=%d dt=%g rho=%.2e  steps equal %s  HIP-f64 %.2e  fp32oracle-f64 %.2e  merit %.2e" % (
              case, plant, N, B, dt, p["rho"], same, e.max(), eo.max(), m), flush=True)
  if verbose:
    print("cases %d  violations %d  beyond fp32 (the fp32 oracle leaves the float64 steps too) %d  ill-conditioned (stage-accurate, end-to-end amplified) %d  trajectories with a stalled fp32 PCG (left out) %d of %d (HIP path %d, fp32 oracle %d)  worst HIP-vs-float64 iterate error %.2e (fp32 oracle's worst %.2e), worst ratio HIP : max(oracle, 1.25e-4) = %.2f, "
          "worst initial-merit error %.2e" % (cases, bad, beyond, amplified, nstalled, ntraj, stall_hip, stall_orc, worst["xu"], worst.get("o32", 0.0), worst.get("ratio", 0.0), worst["merit"]))
  return bad, worst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, nargs="*", default=None, help="diagnose these case numbers of the seed")
    ap.add_argument("--maxb", type=int, default=6, help="batch sizes are drawn from 1..maxb (default 6: the seeds quoted in DESIGN.md)")
    a = ap.parse_args()
    sys.exit(1 if run(a.cases, a.seed, only=a.only, maxb=a.maxb)[0] else 0)
