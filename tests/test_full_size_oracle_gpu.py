"""The CPU oracle on EVERY trajectory of the BASELINE configurations (round-3 review, item 3): C2 indy7 N=32 B=1024, C3 iiwa14 N=128 B=256,
C5's per-GPU shard iiwa14 N=64 B=512 at the sweep's settings (per-trajectory rho over nine decades, dt 0.05, mu 1, pcg_tol 1e-3).  The
reference side is bsqp.cuh:121-176 on python/bsqp/config.py:35-50's parameters -- the bench workload.  Both oracle builds run with OpenMP over
the trajectories (seconds, not minutes); the library's own float64 build (tests/test_f64_gpu.py) stays as the second opinion.

Three results per row: HIP fp32, oracle fp32, oracle float64 (the arbiter of what fp32 can resolve).  A row is RESOLVED when fp32 itself is
meaningful on it: the fp32 oracle takes the float64 oracle's step and lands within 1e-3 of it, and no PCG ran into its cap.  (C5 is where that
matters: the sweep (x_s = 0, one random goal, dt 0.05, rho over nine decades) is beyond fp32 on EVERY row -- the fp32 oracle is 1.5e-2 (median)
from its own float64 build after one iteration at rho <= 1e-3 and does not converge at all at rho >= 1e-2, where float64 needs ~25 PCG
iterations and fp32 runs into the cap; iterates reach |XU| ~ 700.  There are no resolved rows there (the row-by-row assertions are skipped
below 16 of them) and what is asserted is the statistical statement of the last block: the HIP path is as close to float64, and follows the
float64 steps as often, as the fp32 oracle does.)  Asserted:
  resolved rows      the HIP path takes the fp32 oracle's line-search step, except through a NEAR TIE (the two candidates' merits within TIE
                     of each other on either side's own merit table; at most 2 % of the rows) -- C2 / C3: no other departure; C5: at most
                     5 % (a row the fp32 oracle resolves by luck is not resolved for another summation order; the symmetric count is the
                     last assertion); per-trajectory iterate error against the fp32 oracle max|XU - XU*| / max(1, max|XU*|) within
                     XU_BOUND[plant] = (max, 99th percentile, median) with PCG at its floor, 3 x that at the default tolerance (the exit
                     test is a discontinuity: a count that differs by one moves lambda within pcg_tol); there also: PCG iteration counts
                     within +-1 on >= 99 %
  every row          finite; initial merit to 1e-5; the HIP path is as close to float64 as the fp32 oracle is: median distance <= 2 x the
                     oracle's (+1e-5), and it follows the float64 steps on as many rows (-3 %)
Measured numbers go to gpurun_out/r06_parity_full_size.jsonl (copied to profiles/r06_parity_full_size.json; round 5's: profiles/r05_parity_full_size.json)."""
import json
import os

import numpy as np
import pytest

from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem, hparam_problem

pytestmark = pytest.mark.gpu

CASES = {"C2": ("indy7", 32, 1024, "fig8"), "C3": ("iiwa14", 128, 256, "fig8"), "C5": ("iiwa14", 64, 512, "hparam"),
         "C4": ("indy7", 32, 1024, "fig8")}   # C4: one rank's 1024 rows of the 8192-row batch (shard = rank)
# per-trajectory iterate error HIP vs fp32 oracle over the resolved rows: (max, 99th percentile, median).  Measured (profiles/r04_parity_full_size.json):
# C2 4.7e-4 / 1.9e-4 / 3.1e-5, C3 7.3e-4 / 4.1e-4 / 8.3e-5 at PCG's floor -- the north-star's 1e-4 holds in the median and up to the ~95th
# percentile; the tail is cond(S) ~ 1e9 .. 1e10 times fp32 rounding of S and gamma (tools/stage_errors.py: no single stage carries it)
XU_BOUND = {"indy7": (1e-3, 4e-4, 1e-4), "iiwa14": (1.5e-3, 8e-4, 2e-4)}
TIE = 2e-2   # two candidate merits closer than this (relative) are one decision to fp32: 1e-4 in XU is ~1e-2 in the merit (mu |defect|_1 through M^-1)
MIN_RESOLVED = {"C2": 0.97, "C3": 0.97, "C5": 0.0, "C4": 0.97}
MAX_DEPART = {"C2": 0.0, "C3": 0.0, "C5": 0.05, "C4": 0.0}   # resolved rows that leave the fp32 oracle's step without a tie
CHAOTIC = {"C5"}   # no resolved rows to speak of: only the statistical block is asserted, with the margins of a regime where fp32 itself is noise


def _report(**kw):
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "r06_parity_full_size.jsonl"), "a") as f:
            f.write(json.dumps({k: (v if isinstance(v, (str, bool, int, list)) else float(v)) for k, v in kw.items()}) + "\n")
    except OSError:
        pass


def _err(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max(axis=1) / np.maximum(1.0, np.abs(np.asarray(b, np.float64)).max(axis=1))


def _problem(case, shard):
    """C2 / C3: the single-GPU batches; C4 shard r = rows 1024 r .. 1024 r + 1023 of the 8192-row fig-8 batch (bench.py's rank r); C5 shard g =
    cost tuple g of the sweep grid with its own 512 goals (bench.py --workload hparam, rank g)"""
    plant, N, B, kind = CASES[case]
    if kind == "hparam":
        pr = hparam_problem(plant, N, B, shard=3 if shard is None else shard)
        return plant, N, B, pr, dict(pr["params"]), pr["dt"]
    pr = fig8_problem(plant, N, B, batch_offset=B * (shard or 0))
    return plant, N, B, pr, dict(DEFAULT_SOLVER_PARAMS), 0.01


def _run(case, shard=None, **over):
    from gato_amd._lib import NativeSolver
    from oracle.oracle import OracleSolver
    plant, N, B, pr, p, dt = _problem(case, shard)
    p.update(over)
    nt = os.cpu_count() or 1
    solvers = {"hip": NativeSolver(plant, N, B, dt=dt, **p), "o32": OracleSolver(plant, N, B, dt=dt, threads=nt, **p),
               "o64": OracleSolver(plant, N, B, dt=dt, threads=nt, f64=True, **p)}
    out = {}
    for k, s in solvers.items():
        if "rho" in pr:
            s.set_rho_penalty_batch(pr["rho"])
        out[k] = s.solve(pr["xu"], dt, pr["x_s"], pr["ref"])
    out["hip"]["ls_merits"] = solvers["hip"].read("merit").reshape(1, B, 8)   # the HIP path's own candidates of its (only) line search
    return plant, N, B, p, out


def _check(case, tag, plant, N, B, p, out, pcg_counts, xu_scale=1.0):
    g, o32, o64 = out["hip"], out["o32"], out["o64"]
    cap = p["max_pcg_iters"]
    assert np.all(np.isfinite(g["XU"]))
    assert np.abs(g["initial_merit"] - o32["initial_merit"]).max() <= 1e-5 * np.abs(o32["initial_merit"]).max()
    sg, s32, s64 = g["ls_step_size"][0], o32["ls_step_size"][0], o64["ls_step_size"][0].astype(np.float32)
    e32, eg, ego = _err(o32["XU"], o64["XU"]), _err(g["XU"], o64["XU"]), _err(g["XU"], o32["XU"])
    below = (g["pcg_iters"][0] < cap) & (o32["pcg_iters"][0] < cap) & (o64["pcg_iters"][0] < cap)
    resolved = (s32 == s64) & (e32 <= 1e-3) & (below if cap >= 1000 else True)
    same = sg == s32
    # a resolved row may leave the oracle's step only through a near tie, on either side's own merit table
    ties, departs = 0, []
    for b in np.nonzero(resolved & ~same)[0]:
        ok = False
        for r in (g, o32):
            cand = {**{float(2.0 ** -i): float(r["ls_merits"][0, b, i]) for i in range(8)}, -1.0: float(o32["ls_merit_before"][0, b])}
            ok = ok or abs(cand[float(sg[b])] - cand[float(s32[b])]) <= TIE * max(1.0, abs(cand[float(s32[b])]))
        if ok:
            ties += 1
        else:
            departs.append((int(b), float(sg[b]), float(s32[b])))
    if case not in CHAOTIC:
        assert len(departs) <= MAX_DEPART[case] * resolved.sum(), "%s: rows that take another step than the oracle and neither side's merits call it a tie: %r" % (case, departs[:10])
    use = resolved & same
    enough = int(use.sum()) >= 16
    dp = np.abs(g["pcg_iters"][0].astype(int) - o32["pcg_iters"][0].astype(int))
    # at the default tolerance PCG's exit test is a discontinuity: a row whose residual measure sits on the threshold leaves a few iterations earlier
    # or later on another summation order and its lambda then differs by what those iterations would still have changed (measured on C4 shard 2:
    # one row of 1024, 34 against 37 iterations, 8.9e-3; the other 1023 rows <= 4.7e-4).  The maximum is taken over the rows with EQUAL counts;
    # rows with other counts (at most 1 %, asserted below) get 10 x the bound
    eq = (dp == 0) if pcg_counts else np.ones(B, bool)
    if enough:
        assert (use & eq).any(), "%s: no resolved row has the oracle's PCG count -- nothing to take the maximum over (a count shift on every row is a regression)" % case
    mx, p99, med = (float(ego[use & eq].max()), float(np.quantile(ego[use], 0.99)), float(np.median(ego[use]))) if enough else (0.0, 0.0, 0.0)
    mx_other = float(ego[use & ~eq].max()) if (use & ~eq).any() else 0.0
    # the 10 x class is for the odd row that sits ON the exit threshold, not for a kernel that drifts: at most 1 % of the rows may have another count
    # than the oracle, by at most 3 iterations (measured: C2 and seven C4 shards none, C4 shard 2 one row of 1024 at 34 vs 37, C3 two rows of 256 by
    # one) -- a regression that shifts counts on more rows, or further, fails HERE instead of being absorbed by the wider bound
    if pcg_counts and enough:
        other = use & ~eq
        assert other.sum() <= max(1, int(0.01 * B)) and (dp[other].max() if other.any() else 0) <= 3, (int(other.sum()), dp[other].tolist()[:10])
    _report(test=tag, case=case, plant=plant, N=N, B=B, resolved_rows=int(resolved.sum()), steps_equal_on_resolved=int(use.sum()), near_ties=ties, departures=len(departs),
            xu_vs_fp32_oracle_max=mx, xu_vs_fp32_oracle_max_rows_with_other_pcg_counts=mx_other, xu_vs_fp32_oracle_p99=p99, xu_vs_fp32_oracle_median=med,
            rows_within_1e_4_of_the_fp32_oracle=float((ego[use] <= 1e-4).mean()) if enough else 0.0,      # the north-star's "iterate match within 1e-4 rel", row by row
            all_rows_median_dist_to_f64_hip=float(np.median(eg)), all_rows_median_dist_to_f64_oracle32=float(np.median(e32)),
            rows_on_f64_steps_hip=int((sg == s64).sum()), rows_on_f64_steps_oracle32=int((s32 == s64).sum()),
            rows_on_fp32_oracle_steps=int(same.sum()), pcg_equal_on_resolved=int((dp[resolved] == 0).sum()), pcg_within_1_on_resolved=int((dp[resolved] <= 1).sum()), pcg_max=int(max(g["pcg_iters"][0].max(), o32["pcg_iters"][0].max())))
    assert resolved.mean() >= MIN_RESOLVED[case], resolved.sum()
    assert ties <= max(1, 0.02 * resolved.sum()), (ties, resolved.sum())
    bmx, b99, bmed = (xu_scale * v for v in XU_BOUND[plant])
    assert mx <= bmx and p99 <= b99 and med <= bmed and mx_other <= 10 * bmx, (mx, p99, med, mx_other)
    if enough and case not in CHAOTIC:
        # the north-star's own figure, row by row (measured: indy7 0.91-0.93 of the rows on every shard of C4, iiwa14 N = 128 0.61; the fp32 oracle is no
        # closer to its float64 build: the next assertions)
        assert (ego[use] <= 1e-4).mean() >= (0.88 if plant == "indy7" else 0.5), (ego[use] <= 1e-4).mean()
    if case in CHAOTIC:
        # the sweep: the fp32 oracle itself takes the float64 step on fewer than half of the rows.  Two fp32 paths agree with each other far
        # more often than either agrees with float64 (they share the arithmetic, not the summation order), and the HIP path follows float64
        # about as often as the oracle does
        assert same.mean() >= 0.7, same.sum()
        assert (sg == s64).sum() >= (s32 == s64).sum() - 0.10 * B, ((sg == s64).sum(), (s32 == s64).sum())
        return
    assert np.median(eg) <= 2 * np.median(e32) + 1e-5, (np.median(eg), np.median(e32))
    assert (sg == s64).sum() >= (s32 == s64).sum() - 0.03 * B, ((sg == s64).sum(), (s32 == s64).sum())
    assert same.sum() >= (s32 == s64).sum() - 0.10 * B, (same.sum(), (s32 == s64).sum())   # two fp32 paths agree about as often as fp32 agrees with float64
    if pcg_counts and enough:
        assert (dp[resolved] <= 1).mean() >= 0.99, (dp[resolved] <= 1).sum()
        assert np.array_equal(g["kkt_converged"], o32["kkt_converged"])


@pytest.mark.parametrize("case", ["C2", "C3", "C5"])
def test_one_iteration_at_the_pcg_floor_every_trajectory(case):
    plant, N, B, p, out = _run(case, max_sqp_iters=1, pcg_tol=1e-9, max_pcg_iters=1000)
    _check(case, "tight_1it_full", plant, N, B, p, out, pcg_counts=False)


@pytest.mark.parametrize("case", ["C2", "C3", "C5"])
def test_first_iteration_decisions_at_the_default_tolerance_every_trajectory(case):
    """the bench workload's own settings (DEFAULT_SOLVER_PARAMS / the sweep's): what the first iteration of every timed solve decides"""
    plant, N, B, p, out = _run(case, max_sqp_iters=1)
    _check(case, "default_1it_full", plant, N, B, p, out, pcg_counts=True, xu_scale=3.0)


@pytest.mark.parametrize("case,iters", [("C2", 3), ("C3", 3), ("C2", 10), ("C3", 10), ("C5", 10)])
def test_free_running_iterations_every_trajectory(case, iters):
    """Several FREE-RUNNING iterations at the workload's own settings, every trajectory -- 10 is the bench workload itself (bench.py times exactly
    this solve).  Row by row two fp32 implementations part ways after a few iterations (line-search decisions flip on rounding: DESIGN.md
    section 3), so what is asserted is what the reference's own fp32 build could promise: the HIP path and the fp32 oracle are the SAME DISTANCE
    from the float64 oracle -- in the iterates, in how many rows still follow the float64 steps, in the merit they reach and in the PCG work
    they need -- and every row descends.  (The first iteration's decisions are asserted exactly above.)"""
    plant, N, B, p, out = _run(case, max_sqp_iters=iters)
    g, o32, o64 = out["hip"], out["o32"], out["o64"]
    assert g["iters_done"] == o32["iters_done"] == iters
    assert np.all(np.isfinite(g["XU"])) and np.all(g["final_merit"] <= g["initial_merit"])
    eg, e32 = _err(g["XU"], o64["XU"]), _err(o32["XU"], o64["XU"])
    on64_g = np.all(g["ls_step_size"] == o64["ls_step_size"].astype(np.float32), axis=0)
    on64_o = np.all(o32["ls_step_size"] == o64["ls_step_size"].astype(np.float32), axis=0)
    fm = {k: np.median(out[k]["final_merit"].astype(np.float64)) for k in out}
    it = {k: float(out[k]["pcg_iters"].mean()) for k in out}
    _report(test="free_%dit_full" % iters, case=case, plant=plant, N=N, B=B, median_dist_to_f64_hip=float(np.median(eg)), median_dist_to_f64_oracle32=float(np.median(e32)),
            p90_dist_to_f64_hip=float(np.quantile(eg, 0.9)), p90_dist_to_f64_oracle32=float(np.quantile(e32, 0.9)),
            rows_on_f64_steps_hip=int(on64_g.sum()), rows_on_f64_steps_oracle32=int(on64_o.sum()),
            median_final_merit_hip=float(fm["hip"]), median_final_merit_oracle32=float(fm["o32"]), median_final_merit_oracle64=float(fm["o64"]),
            mean_pcg_iters_hip=it["hip"], mean_pcg_iters_oracle32=it["o32"], mean_pcg_iters_oracle64=it["o64"])
    # as close to float64 as the fp32 oracle is (a factor 2 and a floor: medians of heavy-tailed distributions of a few hundred rows)
    assert np.median(eg) <= 2.0 * np.median(e32) + 1e-4, (np.median(eg), np.median(e32))
    assert np.quantile(eg, 0.9) <= 2.5 * np.quantile(e32, 0.9) + 1e-3, (np.quantile(eg, 0.9), np.quantile(e32, 0.9))
    assert on64_g.sum() >= on64_o.sum() - 0.15 * B, (on64_g.sum(), on64_o.sum())
    # the same optimisation result in distribution: the merit reached and the linear-solver work spent
    assert abs(fm["hip"] - fm["o32"]) <= 0.10 * abs(fm["o32"]) + 1e-6, fm
    assert abs(it["hip"] - it["o32"]) <= 0.10 * it["o32"] + 0.5, it


# ---- every shard of the multi-GPU configurations, deterministically (round-4 review, item 1) ---------------------------------------------
# One SQP iteration taken apart: every stage of the HIP path runs from the ORACLE's upstream tensor (the same bits on both sides), so neither
# cond(S) nor a flipped line-search decision can carry an error from one stage into the next -- no PCG in the loop, no statistical statement.
# bsqp.cuh:121-176, stage by stage ("rel" = max|a - b| / max|b| over the buffer, or over the trajectory's slice of it where it says so):
#   merit of the warm start (merit.cuh:17-92)                                     per trajectory rel <= 1e-5
#   KKT blocks A, B, c, Q, q, R, r (setup_kkt.cuh:15-108)                          per buffer rel <= 1e-5
#   (Q + rho I)^-1, R^-1, S (schur_linsys.cuh:14-164)                              per buffer rel <= 1e-4 (Gauss-Jordan without pivoting)
#   gamma (schur_linsys.cuh:81,121-128), dz FROM THE ORACLE'S lambda (:316-431)    C4: per buffer 1e-4 / per trajectory 1e-5.  Both configurations,
#                                                                                  EVERY COMPONENT: within the forward rounding-error bound of the
#                                                                                  formula on the path's own inputs (tests/stage_bounds.py)
#   P^-1 = -(theta + rho I)^-1 and its stair blocks (schur_linsys.cuh:150-260)     C4: per buffer rel <= 1e-4; C5: see below
#   the 8 merits FROM THE ORACLE'S dz (merit.cuh:17-92)                            per trajectory rel <= 1e-5
#   line search + rho / drho + the new iterate FROM THE ORACLE'S MERIT TABLE (line_search.cuh:13-98): step, rho, drho, merit_cur and xu equal
#   BIT FOR BIT (a power-of-two step times dz is exact, so even the update has one correct rounding)
# What the sweep (C5) cannot give, and why (measured: profiles/r05_c5_stage_conditioning.txt): with u_cost = 1e-6 .. 1e-7 the control step is
# dz_u = -R^-1 (r + B^T lambda) = 1e7 x a 14-term dot product that cancels to 1e-4 of its terms, and cond(theta + rho I) ~ 4e9.  From IDENTICAL
# inputs the fp32 oracle is up to 3.7e-2 (dz), 2.7e-4 (gamma), 3e-1 (a P^-1 block) from its own float64 build -- these three single-stage maps are
# ill-conditioned, whoever evaluates them.  So there: gamma and dz are held to the a-priori bound component by component (a theorem for any correct
# fp32 evaluation, violated by any indexing / sign / layout bug: those produce errors of the size of the terms, 1e5 x the bound), and the HIP path's
# own inputs to those formulas are the tensors already compared above; P^-1 is held to the float64 oracle as arbiter: the HIP path's worst block is
# closer to the fp32 oracle than HALF the fp32 oracle's worst block is to float64 (measured: 5x .. 40x closer).
SHARDS = [("C4", r) for r in range(8)] + [("C5", g) for g in range(8)] + [("C3", 0)]   # + the long horizon (iiwa14 N = 128, 256 rows) for completeness


def _rows(a, b, floor=0.0):
    """per-trajectory max|a - b| / max(floor, max|b|)"""
    a = np.asarray(a, np.float64).reshape(len(a), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    return np.abs(a - b).max(axis=1) / np.maximum(np.maximum(floor, 1e-30), np.abs(b).max(axis=1))


def _buf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def _blocks(a, b):
    """per (trajectory, knot) block: max|a - b| / max|b|, max over the blocks"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b).max(axis=(2, 3)) / np.maximum(1e-300, np.abs(b).max(axis=(2, 3)))).max())


@pytest.mark.parametrize("case,shard", SHARDS, ids=["%s-shard%d" % cs for cs in SHARDS])
def test_one_iteration_stage_by_stage_on_every_shard(case, shard):
    import stage_bounds as SB
    from gato_amd._lib import NativeSolver
    from oracle.oracle import OracleSolver
    plant, N, B, pr, p, dt = _problem(case, shard)
    nt = os.cpu_count() or 1
    nat = NativeSolver(plant, N, B, dt=dt, **p)
    orc = OracleSolver(plant, N, B, dt=dt, threads=nt, **p)
    o64 = OracleSolver(plant, N, B, dt=dt, threads=nt, f64=True, **p) if case in CHAOTIC else None
    if "rho" in pr:
        for s in (nat, orc, o64):
            s.set_rho_penalty_batch(pr["rho"])
    xu, xs, ref = pr["xu"], pr["x_s"], pr["ref"]
    nx, nu = nat.nx, nat.nu
    m = {}
    # the warm start's merit
    nat.stage("merit1", xu, dt, xs, ref)
    m0 = orc.merit(xu, xs, ref, dt, num_alphas=1, zero_dz=True)[:, 0]
    m["merit0"] = _rows(nat.read("merit_cur")[:, None], m0[:, None]).max()
    assert m["merit0"] <= 1e-5
    # KKT blocks
    nat.stage("kkt", xu, dt, xs, ref)
    orc.setup_kkt(xu, xs, ref, dt)
    dk = nat.dense_kkt(dt)
    lin_g = {k: dk[k] for k in ("q", "r", "c")}                       # the linearisation's q, r (computeDz overwrites them with the residuals)
    lin_o = {k: orc.buf(k) for k in ("q", "r", "c")}
    for name in ("A", "B", "R", "r"):
        m[name] = _buf(dk[name][:, :N - 1], orc.buf(name)[:, :N - 1])
    for name in ("c", "Q", "q"):
        m[name] = _buf(dk[name], orc.buf(name))
    for name in ("A", "B", "c", "Q", "q", "R", "r"):
        assert m[name] <= 1e-5, (name, m[name])
    # Schur complement, preconditioner, right-hand side
    nat.stage("schur", xu, dt, xs, ref)
    orc.form_schur()
    dk = nat.dense_kkt(dt)
    lin_g.update({k: dk[k] for k in ("A", "B", "Qinv", "Rinv")})
    lin_o.update({k: orc.buf(k) for k in ("A", "B", "Qinv", "Rinv")})
    lin_o["A"][:, N - 1] = 0
    lin_o["B"][:, N - 1] = 0
    m["Qinv"] = _buf(dk["Qinv"], orc.buf("Qinv"))
    m["Rinv"] = _buf(dk["Rinv"][:, :N - 1], orc.buf("Rinv")[:, :N - 1])
    got = {name: nat.read(name).reshape(orc.buf(name).shape) for name in ("S", "Pinv", "gamma")}
    for name in ("S", "Pinv", "gamma"):
        m[name] = _buf(got[name], orc.buf(name))
    for name in ("Qinv", "Rinv", "S"):
        assert m[name] <= 1e-4, (name, m[name])
    # gamma: every component within the rounding-error bound of the formula on the path's own inputs (those inputs are the tensors compared above)
    Kg = 2 * nx + 6
    g_g, b_g = SB.gamma_exact_and_bound(*(lin_g[k] for k in ("A", "B", "Qinv", "Rinv", "q", "r", "c")))
    g_o, b_o = SB.gamma_exact_and_bound(*(lin_o[k] for k in ("A", "B", "Qinv", "Rinv", "q", "r", "c")))
    m["gamma_bound_ratio"] = SB.ratio(got["gamma"], g_g, b_g, Kg)
    m["gamma_bound_ratio_oracle"] = SB.ratio(orc.buf("gamma"), g_o, b_o, Kg)
    m["gamma_inputs_ratio"] = SB.ratio(g_g, g_o, b_o, Kg)
    assert m["gamma_bound_ratio"] <= 1.0 and m["gamma_bound_ratio_oracle"] <= 1.0, m    # (the inputs' own few-ulp differences, amplified: reported)
    if case in CHAOTIC:
        o64.setup_kkt(xu, xs, ref, dt)
        o64.form_schur()
        P32, P64 = orc.buf("Pinv"), o64.buf("Pinv")
        m["Pinv_worst_block_vs_oracle32"] = _blocks(got["Pinv"][:, 1:], P32[:, 1:])
        m["Pinv_worst_block_oracle32_vs_f64"] = _blocks(P32[:, 1:], P64[:, 1:])
        m["gamma_rows_oracle32_vs_f64"] = float(_rows(orc.buf("gamma"), o64.buf("gamma")).max())
        m["gamma_rows"] = float(_rows(got["gamma"], orc.buf("gamma")).max())
        assert m["Pinv_worst_block_vs_oracle32"] <= max(1e-4, 0.5 * m["Pinv_worst_block_oracle32_vs_f64"]), m
        assert m["gamma_rows"] <= max(1e-4, 2.0 * m["gamma_rows_oracle32_vs_f64"]), m
    else:
        assert m["Pinv"] <= 1e-4 and m["gamma"] <= 1e-4, m
    # dz and the residuals from the oracle's lambda (its own PCG at the workload's tolerance: any lambda would do, both sides get the same bits)
    orc.pcg()
    lam = orc.buf("lambda")
    nat.write("lambda", lam)
    nat.stage("dz", xu, dt, xs, ref)
    orc.compute_dz()
    dz = orc.buf("dz")
    dz_g = nat.read("dz").reshape(B, -1)
    Kd = 2 * nx + 4
    e_g, bd_g = SB.dz_exact_and_bound(*(lin_g[k] for k in ("A", "B", "Qinv", "Rinv", "q", "r")), lam)
    e_o, bd_o = SB.dz_exact_and_bound(*(lin_o[k] for k in ("A", "B", "Qinv", "Rinv", "q", "r")), lam)
    m["dz_bound_ratio"] = SB.ratio(dz_g, e_g, bd_g, Kd)
    m["dz_bound_ratio_oracle"] = SB.ratio(dz, e_o, bd_o, Kd)
    m["dz_inputs_ratio"] = SB.ratio(e_g, e_o, bd_o, Kd)
    m["dz_bound_over_dz"] = float((Kd * SB.U * bd_o.max(axis=1) / np.abs(e_o).max(axis=1)).max())   # how much of max|dz| the bound allows: the map's conditioning
    assert m["dz_bound_ratio"] <= 1.0 and m["dz_bound_ratio_oracle"] <= 1.0, m
    m["dz"] = _rows(dz_g, dz).max()
    m["q_res"] = _buf(nat.read("q").reshape(B, N, nx), orc.buf("q"))
    m["r_res"] = _buf(nat.read("r").reshape(B, N, nu), orc.buf("r"))
    if case in CHAOTIC:
        o64.set_lambda(lam)
        o64.compute_dz()
        m["dz_rows_oracle32_vs_f64"] = float(_rows(dz, o64.buf("dz")).max())
        assert m["dz"] <= max(1e-5, 2.0 * m["dz_rows_oracle32_vs_f64"]), m
    else:
        # (iiwa14 N = 128: 1.2e-5 on the worst of 256 rows -- R^-1 = 5e5 there; the bound above is the sharp statement, this one the familiar one)
        assert m["dz"] <= (1e-5 if plant == "indy7" else 2e-5) and m["q_res"] <= 1e-4 and m["r_res"] <= 1e-4, m
    # the 8 merits from the oracle's dz
    nat.write("dz", dz)
    nat.stage("merit8", xu, dt, xs, ref)
    m8 = orc.merit(xu, xs, ref, dt, num_alphas=8)
    m["merit8"] = _rows(nat.read("merit").reshape(B, 8), m8).max()
    assert m["merit8"] <= 1e-5, m
    # the decision, rho / drho and the new iterate from the oracle's merit table
    rho0, drho0 = orc.buf("rho"), orc.buf("drho")
    for s_w in (nat.write, orc.set_buf):
        s_w("merit", m8); s_w("merit_cur", m0); s_w("dz", dz); s_w("rho", rho0); s_w("drho", drho0)
    xg = nat.stage("line_search", xu, dt, xs, ref)
    xo = orc.line_search(xu)
    for name in ("step", "rho", "drho", "merit_cur"):
        np.testing.assert_array_equal(nat.read(name), orc.buf(name), err_msg=name)
    np.testing.assert_array_equal(xg, xo)
    step = orc.buf("step")
    assert (step > 0).any() and np.abs(xo - xu).max() > 0      # the decisions are not trivially equal: steps were taken ...
    m["steps_taken"], m["distinct_steps"] = int((step > 0).sum()), int(len(np.unique(step)))
    _report(test="stagewise_shard", case=case, shard=shard, plant=plant, N=N, B=B, **{("err_" + k) if len(k) <= 5 else k: v for k, v in m.items()})


@pytest.mark.parametrize("shard", range(1, 8))
def test_first_iteration_decisions_on_every_shard_of_c4(shard):
    """rows 1024 r .. 1024 r + 1023 of C4 through the whole first iteration at the workload's own settings (shard 0 = C2 above)"""
    plant, N, B, p, out = _run("C4", shard=shard, max_sqp_iters=1)
    _check("C4", "default_1it_shard%d" % shard, plant, N, B, p, out, pcg_counts=True, xu_scale=3.0)


# ---- the sweep's ill-conditioned stages, settled in float64 (round-5 review, weak 1) ------------------------------------------------------
# In fp32 the sweep's gamma, P^-1 and dz can only be held to "as noisy as the fp32 oracle" (above): cond(theta + rho I) ~ 4e9 and a 1e7-fold
# cancellation in dz_u put single precision's rounding at the size of the result on the worst rows (shards 2 / 3: the a-priori dz bound allows
# 87 % / 96 % of max|dz|).  Conditioning cannot touch the float64 build: the SAME kernel sources compiled with double (libgato_hip_f64.so) against
# the float64 oracle from IDENTICAL inputs leave cond x 1e-16 ~ 1e-7.  This is the check that says "the HIP formulas ARE the oracle's" on every
# trajectory / block of all 8 sweep shards at full size (iiwa14 N = 64, B = 512, per-trajectory rho over nine decades):
#   merit of the warm start, KKT blocks, (Q + rho I)^-1, R^-1, S (also block by block), the 8 merits from the oracle's dz   <= 1e-12 (measured <= 1.7e-14)
#   gamma and dz from the oracle's lambda, per trajectory; the KKT residuals                                              <= 1e-10 (measured <= 1e-12)
#   P^-1, EVERY (trajectory, knot) block, main and stair                                                                  <= 1e-8  (measured <= 1e-9)
#   step, rho, drho, merit_cur and the new iterate from the oracle's merit table                            bit for bit
# schur_linsys.cuh:121-128 (gamma), :150-164 + :213-260 (P^-1 and its stair blocks), :316-431 (dz).
def _blocks3(a, b, nx):
    """S / P^-1 in the reference's layout [B][N][nx][3 nx]: per (trajectory, knot, left | main | right) block max|a - b| / max|b| (zero blocks
    -- row 0's left, row N-1's right -- must be zero on both sides), max over all of them"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    worst = 0.0
    for j in range(3):
        aa, bb = a[..., j * nx:(j + 1) * nx], b[..., j * nx:(j + 1) * nx]
        den = np.abs(bb).max(axis=(2, 3))
        num = np.abs(aa - bb).max(axis=(2, 3))
        assert np.all(num[den == 0] == 0)
        worst = max(worst, float((num[den > 0] / den[den > 0]).max()))
    return worst


@pytest.mark.parametrize("shard", range(8))
def test_float64_build_stage_by_stage_on_every_sweep_shard(shard):
    from gato_amd._lib import NativeSolver
    from oracle.oracle import OracleSolver
    plant, N, B, pr, p, dt = _problem("C5", shard)
    nt = os.cpu_count() or 1
    nat = NativeSolver(plant, N, B, f64=True, dt=dt, **p)
    orc = OracleSolver(plant, N, B, dt=dt, threads=nt, f64=True, **p)
    for s in (nat, orc):
        s.set_rho_penalty_batch(pr["rho"])
    xu, xs, ref = pr["xu"], pr["x_s"], pr["ref"]
    nx, nu = nat.nx, nat.nu
    m = {}
    nat.stage("merit1", xu, dt, xs, ref)
    m0 = orc.merit(xu, xs, ref, dt, num_alphas=1, zero_dz=True)[:, 0]
    m["merit0"] = _rows(nat.read("merit_cur")[:, None], m0[:, None]).max()
    nat.stage("kkt", xu, dt, xs, ref)
    orc.setup_kkt(xu, xs, ref, dt)
    dk = nat.dense_kkt(dt)
    assert dk["A"].dtype == np.float64 and orc.buf("A").dtype == np.float64
    for name in ("A", "B", "R", "r"):
        m[name] = _buf(dk[name][:, :N - 1], orc.buf(name)[:, :N - 1])
    for name in ("c", "Q", "q"):
        m[name] = _buf(dk[name], orc.buf(name))
    nat.stage("schur", xu, dt, xs, ref)
    orc.form_schur()
    dk = nat.dense_kkt(dt)
    m["Qinv"] = _buf(dk["Qinv"], orc.buf("Qinv"))
    m["Rinv"] = _buf(dk["Rinv"][:, :N - 1], orc.buf("Rinv")[:, :N - 1])
    got = {name: nat.read(name).reshape(orc.buf(name).shape) for name in ("S", "Pinv", "gamma")}
    m["S"] = _buf(got["S"], orc.buf("S"))
    m["S_worst_block"] = _blocks3(got["S"], orc.buf("S"), nx)
    m["Pinv_worst_block"] = _blocks3(got["Pinv"], orc.buf("Pinv"), nx)          # every block: main diagonal and both stair off-diagonals of every knot
    m["gamma_rows"] = float(_rows(got["gamma"], orc.buf("gamma")).max())
    orc.pcg()
    lam = orc.buf("lambda")
    nat.write("lambda", lam)
    nat.stage("dz", xu, dt, xs, ref)
    orc.compute_dz()
    dz = orc.buf("dz")
    m["dz_rows"] = float(_rows(nat.read("dz").reshape(B, -1), dz).max())
    m["q_res"] = _buf(nat.read("q").reshape(B, N, nx), orc.buf("q"))
    m["r_res"] = _buf(nat.read("r").reshape(B, N, nu), orc.buf("r"))
    nat.write("dz", dz)
    nat.stage("merit8", xu, dt, xs, ref)
    m8 = orc.merit(xu, xs, ref, dt, num_alphas=8)
    m["merit8"] = _rows(nat.read("merit").reshape(B, 8), m8).max()
    rho0, drho0 = orc.buf("rho"), orc.buf("drho")
    for s_w in (nat.write, orc.set_buf):
        s_w("merit", m8); s_w("merit_cur", m0); s_w("dz", dz); s_w("rho", rho0); s_w("drho", drho0)
    xg = nat.stage("line_search", xu, dt, xs, ref)
    xo = orc.line_search(xu)
    step = orc.buf("step")
    m["steps_taken"], m["distinct_steps"] = int((step > 0).sum()), int(len(np.unique(step)))
    _report(test="stagewise_shard_f64", case="C5", shard=shard, plant=plant, N=N, B=B, rho_min=float(pr["rho"].min()), rho_max=float(pr["rho"].max()),
             **{("err_" + k) if k not in ("steps_taken", "distinct_steps") else k: v for k, v in m.items()})
    # measured on MI355X (profiles/r06_parity_full_size.json): everything well-conditioned <= 1.7e-14, gamma <= 8e-13, dz <= 1e-12, P^-1's worst block <= 1e-9
    for name in ("merit0", "A", "B", "c", "Q", "q", "R", "r", "Qinv", "Rinv", "S", "S_worst_block", "merit8"):
        assert m[name] <= 1e-12, (name, m)
    for name in ("gamma_rows", "dz_rows", "q_res", "r_res"):
        assert m[name] <= 1e-10, (name, m)
    assert m["Pinv_worst_block"] <= 1e-8, m      # cond(theta + rho I) ~ 4e9 x 2e-16, on the worst of 512 x 64 x 3 blocks
    for name in ("step", "rho", "drho", "merit_cur"):
        np.testing.assert_array_equal(nat.read(name), orc.buf(name), err_msg=name)
    np.testing.assert_array_equal(xg, xo)
    assert (step > 0).any() and np.abs(xo - xu).max() > 0
