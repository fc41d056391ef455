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
Measured numbers go to gpurun_out/r04_parity_full_size.jsonl (copied to profiles/r04_parity_full_size.json)."""
import json
import os

import numpy as np
import pytest

from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem, hparam_problem

pytestmark = pytest.mark.gpu

CASES = {"C2": ("indy7", 32, 1024, "fig8"), "C3": ("iiwa14", 128, 256, "fig8"), "C5": ("iiwa14", 64, 512, "hparam")}
# per-trajectory iterate error HIP vs fp32 oracle over the resolved rows: (max, 99th percentile, median).  Measured (profiles/r04_parity_full_size.json):
# C2 4.7e-4 / 1.9e-4 / 3.1e-5, C3 7.3e-4 / 4.1e-4 / 8.3e-5 at PCG's floor -- the north-star's 1e-4 holds in the median and up to the ~95th
# percentile; the tail is cond(S) ~ 1e9 .. 1e10 times fp32 rounding of S and gamma (tools/stage_errors.py: no single stage carries it)
XU_BOUND = {"indy7": (1e-3, 4e-4, 1e-4), "iiwa14": (1.5e-3, 8e-4, 2e-4)}
TIE = 2e-2   # two candidate merits closer than this (relative) are one decision to fp32: 1e-4 in XU is ~1e-2 in the merit (mu |defect|_1 through M^-1)
MIN_RESOLVED = {"C2": 0.97, "C3": 0.97, "C5": 0.0}
MAX_DEPART = {"C2": 0.0, "C3": 0.0, "C5": 0.05}   # resolved rows that leave the fp32 oracle's step without a tie
CHAOTIC = {"C5"}   # no resolved rows to speak of: only the statistical block is asserted, with the margins of a regime where fp32 itself is noise


def _report(**kw):
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "r04_parity_full_size.jsonl"), "a") as f:
            f.write(json.dumps({k: (v if isinstance(v, (str, bool, int, list)) else float(v)) for k, v in kw.items()}) + "\n")
    except OSError:
        pass


def _err(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max(axis=1) / np.maximum(1.0, np.abs(np.asarray(b, np.float64)).max(axis=1))


def _run(case, **over):
    from gato_amd._lib import NativeSolver
    from oracle.oracle import OracleSolver
    plant, N, B, kind = CASES[case]
    if kind == "hparam":
        pr = hparam_problem(plant, N, B, shard=3)
        p, dt = dict(pr["params"]), pr["dt"]
    else:
        pr = fig8_problem(plant, N, B)
        p, dt = dict(DEFAULT_SOLVER_PARAMS), 0.01
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
    mx, p99, med = (float(ego[use].max()), float(np.quantile(ego[use], 0.99)), float(np.median(ego[use]))) if enough else (0.0, 0.0, 0.0)
    dp = np.abs(g["pcg_iters"][0].astype(int) - o32["pcg_iters"][0].astype(int))
    _report(test=tag, case=case, plant=plant, N=N, B=B, resolved_rows=int(resolved.sum()), steps_equal_on_resolved=int(use.sum()), near_ties=ties, departures=len(departs),
            xu_vs_fp32_oracle_max=mx, xu_vs_fp32_oracle_p99=p99, xu_vs_fp32_oracle_median=med,
            all_rows_median_dist_to_f64_hip=float(np.median(eg)), all_rows_median_dist_to_f64_oracle32=float(np.median(e32)),
            rows_on_f64_steps_hip=int((sg == s64).sum()), rows_on_f64_steps_oracle32=int((s32 == s64).sum()),
            rows_on_fp32_oracle_steps=int(same.sum()), pcg_equal_on_resolved=int((dp[resolved] == 0).sum()), pcg_within_1_on_resolved=int((dp[resolved] <= 1).sum()), pcg_max=int(max(g["pcg_iters"][0].max(), o32["pcg_iters"][0].max())))
    assert resolved.mean() >= MIN_RESOLVED[case], resolved.sum()
    assert ties <= max(1, 0.02 * resolved.sum()), (ties, resolved.sum())
    bmx, b99, bmed = (xu_scale * v for v in XU_BOUND[plant])
    assert mx <= bmx and p99 <= b99 and med <= bmed, (mx, p99, med)
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
