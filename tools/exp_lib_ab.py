#!/usr/bin/env python3
"""A / B of two builds of the fp32 library on the headline workload: same bits (sha256 of the iterates, duals and PCG counts of a 10-iteration C2 solve)
and the driver-style bench value (bench.py --steps 200, best and median of `--runs`).  The experimental build is loaded through GATO_HIP_LIB.

    make -C gato_amd/csrc libgato_hip_exp.so EXPFLAGS=-DGATO_EXP_PARTS2
    python tools/exp_lib_ab.py --exp gato_amd/csrc/libgato_hip_exp.so [--runs 5]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DIGEST = r'''
import hashlib, json, sys
sys.path.insert(0, %r)
import numpy as np
from gato_amd._lib import NativeSolver
from gato_amd.bsqp.config import DEFAULT_SOLVER_PARAMS
from gato_amd.bsqp.workloads import fig8_problem
import os
plant, N, B = os.environ.get("AB_PLANT", "indy7"), int(os.environ.get("AB_KNOTS", "32")), int(os.environ.get("AB_BATCH", "1024"))
pr = fig8_problem(plant, N, B)
s = NativeSolver(plant, N, B, dt=0.01, **dict(DEFAULT_SOLVER_PARAMS, max_sqp_iters=10))
r = s.solve(pr["xu"], 0.01, pr["x_s"], pr["ref"])
h = hashlib.sha256()
for a in (r["XU"], s.read("lambda"), r["pcg_iters_all"], r["final_merit"], r["ls_step_size"]):
    h.update(np.ascontiguousarray(a).tobytes())
print(json.dumps({"digest": h.hexdigest()[:16], "sum_max_pcg": int(r["pcg_iters_all"].max(axis=1).sum())}))
''' % ROOT


def run(env, args):
    r = subprocess.run(args, cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(r.stderr[-1500:])
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp", action="append", required=True, help="[name=]path of an experimental build; repeatable")
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_c2_chain.json"))
    ap.add_argument("--plant", default="indy7")
    ap.add_argument("--knots", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1024)
    a = ap.parse_args()
    os.environ.update(AB_PLANT=a.plant, AB_KNOTS=str(a.knots), AB_BATCH=str(a.batch))
    cfg = ["--plant", a.plant, "--knots", str(a.knots), "--batch", str(a.batch)]
    libs = [("base", None)]
    for e in a.exp:
        name, _, path = e.rpartition("=")
        libs.append((name or "exp", os.path.abspath(path)))
    out = {}
    envs = {}
    for name, lib in libs:
        env = dict(os.environ)
        if lib:
            env["GATO_HIP_LIB"] = lib
        envs[name] = env
        out[name] = {"library": lib or "gato_amd/csrc/libgato_hip.so", **run(env, [sys.executable, "-c", DIGEST]), "bench_values": [], "ms_per_solve": [], "pcg_launch_us": []}
    for _ in range(a.runs):                 # round robin: a drift of the box (clock, temperature) hits every build alike
        for name, _lib in libs:
            b = run(envs[name], [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "10", "--no-cpu-baseline", *cfg])
            out[name]["bench_values"].append(b["value"]); out[name]["ms_per_solve"].append(b["ms_per_step"]); out[name]["pcg_launch_us"].append(b["roofline"]["avg_launch_us"])
    for name, _lib in libs:
        v = sorted(out[name]["bench_values"])
        out[name]["best_value"], out[name]["median_value"] = v[-1], v[len(v) // 2]
        print(name, json.dumps(out[name]), flush=True)
    summary = {}
    for name, _lib in libs[1:]:
        summary[name] = {"same_bits": out[name]["digest"] == out["base"]["digest"], "median_gain": out[name]["median_value"] / out["base"]["median_value"] - 1.0,
                         "best_gain": out[name]["best_value"] / out["base"]["best_value"] - 1.0}
    out["vs_base"] = summary
    print(json.dumps(summary))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
