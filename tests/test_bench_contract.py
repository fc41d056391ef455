"""bench.py's contract with the driver: ONE JSON line with the agreed keys, the value derived from the timed region, both baselines beside
it; no HIP device -> it refuses to run (there is no CPU path to fall back to)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_source_hash_names_the_kernel_sources():
    sys.path.insert(0, ROOT)
    import bench
    h = bench.source_hash()
    assert isinstance(h, str) and len(h) == 16 and int(h, 16) >= 0
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    if pmc["build"] != h:   # not an error of the code: bench.py then reports build_matches = false until the profile round is rerun
        pytest.skip("profiles/pmc_summary.json was measured on other kernel sources (%s): rerun tools/profile_round.sh + tools/summarize_pmc.py" % pmc["build"])


def test_refuses_to_run_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    r = _run("--steps", "1", "--warmup", "0", timeout=300)
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout) and not r.stdout.strip().startswith("{")


@pytest.mark.gpu
@pytest.mark.parametrize("extra,workload", [((), "fig8"), (("--workload", "hparam", "--plant", "iiwa14", "--knots", "16", "--batch", "64"), "hparam")])
def test_json_line(extra, workload):
    r = _run("--steps", "3", "--warmup", "1", "--cpu-sample", "8", *extra)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "trajectory-SQP-iterations/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    cfg = d["config"]
    assert workload in cfg["workload"] or cfg["workload"]
    assert "model" not in cfg
    B = 64 if workload == "hparam" else 1024
    iters = d["roofline"]["stage_us_per_solve"] and 10
    assert abs(d["value"] - B * iters / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]        # whole-job throughput from the timed region
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "valu", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    assert d.get("solution_ok") is True
    ps = d["parity_sample"]
    assert ps["rows"] >= 8 and ps["rows_on_the_oracles_step_or_a_tie"] >= ps["rows_required"] and ps["initial_merit_rel_err"] < 1e-5
