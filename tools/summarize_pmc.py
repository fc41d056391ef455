#!/usr/bin/env python3
"""Per-kernel summary of one profiling round (tools/profile_round.sh <tag>) -> profiles/pmc_summary.json + profiles/<tag>_*_kernel_stats.csv.

For every configuration (c2 = the bench line's workload, c3 = iiwa14 N=128 B=256, c5 = iiwa14 N=64 B=512 sweep shard) and every
kernel: dispatches, average duration (rocprofv3 --kernel-trace --stats), HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes)
and the SQ counters per dispatch, plus a few ratios a reader can recompute from the raw numbers next to them.

Units and corrections per MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B), so the read side is doubled for `hbm_bytes`.  These
kernels mix 16-, 8- and 4-byte per-lane accesses, for which the guide calls the counter uncalibrated: the doubled figure is an upper
estimate of the read side, the raw one a lower estimate; both are recorded.

The file carries the hash of the kernel sources it was measured on (`build`), which bench.py compares with the sources it runs.
"""
import csv
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_HZ = 2.4e9          # MI355X peak engine clock (MI355X_MICROARCH.md, chip-level parameters)
SIMDS = 256 * 4
CONFIGS = ("c2", "c3", "c5", "cr")   # cr: direct mode, indy7 N=128 B=8 (block cyclic reduction)
SRC = ["gato_amd/csrc/kernels.hpp", "gato_amd/csrc/rbd.hpp", "gato_amd/csrc/solver.hip", "gato_amd/csrc/robot_models.hpp"]


def build_hash():
    sys.path.insert(0, ROOT)
    from tools.source_hash import source_hash   # the one definition (also baked into the libraries and printed by bench.py)
    return source_hash()


def short(name):
    m = re.match(r"(?:void )?(?:gato::)?([A-Za-z0-9_]+)(<[^(]*>)?\(", name)
    if not m:
        return name.split("(")[0]
    return m.group(1) + (m.group(2) or "").replace("gato::", "")


def counters(dirpath):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    f = os.path.join(dirpath, "p_counter_collection.csv")
    if not os.path.exists(f):
        return acc
    for row in csv.DictReader(open(f)):
        a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
    return acc


def main(tag):
    out = {"build": build_hash(), "tag": tag,
           "git_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
           "_note": "rocprofv3 --kernel-trace --stats (avg_us) and --pmc passes (separate runs); FETCH/WRITE in bytes per dispatch (KiB x 1024), "
                    "hbm_bytes = 2 x fetch + write (gfx950 read-side correction, upper estimate), hbm_bytes_lower = fetch + write; SQ counters "
                    "are sums over the chip per dispatch; valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (avg duration x 2.4 GHz x 1024 SIMDs)"}
    for cfg in CONFIGS:
        stats_f = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, cfg), "k_kernel_stats.csv")
        if not os.path.exists(stats_f):
            continue
        shutil.copy(stats_f, os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, cfg)))
        kern = {}
        for row in csv.DictReader(open(stats_f)):
            n = short(row["Name"])
            if n.startswith("__amd") or "at::native" in row["Name"]:
                continue
            kern[n] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "min_us": float(row["MinNs"]) / 1e3,
                       "max_us": float(row["MaxNs"]) / 1e3, "pct_of_gpu_time": float(row["Percentage"])}
        i = 0
        while os.path.isdir(os.path.join(ROOT, "gpurun_out", "pmc_%s_%s_%d" % (tag, cfg, i))):
            for k, cs in counters(os.path.join(ROOT, "gpurun_out", "pmc_%s_%s_%d" % (tag, cfg, i))).items():
                if k not in kern:
                    continue
                for cname, (tot, n) in cs.items():
                    kern[k][cname] = tot / n
            i += 1
        for k, d in kern.items():
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                d["fetch_bytes_raw"] = d.pop("FETCH_SIZE") * 1024.0
                d["write_bytes"] = d.pop("WRITE_SIZE") * 1024.0
                d["hbm_bytes"] = 2.0 * d["fetch_bytes_raw"] + d["write_bytes"]
                d["hbm_bytes_lower"] = d["fetch_bytes_raw"] + d["write_bytes"]
                d["hbm_GBps_upper"] = d["hbm_bytes"] / (d["avg_us"] * 1e-6) / 1e9
            if "SQ_INSTS_VALU" in d:
                d["valu_issue_frac"] = d["SQ_INSTS_VALU"] * 4.0 / (d["avg_us"] * 1e-6 * CLOCK_HZ * SIMDS)
                if d.get("SQ_WAVES"):
                    d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
            if d.get("SQ_LDS_IDX_ACTIVE"):
                d["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
                d["mfma_busy_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]   # no MFMA anywhere in this path: must read 0
        out[cfg] = kern
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    for cfg in CONFIGS:
        if cfg not in out:
            continue
        print(cfg)
        for k, d in sorted(out[cfg].items(), key=lambda kv: -kv[1]["pct_of_gpu_time"]):
            print("  %-44s %3d x %8.1f us  %5.1f%%  hbm %6.1f MB (%.0f GB/s)  valu_issue %.2f  lds_conflict %.2f  mfma_busy %s" % (
                k[:44], d["calls"], d["avg_us"], d["pct_of_gpu_time"], d.get("hbm_bytes", 0) / 1e6, d.get("hbm_GBps_upper", 0),
                d.get("valu_issue_frac", 0), d.get("lds_bank_conflict_frac", 0), d.get("mfma_busy_cycles", "-")))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02a")
