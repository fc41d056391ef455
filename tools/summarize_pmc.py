#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/pmc_summary.json.

Units and corrections per MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B), so reads are doubled.  Our kernels mix 16-B, 8-B and
4-B per-lane accesses, for which the guide calls the counter uncalibrated: the doubled figure is an upper estimate of the read
side, the raw figure a lower one; both are recorded.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = [("pcg", r"pcgc?_kernel"), ("kkt", r"kkt_kernel"), ("schur", r"schur[2q]?_kernel|pinv_kernel"), ("merit", r"merit_kernel|step_kernel"), ("dz", r"dz_kernel"),
         ("line_search", r"line_search_kernel")]


def load(dirpat, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(dirpat, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def main(tag):
    out = {}
    fetch = load(os.path.join(ROOT, "gpurun_out", "pmc_%s_FETCH_SIZE" % tag), "FETCH_SIZE")
    write = load(os.path.join(ROOT, "gpurun_out", "pmc_%s_WRITE_SIZE" % tag), "WRITE_SIZE")
    n_iter = sum(v[1] for k, v in fetch.items() if re.search(r"kkt_kernel", k))  # one assembly launch per SQP iteration
    for stage, pat in STAGE:
        fk = sum(v[0] for k, v in fetch.items() if re.search(pat, k))
        fn = sum(v[1] for k, v in fetch.items() if re.search(pat, k))
        wk = sum(v[0] for k, v in write.items() if re.search(pat, k))
        wn = sum(v[1] for k, v in write.items() if re.search(pat, k))
        if not fn or not wn:
            continue
        # a "launch" of a family = everything it runs in one SQP iteration (the merit family: per dispatch, like bench.py's clock)
        launches_f = fn if stage == "merit" else n_iter
        launches_w = wn if stage == "merit" else sum(v[1] for k, v in write.items() if re.search(r"kkt_kernel", k))
        per_f = fk / launches_f * 1024.0
        per_w = wk / launches_w * 1024.0
        out[stage] = {"fetch_bytes_raw_per_launch": per_f, "write_bytes_per_launch": per_w,
                      "hbm_bytes_per_launch": 2.0 * per_f + per_w, "hbm_bytes_per_launch_lower": per_f + per_w, "dispatches": fn}
    out["_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, read side doubled per the gfx950 correction; bench.py at C2 (indy7 N=32 B=1024)"
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01b")
