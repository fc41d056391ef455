#!/usr/bin/env python3
"""Carry tools/microbench/*.patch (experiments that were measured and taken out of the product) over a change of the kernel sources: each patch is applied
to the sources of the commit it still applies to (`--base`, default HEAD), merged three-way with the working tree (git merge-file) and re-diffed against
it.  Conflicts are left for the caller (exit code 1, the merged files stay under the printed directory); tests/test_abi.py checks that every patch applies.

    python tools/refresh_patches.py [--base HEAD] pcgs_split pcg_phase_trace
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("gato_amd/csrc/kernels.hpp", "gato_amd/csrc/solver.hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", default="HEAD")
    ap.add_argument("patches", nargs="+")
    a = ap.parse_args()
    rc = 0
    for name in a.patches:
        patch = os.path.join(ROOT, "tools", "microbench", name + ".patch")
        d = tempfile.mkdtemp(prefix="patch_%s_" % name)
        for sub in ("base", "pat", "a", "b"):
            os.makedirs(os.path.join(d, sub, "gato_amd", "csrc"))
        for f in FILES:
            src = subprocess.run(["git", "show", "%s:%s" % (a.base, f)], cwd=ROOT, capture_output=True, text=True, check=True).stdout
            for sub in ("base", "pat"):
                open(os.path.join(d, sub, f), "w").write(src)
            shutil.copy(os.path.join(ROOT, f), os.path.join(d, "a", f))
        r = subprocess.run(["patch", "-p1", "--batch", "-s", "-i", patch], cwd=os.path.join(d, "pat"), capture_output=True, text=True)
        if r.returncode != 0:
            print("%s does not apply to %s: %s" % (name, a.base, r.stdout[-400:]))
            rc = 1
            continue
        conflicts = 0
        for f in FILES:
            m = os.path.join(d, "b", f)
            shutil.copy(os.path.join(ROOT, f), m)
            conflicts += subprocess.run(["git", "merge-file", "-q", m, os.path.join(d, "base", f), os.path.join(d, "pat", f)]).returncode
        if conflicts:
            print("%s: %d conflict(s), resolve in %s/b and re-diff (diff -u -r a b)" % (name, conflicts, d))
            rc = 1
            continue
        out = subprocess.run(["diff", "-u", "-r", "a", "b"], cwd=d, capture_output=True, text=True).stdout
        open(patch, "w").write(out)
        print("%s refreshed (%d lines)" % (name, len(out.splitlines())))
    sys.exit(rc)


if __name__ == "__main__":
    main()
