#!/usr/bin/env python3
"""Static instruction mix per device function from `hipcc -S --cuda-device-only` output (straight-line lane kernels: static == dynamic)."""
import collections
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
cur = None
stats = {}
for line in txt:
    m = re.match(r"^(_ZN4gato\S*):", line)
    if m:
        cur = m.group(1)
        stats[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    s = line.strip()
    if s.startswith(".Lfunc_end"):
        cur = None
        continue
    m = re.match(r"([a-z_0-9]+)(\s|$)", s)
    if not m:
        continue
    op = m.group(1)
    c = stats[cur]
    if op.startswith("v_accvgpr"):
        c["accvgpr"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
    elif op.startswith("scratch_"):
        c["scratch"] += 1
    elif op.startswith("global_") or op.startswith("buffer_"):
        c["vmem"] += 1
    elif op.startswith("s_waitcnt"):
        c["wait"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
for k, v in stats.items():
    if pat.search(k):
        print(k[:70], dict(v))
