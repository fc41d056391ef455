#!/usr/bin/env python3
"""sha256 (first 16 hex digits) of the kernel sources, in a fixed order.  ONE definition for three users: gato_amd/csrc/Makefile bakes it into the
libraries (-DGATO_SRC_HASH, read back through gato_source_hash() / gato_version()), bench.py puts both -- the loaded library's and the tree's --
on its JSON line, tools/summarize_pmc.py stamps profiles/pmc_summary.json with it."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("gato_amd/csrc/kernels.hpp", "gato_amd/csrc/rbd.hpp", "gato_amd/csrc/solver.hip", "gato_amd/csrc/robot_models.hpp")


def source_hash(root=ROOT):
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(root, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
