"""Host-side mirror of the reference's `python/bsqp` package for the batched-SQP path.

`install_as_bsqp()` registers this package under the top-level name `bsqp`, so the reference's own examples
(`from bsqp.interface import BSQP`, `importlib.import_module("bsqp.bsqpN32_indy7")`) run unchanged on the MI355X library.
"""
import sys


def install_as_bsqp():
    import importlib
    me = sys.modules[__name__]
    sys.modules.setdefault("bsqp", me)
    for sub in ("interface", "common", "config", "workloads"):
        sys.modules.setdefault("bsqp." + sub, importlib.import_module(__name__ + "." + sub))
    return me
