"""gato_amd -- MI355X-native batched SQP trajectory optimizer (drop-in for A2R-Lab/GATO's bsqp path).

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C-ABI library libgato_hip.so), `_lib` (ctypes binding
of include/gato_abi.h), `bsqp/` (host-side mirror of the reference's python/bsqp package: `BSQP` facade, the
`bsqpN{N}_{plant}` modules with their `BSQP_{B}_float` classes, fig-8 / warm-start helpers, default parameter sets).
"""
__version__ = "0.1.0"
