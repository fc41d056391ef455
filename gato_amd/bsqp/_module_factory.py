"""Builds the contents of the `bsqpN{N}_{plant}` modules: attribute KNOT_POINTS and the classes `BSQP_{B}_float`
(python/bindings.cu:222-266).  The classes are subclasses of the COMPILED pybind11 class `gato_amd._gato_ext.BSQP` (csrc/pyext.cpp,
over the C ABI of libgato_hip.so) that fix plant, horizon and batch size -- the three the reference bakes into a module / a class
at compile time and this library takes at run time.  The batch sizes the reference registers are listed eagerly; any other
positive B resolves through the module's `__getattr__`.  There is no fallback: without the built extension the import raises.
`BSQP_{B}_double`, B in 1..128 -- what a USE_DOUBLES build of the reference registers instead (bindings.cu:244-252) -- resolve the same
way to subclasses of `gato_amd._gato_ext_f64.BSQP`, the binding of the float64 build libgato_hip_f64.so."""
import importlib
import re
import sys

from .. import _lib
from .config import STANDARD_BATCH_SIZES

_NARGS = 15  # dt, max_sqp_iters, kkt_tol, max_pcg_iters, pcg_tol, solve_ratio, mu, q/qd/u/N/q_lim/vel_lim/ctrl_lim cost, rho (bindings.cu:35-56)
_ext = None
_ext64 = None
DOUBLE_BATCH_SIZES = (1, 2, 4, 8, 16, 32, 64, 128)   # bindings.cu:245-252


def load_ext():
    """`gato_amd._gato_ext`.  torch (when installed) is imported first so the extension shares torch's HIP runtime, like _lib.load()."""
    global _ext
    if _ext is None:
        _lib.preload_torch()
        try:
            _ext = importlib.import_module("gato_amd._gato_ext")
        except ImportError as e:
            raise _lib.GatoError("the pybind11 extension gato_amd/_gato_ext*.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                                 "g.build()'` or `make -C gato_amd/csrc`." % e)
    return _ext


def load_ext_f64():
    global _ext64
    if _ext64 is None:
        _lib.preload_torch()
        try:
            _ext64 = importlib.import_module("gato_amd._gato_ext_f64")
        except ImportError as e:
            raise _lib.GatoError("the float64 extension gato_amd/_gato_ext_f64*.so is not built (%s). Run `make -C gato_amd/csrc`." % e)
    return _ext64


def make_class(plant, knot_points, batch_size, double=False):
    ext = load_ext_f64() if double else load_ext()

    def __init__(self, *args):
        # py::init<>() or py::init<T, uint32_t, T, uint32_t, T x 11>() (bindings.cu:226-227)
        if len(args) not in (0, _NARGS):
            raise TypeError("__init__(): incompatible constructor arguments: expected 0 or %d, got %d" % (_NARGS, len(args)))
        ext.BSQP.__init__(self, plant, knot_points, batch_size, *args)

    tname, lib = ("double", "libgato_hip_f64.so") if double else ("float", "libgato_hip.so")
    return type("BSQP_%d_%s" % (batch_size, tname), (ext.BSQP,), {
        "__init__": __init__, "__doc__": "`PyBSQP<%s, %d>` for %s, KNOT_POINTS = %d, on %s." % (tname, batch_size, plant, knot_points, lib),
        "PLANT": plant, "KNOT_POINTS": knot_points, "BATCH_SIZE": batch_size})


def populate(namespace, plant, knot_points):
    namespace["KNOT_POINTS"] = knot_points
    for b in STANDARD_BATCH_SIZES:
        namespace["BSQP_%d_float" % b] = make_class(plant, knot_points, b)

    def __getattr__(name):
        m = re.fullmatch(r"BSQP_(\d+)_float", name)
        if m and int(m.group(1)) >= 1:
            cls = make_class(plant, knot_points, int(m.group(1)))
            namespace[name] = cls
            return cls
        m = re.fullmatch(r"BSQP_(\d+)_double", name)
        if m and int(m.group(1)) in DOUBLE_BATCH_SIZES:
            cls = make_class(plant, knot_points, int(m.group(1)), double=True)
            namespace[name] = cls
            return cls
        raise AttributeError("module %r has no attribute %r" % (namespace.get("__name__"), name))

    namespace["__getattr__"] = __getattr__
