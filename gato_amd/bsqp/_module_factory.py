"""Builds the contents of the `bsqpN{N}_{plant}` modules: attribute KNOT_POINTS and the classes `BSQP_{B}_float`
(python/bindings.cu:224-264).  One run-time-B native solver backs every class; the batch sizes the reference registers are
listed eagerly and any other positive B resolves through the module's `__getattr__`."""
import re

import numpy as np

from .. import _lib
from .config import STANDARD_BATCH_SIZES

_ARG_ORDER = ["dt", "max_sqp_iters", "kkt_tol", "max_pcg_iters", "pcg_tol", "solve_ratio", "mu", "q_cost", "qd_cost", "u_cost", "N_cost",
              "q_lim_cost", "vel_lim_cost", "ctrl_lim_cost", "rho"]
_RESULT_KEYS = ["XU", "sqp_time_us", "sqp_iters", "kkt_converged", "final_merit", "initial_merit", "ls_num_iters", "pcg_times_us", "pcg_iters",
                "ls_min_merit", "ls_step_size"]


def make_class(plant, knot_points, batch_size):
    class _BSQP:
        """`PyBSQP<float, %d>` for %s, KNOT_POINTS = %d, on libgato_hip.so.""" % (batch_size, plant, knot_points)
        PLANT = plant
        KNOT_POINTS = knot_points
        BATCH_SIZE = batch_size

        def __init__(self, *args):
            # py::init<>() or py::init<T, uint32_t, T, uint32_t, T x 11>() (bindings.cu:226-227)
            if len(args) not in (0, len(_ARG_ORDER)):
                raise TypeError("__init__(): incompatible constructor arguments: expected 0 or %d, got %d" % (len(_ARG_ORDER), len(args)))
            kw = dict(zip(_ARG_ORDER, args))
            for k in ("max_sqp_iters", "max_pcg_iters"):
                if k in kw:
                    kw[k] = int(kw[k])
            self._s = _lib.NativeSolver(plant, knot_points, batch_size, **kw)

        def solve(self, xu, timestep, x_s, ref):
            out = self._s.solve(xu, timestep, x_s, ref)
            return {k: out[k] for k in _RESULT_KEYS}

        def reset_dual(self):
            self._s.reset_dual()

        def reset_rho(self):
            self._s.reset_rho()

        def set_f_ext_batch(self, f_ext_batch):
            self._s.set_f_ext_batch(f_ext_batch)

        def set_rho_penalty_batch(self, rho_batch, set_as_reset_default=True):
            self._s.set_rho_penalty_batch(rho_batch, set_as_reset_default)

        def set_drho_batch(self, drho_batch, set_as_reset_default=True):
            self._s.set_drho_batch(drho_batch, set_as_reset_default)

        def set_mu_batch(self, mu_batch):
            self._s.set_mu_batch(mu_batch)

        def set_cost_weights_batch(self, w):
            """extension: w[B,7] = q, qd, u, N, q_lim, vel_lim, ctrl_lim cost weights per trajectory"""
            self._s.set_cost_weights_batch(w)

        def set_pcg_tol_batch(self, pcg_tol_batch):
            self._s.set_pcg_tol_batch(pcg_tol_batch)

        def sim_forward(self, xk, uk, dt):
            return self._s.sim_forward(xk, uk, dt)

        def set_rho_adaptation(self, enabled):
            self._s.set_rho_adaptation(enabled)

        # not part of the reference surface: used by the facade's ee_pos (the reference goes through pinocchio there)
        def ee_pos(self, q):
            return self._s.ee_pos(np.asarray(q, np.float32))

    _BSQP.__name__ = _BSQP.__qualname__ = "BSQP_%d_float" % batch_size
    return _BSQP


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
        raise AttributeError("module %r has no attribute %r" % (namespace.get("__name__"), name))

    namespace["__getattr__"] = __getattr__
