"""ctypes binding of include/gato_abi.h (libgato_hip.so).  Fails loudly when the HIP library is missing: there is no CPU fallback."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GATO_HIP_LIB") or os.path.join(HERE, "csrc", "libgato_hip.so")   # GATO_HIP_LIB: an experimental build (tools/)
LIB_PATH_F64 = os.path.join(HERE, "csrc", "libgato_hip_f64.so")   # the USE_DOUBLES build (gato/settings.h:7-11): same entry points on double
PLANTS = {"indy7": 0, "iiwa14": 1}
NQ = {"indy7": 6, "iiwa14": 7}

SYMBOLS = [
    "gato_default_params", "gato_dims", "gato_create", "gato_destroy", "gato_solve", "gato_solve_device", "gato_get_counts",
    "gato_get_sqp_iters", "gato_get_kkt_converged", "gato_get_final_merit", "gato_get_initial_merit", "gato_get_pcg_iters",
    "gato_get_ls_min_merit", "gato_get_ls_step_size", "gato_set_f_ext_batch", "gato_set_rho_penalty_batch", "gato_set_drho_batch",
    "gato_set_mu_batch", "gato_set_pcg_tol_batch", "gato_reset_dual", "gato_reset_rho", "gato_set_rho_adaptation", "gato_sim_forward",
    "gato_ee_pos", "gato_debug_read", "gato_debug_write", "gato_debug_stage", "gato_set_profiling", "gato_get_stage_times_us",
    "gato_last_error", "gato_version", "gato_reset_async", "gato_copy_final_merit_device", "gato_set_cost_weights_batch",
    "gato_synchronize", "gato_sim_forward_device", "gato_select_best", "gato_select_best_device",
    "gato_plant_rk4", "gato_fk_placements", "gato_set_linear_solver", "gato_set_graph_mode",
    "gato_mpc_begin", "gato_mpc_step", "gato_mpc_get_best", "gato_plant_payload_rk4", "gato_mpc_set_payload", "gato_mpc_get_payload",
    "gato_comm_unique_id", "gato_comm_init", "gato_comm_destroy", "gato_gather_results", "gato_debug_set_remote_solved", "gato_abi_real_size",
    "gato_set_solved_count_mode", "gato_get_shard_stats", "gato_comm_available", "gato_abi_version",
    "gato_comm_init_rank", "gato_comm_confirm", "gato_get_solved_count_state", "gato_source_hash",
]
ABI_VERSION = 6   # GATO_ABI_VERSION of the include/gato_abi.h this binding was written against


def _params_struct(ft, name):
    return type(name, (C.Structure,), {"_fields_": [
        ("dt", ft), ("max_sqp_iters", C.c_uint32), ("kkt_tol", ft), ("max_pcg_iters", C.c_uint32), ("pcg_tol", ft), ("solve_ratio", ft),
        ("mu", ft), ("q_cost", ft), ("qd_cost", ft), ("u_cost", ft), ("N_cost", ft), ("q_lim_cost", ft), ("vel_lim_cost", ft),
        ("ctrl_lim_cost", ft), ("rho", ft)]})


def _mpc_struct(ft, name):
    """GatoMpcStep of include/gato_abi.h"""
    return type(name, (C.Structure,), {"_fields_": [
        ("struct_size", C.c_uint32), ("phases", C.c_int32), ("plant_steps", C.c_int32), ("sim_dt", ft), ("steps_per_knot", C.c_double), ("plant_wrench", ft * 6),
        ("ref_window", C.POINTER(ft)), ("hyp_world", C.POINTER(ft)), ("select", C.c_int32), ("select_dt", ft), ("x", ft * 16), ("ee", ft * 3),
        ("best", C.c_int32), ("solve_us", C.c_double), ("errors", C.POINTER(ft)), ("solve_wall_us", C.c_double), ("plant_us", C.c_double)]})


GatoParams = _params_struct(C.c_float, "GatoParams")
GatoParamsF64 = _params_struct(C.c_double, "GatoParamsF64")


class GatoError(RuntimeError):
    pass


_lib = None
_libs = {}


def preload_torch():
    """torch (when installed) is imported BEFORE libgato_hip.so so that both share torch's bundled libamdhip64 instead of loading a
    second HIP runtime into the process."""
    if "torch" not in sys.modules and os.environ.get("GATO_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def load(f64=False):
    """Loads libgato_hip.so (or its float64 build) for the ctypes binding (the tests' back door to the stage / debug entry points; the
    product's Python classes are the compiled ones of gato_amd._gato_ext)."""
    global _lib
    f64 = bool(f64)
    if f64 in _libs:
        return _libs[f64]
    path = LIB_PATH_F64 if f64 else LIB_PATH
    if not os.path.exists(path):
        raise GatoError("%s is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C gato_amd/csrc`."
                        % (os.path.basename(path), path))
    preload_torch()
    L = C.CDLL(path)
    ft = C.c_double if f64 else C.c_float
    PT = GatoParamsF64 if f64 else GatoParams
    L._ft, L._np, L._PT = ft, (np.float64 if f64 else np.float32), PT
    L._MPC = _mpc_struct(ft, "GatoMpcStepF64" if f64 else "GatoMpcStep")
    fp, ip, vp = C.POINTER(ft), C.POINTER(C.c_int32), C.c_void_p
    L.gato_mpc_begin.argtypes = [vp, fp]
    L.gato_mpc_step.argtypes = [vp, C.POINTER(L._MPC)]
    L.gato_mpc_get_best.argtypes = [vp, fp]
    L.gato_comm_unique_id.argtypes = [C.c_char_p]
    L.gato_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.c_int64]
    L.gato_comm_destroy.argtypes = [vp]
    L.gato_gather_results.argtypes = [vp, vp, vp, C.c_uint64, vp]
    L.gato_debug_set_remote_solved.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int, C.c_int64]
    L.gato_set_solved_count_mode.argtypes = [vp, C.c_int]
    L.gato_get_shard_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.gato_get_solved_count_state.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.gato_comm_init_rank.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.c_int64]
    L.gato_comm_confirm.argtypes = [vp]
    L.gato_source_hash.restype = C.c_char_p
    L.gato_default_params.argtypes = [C.POINTER(PT)]
    L.gato_default_params.restype = None
    L.gato_dims.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gato_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(PT), C.POINTER(vp)]
    L.gato_destroy.argtypes = [vp]
    L.gato_solve.argtypes = [vp, fp, ft, fp, fp, C.POINTER(C.c_double)]
    L.gato_solve_device.argtypes = [vp, vp, ft, vp, vp, vp]
    L.gato_get_counts.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    for n in ("gato_get_sqp_iters", "gato_get_kkt_converged", "gato_get_pcg_iters"):
        getattr(L, n).argtypes = [vp, ip]
    for n in ("gato_get_final_merit", "gato_get_initial_merit", "gato_get_ls_min_merit", "gato_get_ls_step_size", "gato_set_f_ext_batch",
              "gato_set_mu_batch", "gato_set_pcg_tol_batch", "gato_set_cost_weights_batch"):
        getattr(L, n).argtypes = [vp, fp]
    for n in ("gato_set_rho_penalty_batch", "gato_set_drho_batch"):
        getattr(L, n).argtypes = [vp, fp, C.c_int]
    for n in ("gato_reset_dual", "gato_reset_rho"):
        getattr(L, n).argtypes = [vp]
    L.gato_set_rho_adaptation.argtypes = [vp, C.c_int]
    L.gato_set_linear_solver.argtypes = [vp, C.c_int]
    L.gato_set_graph_mode.argtypes = [vp, C.c_int]
    L.gato_sim_forward.argtypes = [vp, fp, fp, fp, ft]
    L.gato_ee_pos.argtypes = [vp, fp, C.c_int, fp]
    L.gato_debug_read.argtypes = [vp, C.c_char_p, fp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.gato_debug_write.argtypes = [vp, C.c_char_p, fp, C.c_uint64]
    L.gato_debug_stage.argtypes = [vp, C.c_int, fp, ft, fp, fp, fp]
    L.gato_set_profiling.argtypes = [vp, C.c_int]
    L.gato_get_stage_times_us.argtypes = [vp, C.POINTER(C.c_double)]
    L.gato_reset_async.argtypes = [vp, C.c_int, C.c_int, vp]
    L.gato_copy_final_merit_device.argtypes = [vp, vp, vp]
    L.gato_synchronize.argtypes = [vp]
    L.gato_plant_rk4.argtypes = [vp, fp, fp, C.c_int, fp, ft]
    L.gato_plant_payload_rk4.argtypes = [vp, fp, fp, fp, C.c_int, fp, ft]
    L.gato_mpc_set_payload.argtypes = [vp, fp]
    L.gato_mpc_get_payload.argtypes = [vp, fp]
    L.gato_fk_placements.argtypes = [C.c_int, fp, C.POINTER(C.c_double)]
    L.gato_select_best.argtypes = [vp, fp, fp, fp, ft, C.POINTER(C.c_int), fp]
    L.gato_select_best_device.argtypes = [vp, vp, vp, vp, ft, vp, vp, vp]
    L.gato_sim_forward_device.argtypes = [vp, vp, vp, vp, ft, vp]
    L.gato_last_error.restype = C.c_char_p
    L.gato_version.restype = C.c_char_p
    if L.gato_abi_version() != ABI_VERSION:
        raise GatoError("%s has ABI version %d, this binding expects %d (rebuild: make -C gato_amd/csrc)" % (path, L.gato_abi_version(), ABI_VERSION))
    if L.gato_abi_real_size() != C.sizeof(ft):
        raise GatoError("%s carries %d-byte reals, the binding expects %d" % (path, L.gato_abi_real_size(), C.sizeof(ft)))
    _libs[f64] = L
    if not f64:
        _lib = L
    return L


def _chk(rc, L=None):
    if rc != 0:
        raise GatoError("libgato_hip: status %d: %s" % (rc, (L or load()).gato_last_error().decode()))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def fk_placements(plant, q):
    """(R[nq,3,3], p[nq,3]) float64: world placements of the joint frames from the library's kinematic tables (no device needed)"""
    nq = NQ[plant]
    out = np.zeros((nq, 12), np.float64)
    _chk(load().gato_fk_placements(PLANTS[plant], _p(_f32(q, (nq,))), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out[:, :9].reshape(nq, 3, 3).copy(), out[:, 9:].copy()


PARAM_ORDER = [f[0] for f in GatoParams._fields_]
STAGES = {"merit8": 0, "kkt": 1, "schur": 2, "pcg": 3, "dz": 4, "line_search": 5, "merit1": 6, "direct": 7}


class NativeSolver:
    """Owns one `GatoSolver*`.  Method names follow PyBSQP<T,B> (python/bindings.cu:224-237)."""

    def __init__(self, plant, knot_points, batch_size, f64=False, **params):
        """f64: the USE_DOUBLES build (libgato_hip_f64.so, validation mode: stand-alone kernels, double buffers)"""
        L = self.L = load(f64)
        self.dtype, self._ft = L._np, L._ft
        if plant not in PLANTS:
            raise ValueError("unknown plant %r" % (plant,))
        self.plant, self.N, self.B = plant, int(knot_points), int(batch_size)
        self.nq = NQ[plant]
        self.nx, self.nu = 2 * self.nq, self.nq
        self.traj = (self.nx + self.nu) * self.N - self.nu
        p = L._PT()
        L.gato_default_params(C.byref(p))
        for k, v in params.items():
            if k not in PARAM_ORDER:
                raise TypeError("unknown solver parameter %r" % k)
            setattr(p, k, v)
        self.params = p
        h = C.c_void_p()
        self._chk(L.gato_create(PLANTS[plant], self.N, self.B, C.byref(p), C.byref(h)))
        self.h = h

    def _chk(self, rc):
        _chk(rc, self.L)

    def _f(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        return a if shape is None else a.reshape(shape)

    def _p(self, a):
        return a.ctypes.data_as(C.POINTER(self._ft))

    def close(self):
        if getattr(self, "h", None):
            self.L.gato_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- solve ----
    def solve(self, xu, timestep, x_s, ref):
        """PyBSQP::solve (bindings.cu:68-148): returns the same dict (XU, sqp_time_us, sqp_iters, kkt_converged, final_merit,
        initial_merit, ls_num_iters, pcg_times_us, pcg_iters, ls_min_merit, ls_step_size)."""
        L = self.L
        xu = np.array(xu, dtype=self.dtype, order="C").reshape(self.B, self.traj)
        x_s = self._f(x_s, (self.B, self.nx))
        ref = self._f(ref, (self.B, 6 * self.N))
        t = C.c_double(0.0)
        self._chk(L.gato_solve(self.h, self._p(xu), float(timestep), self._p(x_s), self._p(ref), C.byref(t)))
        out = self.stats()
        out["XU"] = xu
        out["sqp_time_us"] = t.value
        return out

    def solve_device(self, d_xu, timestep, d_x_s, d_ref, stream=0):
        """BSQP::solve on raw device pointers (ints), asynchronous on `stream`."""
        self._chk(self.L.gato_solve_device(self.h, C.c_void_p(d_xu), float(timestep), C.c_void_p(d_x_s), C.c_void_p(d_ref), C.c_void_p(stream)))

    def reset_async(self, dual=True, rho=True, stream=0):
        self._chk(self.L.gato_reset_async(self.h, int(dual), int(rho), C.c_void_p(stream)))

    def copy_final_merit_device(self, d_out, stream=0):
        self._chk(self.L.gato_copy_final_merit_device(self.h, C.c_void_p(d_out), C.c_void_p(stream)))

    def stats(self):
        L = self.L
        B = self.B
        it, ls = C.c_uint32(0), C.c_uint32(0)
        self._chk(L.gato_get_counts(self.h, C.byref(it), C.byref(ls)))
        it, ls = it.value, ls.value
        sqp_iters = np.zeros(B, np.int32)
        conv = np.zeros(B, np.int32)
        fm = np.zeros(B, self.dtype)
        im = np.zeros(B, self.dtype)
        pcg = np.zeros((max(it, 1), B), np.int32)
        mm = np.zeros((max(ls, 1), B), self.dtype)
        ss = np.zeros((max(ls, 1), B), self.dtype)
        ip = C.POINTER(C.c_int32)
        self._chk(L.gato_get_sqp_iters(self.h, sqp_iters.ctypes.data_as(ip)))
        self._chk(L.gato_get_kkt_converged(self.h, conv.ctypes.data_as(ip)))
        self._chk(L.gato_get_final_merit(self.h, self._p(fm)))
        self._chk(L.gato_get_initial_merit(self.h, self._p(im)))
        self._chk(L.gato_get_pcg_iters(self.h, pcg.ctypes.data_as(ip)))
        self._chk(L.gato_get_ls_min_merit(self.h, self._p(mm)))
        self._chk(L.gato_get_ls_step_size(self.h, self._p(ss)))
        return {
            "sqp_iters": sqp_iters, "kkt_converged": conv, "final_merit": fm, "initial_merit": im, "ls_num_iters": int(ls),
            "pcg_times_us": np.zeros(ls, self.dtype),       # always 0 in the reference too (bsqp.cuh:138)
            "pcg_iters": pcg[:ls],                            # the reference truncates to the line searches done (bindings.cu:111-128)
            "ls_min_merit": mm[:ls], "ls_step_size": ss[:ls],
            "iters_done": int(it), "pcg_iters_all": pcg[:it],
        }

    # ---- setters ----
    def set_f_ext_batch(self, f):
        self._chk(self.L.gato_set_f_ext_batch(self.h, self._p(self._f(f, (self.B, 6)))))

    def set_rho_penalty_batch(self, v, set_as_reset_default=True):
        self._chk(self.L.gato_set_rho_penalty_batch(self.h, self._p(self._f(v, (self.B,))), int(bool(set_as_reset_default))))

    def set_drho_batch(self, v, set_as_reset_default=True):
        self._chk(self.L.gato_set_drho_batch(self.h, self._p(self._f(v, (self.B,))), int(bool(set_as_reset_default))))

    def set_mu_batch(self, v):
        self._chk(self.L.gato_set_mu_batch(self.h, self._p(self._f(v, (self.B,)))))

    def set_cost_weights_batch(self, w):
        """w[B,7] = q, qd, u, N, q_lim, vel_lim, ctrl_lim cost weights per trajectory (extension: SURVEY 8(f)3)"""
        self._chk(self.L.gato_set_cost_weights_batch(self.h, self._p(self._f(w, (self.B, 7)))))

    def set_pcg_tol_batch(self, v):
        self._chk(self.L.gato_set_pcg_tol_batch(self.h, self._p(self._f(v, (self.B,)))))

    def reset_dual(self):
        self._chk(self.L.gato_reset_dual(self.h))

    def reset_rho(self):
        self._chk(self.L.gato_reset_rho(self.h))

    def set_rho_adaptation(self, enabled):
        self._chk(self.L.gato_set_rho_adaptation(self.h, int(bool(enabled))))

    def set_graph_mode(self, enabled):
        """replay the host-buffer solve as a hipGraph (bit-identical; off by default)"""
        self._chk(self.L.gato_set_graph_mode(self.h, int(bool(enabled))))

    def set_linear_solver(self, mode):
        """"pcg" (the reference's solver) or "direct" (block-tridiagonal LU sweep; extension)"""
        self._chk(self.L.gato_set_linear_solver(self.h, {"pcg": 0, "direct": 1}[mode]))

    def sim_forward(self, xk, uk, dt):
        out = np.zeros((self.B, self.nx), self.dtype)
        self._chk(self.L.gato_sim_forward(self.h, self._p(out), self._p(self._f(xk, (self.nx,))), self._p(self._f(uk, (self.nu,))), float(dt)))
        return out

    def sim_forward_device(self, d_xkp1, d_xk, d_uk, dt, stream=0):
        """BSQP::sim_forward on raw device pointers (ints), asynchronous on `stream` (bsqp.cuh:91)."""
        self._chk(self.L.gato_sim_forward_device(self.h, C.c_void_p(d_xkp1), C.c_void_p(d_xk), C.c_void_p(d_uk), float(dt), C.c_void_p(stream)))

    def select_best(self, x_last, u_last, x_meas, dt):
        """(best index, errors[B]): MPC hypothesis selection on the device (mpc_controller.py:294-309)"""
        err = np.zeros(self.B, self.dtype)
        best = C.c_int(0)
        self._chk(self.L.gato_select_best(self.h, self._p(self._f(x_last, (self.nx,))), self._p(self._f(u_last, (self.nu,))), self._p(self._f(x_meas, (self.nx,))), float(dt),
                                     C.byref(best), self._p(err)))
        return best.value, err

    def plant_rk4(self, x, u_seq, f_ext6, sim_dt):
        """nsteps = len(u_seq) RK4 steps of the library's forward dynamics (the MPC loop's plant simulator); returns the new state"""
        x = np.array(x, dtype=self.dtype).reshape(self.nx)
        u = self._f(u_seq).reshape(-1, self.nu)
        self._chk(self.L.gato_plant_rk4(self.h, self._p(x), self._p(u), int(u.shape[0]), self._p(self._f(f_ext6, (6,))), float(sim_dt)))
        return x

    def plant_payload_rk4(self, x, pend11, u_seq, f_ext6, sim_dt):
        """plant_rk4 with the swinging payload pend11 = [quat xyzw | w | mass, length, damping, inertia]; returns (new state, new pend11)"""
        x = np.array(x, dtype=self.dtype).reshape(self.nx)
        pend = np.array(pend11, dtype=self.dtype).reshape(11)
        u = self._f(u_seq).reshape(-1, self.nu)
        self._chk(self.L.gato_plant_payload_rk4(self.h, self._p(x), self._p(pend), self._p(u), int(u.shape[0]), self._p(self._f(f_ext6, (6,))), float(sim_dt)))
        return x, pend

    def mpc_set_payload(self, pend11):
        self._chk(self.L.gato_mpc_set_payload(self.h, None if pend11 is None else self._p(self._f(pend11, (11,)))))

    def mpc_payload(self):
        out = np.zeros(7, self.dtype)
        self._chk(self.L.gato_mpc_get_payload(self.h, self._p(out)))
        return out

    # ---- MPC session (gato_mpc_*): one call per MPC step, the loop's state stays on the device ----
    def mpc_begin(self, x0):
        self._chk(self.L.gato_mpc_begin(self.h, self._p(self._f(x0, (self.nx,)))))

    def mpc_step(self, advance=True, plan=True, plant_steps=0, sim_dt=0.001, steps_per_knot=1.0, plant_wrench=None, ref_window=None, hyp_world=None,
                 select=False, select_dt=0.0, time_solve_wall=False):
        io = self.L._MPC()
        io.struct_size = C.sizeof(io)
        io.phases = (1 if advance else 0) | (2 if plan else 0) | (4 if time_solve_wall else 0)
        io.plant_steps, io.sim_dt, io.steps_per_knot = int(plant_steps), float(sim_dt), float(steps_per_knot)
        fw = self._f(np.zeros(6) if plant_wrench is None else plant_wrench, (6,))
        for i in range(6):
            io.plant_wrench[i] = float(fw[i])
        keep = []
        if ref_window is not None:
            rw = self._f(ref_window, (6 * self.N,)); keep.append(rw); io.ref_window = self._p(rw)
        if hyp_world is not None:
            hw = self._f(hyp_world, (6 * self.B,)); keep.append(hw); io.hyp_world = self._p(hw)
        io.select, io.select_dt = int(bool(select)), float(select_dt)
        err = np.zeros(self.B, self.dtype)
        io.errors = self._p(err)
        self._chk(self.L.gato_mpc_step(self.h, C.byref(io)))
        return {"x": np.array(io.x[: self.nx], self.dtype), "ee": np.array(io.ee[:3], self.dtype), "best": int(io.best), "solve_us": float(io.solve_us), "solve_wall_us": float(io.solve_wall_us),
                "plant_us": float(io.plant_us), "errors": err}

    def mpc_best(self):
        out = np.zeros(self.traj, self.dtype)
        self._chk(self.L.gato_mpc_get_best(self.h, self._p(out)))
        return out

    # ---- one batch sharded over the GPUs of a node (gato_comm_*): RCCL inside the library, no torch type in sight ----
    @staticmethod
    def comm_unique_id(f64=False):
        """128 bytes from ncclGetUniqueId: rank 0 creates them, every rank passes them to comm_init"""
        L = load(f64)
        buf = C.create_string_buffer(128)
        _chk(L.gato_comm_unique_id(buf), L)
        return buf.raw

    @staticmethod
    def comm_available(f64=False):
        """None when librccl can be opened and has every entry point the library binds, else the reason; no RCCL call is made"""
        L = load(f64)
        return None if L.gato_comm_available() == 0 else L.gato_last_error().decode()

    def comm_init(self, unique_id, world_size, rank):
        """collective over all ranks: this solver becomes shard `rank` of a batch of world_size x B trajectories"""
        self._chk(self.L.gato_comm_init(self.h, C.c_char_p(bytes(unique_id)), int(world_size), int(rank), int(world_size) * self.B))

    def comm_init_rank(self, unique_id, world_size, rank):
        """step 1 of comm_init: ncclCommInitRank alone -- no collective on the new communicator yet (sharding.connect compares the ranks' outcomes first)"""
        self._chk(self.L.gato_comm_init_rank(self.h, C.c_char_p(bytes(unique_id)), int(world_size), int(rank), int(world_size) * self.B))

    def comm_confirm(self):
        """step 2, collective on the new communicator: the ranks agree on the solved-count mode; sharded solves refuse before it"""
        self._chk(self.L.gato_comm_confirm(self.h))

    def comm_destroy(self):
        self._chk(self.L.gato_comm_destroy(self.h))

    def gather_results(self, d_local, d_all, count, stream=0):
        """ncclAllGather of `count` reals per rank (raw device pointers), asynchronous on `stream`"""
        self._chk(self.L.gato_gather_results(self.h, C.c_void_p(d_local), C.c_void_p(d_all), int(count), C.c_void_p(stream)))

    def debug_set_remote_solved(self, per_iter, global_batch):
        """test hook: the other shards' solved counts per SQP iteration (no communicator); global_batch = 0 ends it"""
        a = np.ascontiguousarray(per_iter, dtype=np.uint32)
        self._chk(self.L.gato_debug_set_remote_solved(self.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), int(a.size), int(global_batch)))

    def set_solved_count_mode(self, mode):
        """sharded solves: "deferred" (default: speculative run, ONE reduction of the count vector per solve, exact replay when the exit rule
        fired) or "per_iteration" (one 4-byte all-reduce per SQP iteration)"""
        self._chk(self.L.gato_set_solved_count_mode(self.h, {"per_iteration": 0, "deferred": 1}[mode]))

    def shard_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.L.gato_get_shard_stats(self.h, C.byref(a), C.byref(b)))
        return {"deferred_solves": int(a.value), "replays": int(b.value)}

    def solved_count_state(self):
        """the count mode the handle is in, the sharded solves that counted per iteration so far (mode, capture, back-off after a replay) and
        how many more the back-off will take that way"""
        m, n, left = C.c_int(), C.c_uint64(), C.c_uint32()
        self._chk(self.L.gato_get_solved_count_state(self.h, C.byref(m), C.byref(n), C.byref(left)))
        return {"mode": "deferred" if m.value == 1 else "per_iteration", "per_iteration_solves": int(n.value), "per_iteration_left": int(left.value)}

    def synchronize(self):
        self._chk(self.L.gato_synchronize(self.h))

    def ee_pos(self, q):
        q = self._f(q).reshape(-1, self.nq)
        out = np.zeros((q.shape[0], 3), self.dtype)
        self._chk(self.L.gato_ee_pos(self.h, self._p(q), q.shape[0], self._p(out)))
        return out

    # ---- profiling / debug ----
    def set_profiling(self, enabled):
        self._chk(self.L.gato_set_profiling(self.h, int(bool(enabled))))

    def stage_times_us(self):
        out = (C.c_double * 7)()
        self._chk(self.L.gato_get_stage_times_us(self.h, out))
        return dict(zip(["merit", "kkt", "schur", "pcg", "dz", "line_search", "total"], list(out)))

    def read(self, name):
        n = C.c_uint64(0)
        self._chk(self.L.gato_debug_read(self.h, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(n.value, self.dtype)
        self._chk(self.L.gato_debug_read(self.h, name.encode(), self._p(out), n.value, None))
        return out

    def write(self, name, arr):
        a = self._f(arr).reshape(-1)
        self._chk(self.L.gato_debug_write(self.h, name.encode(), self._p(a), a.size))

    def stage(self, stage, xu, timestep, x_s, ref):
        xu = np.array(xu, dtype=self.dtype, order="C").reshape(self.B, self.traj)
        x_s = self._f(x_s, (self.B, self.nx))
        ref = self._f(ref, (self.B, 6 * self.N))
        self._chk(self.L.gato_debug_stage(self.h, STAGES[stage], self._p(xu), float(timestep), self._p(x_s), self._p(ref), None))
        return xu

    # expansion of the compact KKT storage into the reference's dense blocks (for stage comparisons)
    def dense_kkt(self, dt):
        B, N, nq, nx, nu = self.B, self.N, self.nq, self.nx, self.nu
        D = self.read("D").reshape(B, N, 3 * nq, nq)           # [c][r] col-major nq x 3nq
        h2 = self.dtype(0.5 * float(self.dtype(dt)) * float(self.dtype(dt)))
        dtf = self.dtype(dt)
        A = np.zeros((B, N, nx, nx), self.dtype)                # A[b,k,c,r] (col-major blocks like the reference's memory)
        Bm = np.zeros((B, N, nu, nx), self.dtype)
        eye = np.eye(nq, dtype=self.dtype)
        Dq, Dd, Mi = D[:, :, :nq], D[:, :, nq:2 * nq], D[:, :, 2 * nq:]
        A[:, :, :nq, :nq] = eye + h2 * Dq
        A[:, :, :nq, nq:] = dtf * Dq
        A[:, :, nq:, :nq] = dtf * eye + h2 * Dd
        A[:, :, nq:, nq:] = eye + dtf * Dd
        Bm[:, :, :, :nq] = h2 * Mi
        Bm[:, :, :, nq:] = dtf * Mi
        A[:, N - 1] = 0
        Bm[:, N - 1] = 0

        def blk(qq, dd):
            out = np.zeros((B, N, nx, nx), self.dtype)
            out[:, :, :nq, :nq] = qq.reshape(B, N, nq, nq)
            idx = np.arange(nq)
            out[:, :, nq + idx, nq + idx] = dd.reshape(B, N, nq)
            return out

        def dg(d):
            out = np.zeros((B, N, nu, nu), self.dtype)
            idx = np.arange(nu)
            out[:, :, idx, idx] = d.reshape(B, N, nu)
            return out

        return dict(A=A, B=Bm, Q=blk(self.read("Qq"), self.read("Qd")), R=dg(self.read("Rd")), q=self.read("q").reshape(B, N, nx),
                    r=self.read("r").reshape(B, N, nu), c=self.read("c").reshape(B, N, nx),
                    Qinv=blk(self.read("Qqi"), self.read("Qdi")), Rinv=dg(self.read("Rdi")))
