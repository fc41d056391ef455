"""ctypes front-end of the CPU oracle (oracle/gato_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by gato_amd/.
Parity status: "parity unpinned by reference execution" -- see the header of gato_oracle.c and DESIGN.md.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgato_oracle.so")


def build_native(out_dir):
    """The fp32 oracle compiled for THIS host (-march=native, BASELINE.md section 3's CPU baseline flags) into out_dir; the committed
    Makefile target is x86-64-v3 because that .so travels from the authoring container.  Returns the path, or None without a compiler."""
    out = os.path.join(out_dir, "libgato_oracle_native.so")
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=c99", "-ffp-contract=fast", "-shared", "-o", out,
                               os.path.join(HERE, "gato_oracle.c"), "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out
    except Exception:   # noqa: BLE001
        return None


def use_library(path):
    """fp32 oracle from another build of the same source (build_native); must be called before the first solver is created"""
    global LIB_PATH
    assert False not in _libs, "the fp32 oracle library is already loaded"
    LIB_PATH = path
PLANTS = {"indy7": 0, "iiwa14": 1}
NQ = {"indy7": 6, "iiwa14": 7}


def _params_struct(ft):
    class P(C.Structure):
        _fields_ = [("dt", ft), ("max_sqp_iters", C.c_uint32), ("kkt_tol", ft), ("max_pcg_iters", C.c_uint32), ("pcg_tol", ft),
                    ("solve_ratio", ft), ("mu", ft), ("q_cost", ft), ("qd_cost", ft), ("u_cost", ft), ("N_cost", ft), ("q_lim_cost", ft),
                    ("vel_lim_cost", ft), ("ctrl_lim_cost", ft), ("rho", ft)]
    return P


OrcParams = _params_struct(C.c_float)
OrcParams64 = _params_struct(C.c_double)   # the -Dfloat=double build of the same source (oracle/Makefile: libgato_oracle_f64.so)
LIB_PATH_F64 = os.path.join(HERE, "libgato_oracle_f64.so")


def build(force=False):
    src = os.path.join(HERE, "gato_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "libgato_oracle.so"])
    return LIB_PATH


_lib = None


_libs = {}


def lib(f64=False, path=None):
    """The fp32 oracle (default) or the float64 build of the same source (sensitivity studies: tests/test_oracle_sensitivity.py).
    path: another fp32 build of the same source (build_native), loaded beside the default one (bench.py times both)."""
    key = bool(f64) if path is None else os.path.abspath(path)
    if key not in _libs:
        f64 = bool(f64) and path is None
        if path is None:
            path = LIB_PATH_F64 if f64 else LIB_PATH
        if not os.path.exists(path):
            build()
        ft = C.c_double if f64 else C.c_float
        PT = OrcParams64 if f64 else OrcParams
        L = C.CDLL(path)
        fp = C.POINTER(ft)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(PT)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_solve.restype = C.c_uint32
        L.orc_solve.argtypes = [C.c_void_p, fp, ft, fp, fp]
        L.orc_buf.restype = fp
        L.orc_buf.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_ibuf.restype = C.POINTER(C.c_int32)
        L.orc_ibuf.argtypes = [C.c_void_p, C.c_char_p]
        for name in ("orc_set_f_ext", "orc_set_mu", "orc_set_pcg_tol", "orc_set_cost_weights"):
            getattr(L, name).argtypes = [C.c_void_p, fp]
        for name in ("orc_set_rho", "orc_set_drho"):
            getattr(L, name).argtypes = [C.c_void_p, fp, C.c_int]
        for name in ("orc_reset_dual", "orc_reset_rho", "orc_form_schur", "orc_pcg", "orc_compute_dz"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_set_rho_adaptation.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_setup_kkt.argtypes = [C.c_void_p, fp, fp, fp, ft]
        L.orc_merit.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp, ft, C.c_int]
        L.orc_line_search.argtypes = [C.c_void_p, fp]
        L.orc_sim_forward.argtypes = [C.c_void_p, fp, fp, fp, ft]
        L._REDUCE = C.CFUNCTYPE(C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p)
        L.orc_set_shard.argtypes = [C.c_void_p, L._REDUCE, C.c_void_p, C.c_long]
        L.orc_iters_done.restype = C.c_uint32
        L.orc_iters_done.argtypes = [C.c_void_p]
        L.orc_ls_done.restype = C.c_uint32
        L.orc_ls_done.argtypes = [C.c_void_p]
        L.orc_fd.argtypes = [C.c_int, fp, fp, fp, fp, fp]
        L.orc_fd_grad.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp]
        L.orc_rnea.argtypes = [C.c_int, fp, fp, fp, fp, fp]
        L.orc_minv.argtypes = [C.c_int, fp, fp]
        L.orc_ee.argtypes = [C.c_int, fp, fp, fp]
        L.orc_gj_inverse.argtypes = [C.c_int, fp, fp, C.c_int]
        L._ft, L._np, L._PT = ft, (np.float64 if f64 else np.float32), PT
        _libs[key] = L
    return _libs[key]


def _f(a, L=None):
    dt = np.float32 if L is None else L._np
    a = np.ascontiguousarray(a, dtype=dt)
    return a, a.ctypes.data_as(C.POINTER(C.c_float if L is None else L._ft))


PARAM_ORDER = ["dt", "max_sqp_iters", "kkt_tol", "max_pcg_iters", "pcg_tol", "solve_ratio", "mu", "q_cost", "qd_cost", "u_cost", "N_cost",
               "q_lim_cost", "vel_lim_cost", "ctrl_lim_cost", "rho"]


class OracleSolver:
    """Mirror of the `BSQP_{B}_float` class surface (python/bindings.cu:224-237) on the CPU oracle, plus stage access."""

    def __init__(self, plant, N, B, dt=0.01, max_sqp_iters=5, kkt_tol=1e-4, max_pcg_iters=100, pcg_tol=1e-5, solve_ratio=1.0, mu=10.0,
                 q_cost=1.0, qd_cost=1e-3, u_cost=1e-6, N_cost=50.0, q_lim_cost=1e-3, vel_lim_cost=0.0, ctrl_lim_cost=0.0, rho=1e-3, threads=1, f64=False, library=None):
        self.plant, self.N, self.B = plant, N, B
        self.nq = NQ[plant]
        self.nx, self.nu = 2 * self.nq, self.nq
        self.traj = (self.nx + self.nu) * N - self.nu
        self.max_sqp_iters = max_sqp_iters
        self.L = lib(f64, library)
        self.dtype = self.L._np
        self.p = self.L._PT(dt, max_sqp_iters, kkt_tol, max_pcg_iters, pcg_tol, solve_ratio, mu, q_cost, qd_cost, u_cost, N_cost, q_lim_cost,
                           vel_lim_cost, ctrl_lim_cost, rho)
        self.h = self.L.orc_create(PLANTS[plant], N, B, C.byref(self.p))
        self.L.orc_set_threads(self.h, threads)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    # ---- setters ----
    def set_f_ext_batch(self, f):
        a, p = _f(np.asarray(f).reshape(self.B, 6), self.L); self.L.orc_set_f_ext(self.h, p)

    def set_rho_penalty_batch(self, v, set_as_reset_default=True):
        a, p = _f(v, self.L); self.L.orc_set_rho(self.h, p, int(set_as_reset_default))

    def set_drho_batch(self, v, set_as_reset_default=True):
        a, p = _f(v, self.L); self.L.orc_set_drho(self.h, p, int(set_as_reset_default))

    def set_cost_weights_batch(self, w):
        """w[B,7] = q, qd, u, N, q_lim, vel_lim, ctrl_lim cost weights per trajectory"""
        a, p = _f(w, self.L); assert a.size == 7 * self.B; self.L.orc_set_cost_weights(self.h, p)

    def set_mu_batch(self, v):
        a, p = _f(v, self.L); self.L.orc_set_mu(self.h, p)

    def set_pcg_tol_batch(self, v):
        a, p = _f(v, self.L); self.L.orc_set_pcg_tol(self.h, p)

    def reset_dual(self):
        self.L.orc_reset_dual(self.h)

    def reset_rho(self):
        self.L.orc_reset_rho(self.h)

    def set_rho_adaptation(self, e):
        self.L.orc_set_rho_adaptation(self.h, int(bool(e)))

    def set_shard(self, reduce, global_batch):
        """Multi-process tests: this solver holds a shard of a batch of `global_batch` trajectories; `reduce(local_solved_count, sqp_iter)`
        returns the count over all ranks (the 4-byte all-reduce per SQP iteration of SURVEY 8(e)); the exit rule of bsqp.cuh:165 then sees the
        whole batch."""
        self._reduce_cb = self.L._REDUCE(lambda n, it, ctx: int(reduce(int(n), int(it))))   # keep the thunk alive
        self.L.orc_set_shard(self.h, self._reduce_cb, None, int(global_batch))

    # ---- stage access ----
    SHAPES = {"Q": ("N", "nx", "nx"), "A": ("N", "nx", "nx"), "Qinv": ("N", "nx", "nx"), "R": ("N", "nu", "nu"), "Rinv": ("N", "nu", "nu"),
              "B": ("N", "nu", "nx"), "q": ("N", "nx"), "c": ("N", "nx"), "r": ("N", "nu"), "S": ("N", "nx", "3nx"), "Pinv": ("N", "nx", "3nx"),
              "gamma": ("N+2", "nx"), "lambda": ("N+2", "nx"), "dz": ("traj",), "merit": (8,), "merit_cur": (), "merit_init0": (), "step": (),
              "rho": (), "drho": ()}

    def _dim(self, d):
        return {"N": self.N, "nx": self.nx, "nu": self.nu, "3nx": 3 * self.nx, "N+2": self.N + 2, "traj": self.traj}.get(d, d)

    def buf(self, name):
        """Copy of a stage buffer as [B, ...]; matrices keep the reference's in-memory order (col-major blocks, row-major S/Pinv rows)."""
        shape = (self.B,) + tuple(self._dim(d) for d in self.SHAPES[name])
        ptr = self.L.orc_buf(self.h, name.encode())
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape).copy()

    def ibuf(self, name, shape):
        ptr = self.L.orc_ibuf(self.h, name.encode())
        return np.ctypeslib.as_array(ptr, shape=(int(np.prod(shape)),)).reshape(shape).copy()

    def set_lambda(self, lam):
        ptr = self.L.orc_buf(self.h, b"lambda")
        np.ctypeslib.as_array(ptr, shape=(self.B * (self.N + 2) * self.nx,))[:] = np.asarray(lam, self.dtype).reshape(-1)

    def set_dz(self, dz):
        ptr = self.L.orc_buf(self.h, b"dz")
        np.ctypeslib.as_array(ptr, shape=(self.B * self.traj,))[:] = np.asarray(dz, self.dtype).reshape(-1)

    def set_buf(self, name, arr):
        """overwrite a stage buffer (teacher-forced stage tests: the line search from a GIVEN merit table, ...)"""
        shape = (self.B,) + tuple(self._dim(d) for d in self.SHAPES[name])
        n = int(np.prod(shape))
        np.ctypeslib.as_array(self.L.orc_buf(self.h, name.encode()), shape=(n,))[:] = np.asarray(arr, self.dtype).reshape(-1)

    def line_search(self, xu):
        """lineSearchAndUpdateBatchedKernel (line_search.cuh:13-98) on the solver's merit table / merit_cur / rho / drho / dz; returns the new xu"""
        xu = np.array(xu, dtype=self.dtype, order="C").reshape(self.B, self.traj)
        self.L.orc_line_search(self.h, xu.ctypes.data_as(C.POINTER(self.L._ft)))
        return xu

    def setup_kkt(self, xu, x_s, ref, dt):
        (_, a), (_, b), (_, c) = _f(xu, self.L), _f(x_s, self.L), _f(ref, self.L)
        self.L.orc_setup_kkt(self.h, a, b, c, dt)

    def form_schur(self):
        self.L.orc_form_schur(self.h)

    def pcg(self):
        self.L.orc_pcg(self.h)

    def compute_dz(self):
        self.L.orc_compute_dz(self.h)

    def merit(self, xu, x_s, ref, dt, num_alphas=8, zero_dz=False):
        out = np.zeros((self.B, num_alphas), self.dtype)
        (_, a), (_, b), (_, c) = _f(xu, self.L), _f(x_s, self.L), _f(ref, self.L)
        self.L.orc_merit(self.h, num_alphas, out.ctypes.data_as(C.POINTER(self.L._ft)), a, b, c, dt, int(zero_dz))
        return out

    # ---- solve ----
    def solve(self, xu, dt, x_s, ref):
        """Same result dict as PyBSQP::solve (python/bindings.cu:96-145); sqp_time_us is this CPU's wall time."""
        import time
        xu = np.array(xu, dtype=self.dtype, order="C").reshape(self.B, self.traj)
        xs, pxs = _f(np.asarray(x_s).reshape(self.B, self.nx), self.L)
        rf, prf = _f(np.asarray(ref).reshape(self.B, 6 * self.N), self.L)
        t0 = time.perf_counter()
        iters = self.L.orc_solve(self.h, xu.ctypes.data_as(C.POINTER(self.L._ft)), dt, pxs, prf)
        t1 = time.perf_counter()
        ls = self.L.orc_ls_done(self.h)
        B = self.B
        mi = max(self.max_sqp_iters, 1)
        return {
            "XU": xu,
            "sqp_time_us": (t1 - t0) * 1e6,
            "sqp_iters": self.ibuf("sqp_iters", (B,)),
            "kkt_converged": self.ibuf("kkt_converged", (B,)),
            "final_merit": self.buf("merit_cur"),
            "initial_merit": self.buf("merit_init0"),
            "ls_num_iters": int(ls),
            "pcg_times_us": np.zeros(ls, self.dtype),
            "pcg_iters": self.ibuf("st_pcg_iters", (mi, B))[:ls],
            "ls_min_merit": np.ctypeslib.as_array(self.L.orc_buf(self.h, b"st_min_merit"), shape=(mi * B,)).reshape(mi, B)[:ls].copy(),
            "ls_step_size": np.ctypeslib.as_array(self.L.orc_buf(self.h, b"st_step"), shape=(mi * B,)).reshape(mi, B)[:ls].copy(),
            # not part of the reference's dict: the candidates of every line search (8 merits + the merit before it), for the tests
            "ls_merits": np.ctypeslib.as_array(self.L.orc_buf(self.h, b"st_merits"), shape=(mi * B * 8,)).reshape(mi, B, 8)[:ls].copy(),
            "ls_merit_before": np.ctypeslib.as_array(self.L.orc_buf(self.h, b"st_merit_before"), shape=(mi * B,)).reshape(mi, B)[:ls].copy(),
            "iters_done": int(iters),
            "pcg_iters_all": self.ibuf("st_pcg_iters", (mi, B))[:iters],
        }

    def sim_forward(self, xk, uk, dt):
        out = np.zeros((self.B, self.nx), self.dtype)
        (_, a), (_, b) = _f(xk, self.L), _f(uk, self.L)
        self.L.orc_sim_forward(self.h, out.ctypes.data_as(C.POINTER(self.L._ft)), a, b, dt)
        return out


# ---- unit-level functions ----
def fd(plant, q, qd, u, f_ext=None):
    nq = NQ[plant]
    f_ext = np.zeros(6) if f_ext is None else f_ext
    out = np.zeros(nq, np.float32)
    (_, a), (_, b), (_, c), (_, d) = _f(q), _f(qd), _f(u), _f(f_ext)
    lib().orc_fd(PLANTS[plant], a, b, c, d, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def fd_grad(plant, q, qd, u, f_ext=None):
    nq = NQ[plant]
    f_ext = np.zeros(6) if f_ext is None else f_ext
    qdd = np.zeros(nq, np.float32)
    d = np.zeros(3 * nq * nq, np.float32)
    (_, a), (_, b), (_, c), (_, e) = _f(q), _f(qd), _f(u), _f(f_ext)
    lib().orc_fd_grad(PLANTS[plant], a, b, c, e, qdd.ctypes.data_as(C.POINTER(C.c_float)), d.ctypes.data_as(C.POINTER(C.c_float)))
    return qdd, d.reshape(3 * nq, nq).T.copy()  # [nq, 3nq] = [dqdd/dq | dqdd/dqd | Minv]


def rnea(plant, q, qd, qdd, f_ext=None):
    nq = NQ[plant]
    f_ext = np.zeros(6) if f_ext is None else f_ext
    out = np.zeros(nq, np.float32)
    (_, a), (_, b), (_, c), (_, d) = _f(q), _f(qd), _f(qdd), _f(f_ext)
    lib().orc_rnea(PLANTS[plant], a, b, c, d, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def minv(plant, q):
    nq = NQ[plant]
    out = np.zeros(nq * nq, np.float32)
    _, a = _f(q)
    lib().orc_minv(PLANTS[plant], a, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(nq, nq).T.copy()


def ee(plant, q):
    nq = NQ[plant]
    e = np.zeros(3, np.float32)
    J = np.zeros(3 * nq, np.float32)
    _, a = _f(q)
    lib().orc_ee(PLANTS[plant], a, e.ctypes.data_as(C.POINTER(C.c_float)), J.ctypes.data_as(C.POINTER(C.c_float)))
    return e, J.reshape(nq, 3).T.copy()  # J: [3, nq]


def gj_inverse(M, one_matrix_form=False):
    n = M.shape[0]
    V = np.asfortranarray(M, dtype=np.float32).ravel(order="F").copy()
    out = np.zeros(n * n, np.float32)
    lib().orc_gj_inverse(n, V.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)), int(one_matrix_form))
    return out.reshape(n, n).T.copy()
