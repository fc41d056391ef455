"""The C++ boundary on the device: a hip*-renamed program of the shape of the reference's examples/bsqp.cu:7-77, compiled against
include/bsqp.hpp and libgato_hip.so, must produce the bits of the Python path (both sit on the same C ABI)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "gato_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "cpp", "example_bsqp.cpp")


def _build(tmp_path, f64=False):
    exe = str(tmp_path / ("example_bsqp_f64" if f64 else "example_bsqp"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L" + LIBDIR, "-lgato_hip_f64" if f64 else "-lgato_hip", "-Wl,-rpath," + LIBDIR] + (["-DGATO_DOUBLE"] if f64 else []))
    return exe


@pytest.mark.parametrize("f64", [False, True])
def test_cpp_example_compiles(tmp_path, f64):
    """no GPU needed: the translation unit builds and links against the C ABI (hipcc host compile); BSQP<double, B> with -DGATO_DOUBLE"""
    assert os.path.exists(_build(tmp_path, f64))


@pytest.mark.gpu
@pytest.mark.parametrize("f64", [False, True])
def test_cpp_example_equals_python_path(tmp_path, f64):
    from gato_amd._lib import NativeSolver
    exe = _build(tmp_path, f64)
    T = np.float64 if f64 else np.float32
    w = 8 if f64 else 4
    out = str(tmp_path / "out.bin")
    r = subprocess.run([exe, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "XU Traj:" in r.stdout
    B, N, nx, nu = 16, 16, 12, 6
    traj = (nx + nu) * (N - 1) + nx
    raw = np.fromfile(out, dtype=np.uint8)
    nf = B * traj + B + B * nx
    f = raw[: w * nf].view(T)
    meta = raw[w * nf:].view(np.float64)
    xu_c, merit_c, next_c = f[: B * traj].reshape(B, traj), f[B * traj: B * traj + B], f[B * traj + B:].reshape(B, nx)
    # the same problem through the Python binding
    ref = np.zeros((B, N, 6), np.float32)
    b_, k_ = np.meshgrid(np.arange(B), np.arange(N), indexing="ij")
    ref[:, :, 0] = np.float32(0.30) + np.float32(0.005) * k_.astype(np.float32) + np.float32(0.01) * b_.astype(np.float32)
    ref[:, :, 1] = np.float32(0.35) - np.float32(0.002) * k_.astype(np.float32)
    ref[:, :, 2] = np.float32(0.80) - np.float32(0.004) * k_.astype(np.float32)
    x0 = np.array([-1.0, -0.1, 0.8, -0.1, 0.5, 0.01, 0, 0, 0, 0, 0, 0], np.float32)
    xs = np.tile(x0.astype(T), (B, 1))                           # x0[i] (T) + 0.01f * b (float): the sum is formed in T
    xs[:, :6] += (np.float32(0.01) * np.arange(B, dtype=np.float32)).astype(T)[:, None]
    xu = np.zeros((B, traj), T)
    for k in range(N):
        xu[:, k * (nx + nu): k * (nx + nu) + nx] = xs
    fext = np.zeros((B, 6), np.float32)
    fext[:, 2] = np.float32(0.5) * np.arange(B, dtype=np.float32)
    c = lambda v: float(np.float32(v))                           # the example's float literals, widened in the double build
    s = NativeSolver("indy7", N, B, f64=f64, dt=c(0.03), max_sqp_iters=4, kkt_tol=c(1e-3), max_pcg_iters=100, pcg_tol=c(1e-4), solve_ratio=1.0,
                     mu=10.0, q_cost=2.0, qd_cost=c(1e-2), u_cost=c(2e-6), N_cost=50.0, q_lim_cost=c(0.01), vel_lim_cost=0.0, ctrl_lim_cost=0.0,
                     rho=c(0.01))
    s.set_f_ext_batch(fext)
    rp = s.solve(xu, c(0.03), xs, ref.reshape(B, -1))
    assert not np.array_equal(rp["XU"], xu)                      # the solve moved the iterates (non-zero costs)
    np.testing.assert_array_equal(xu_c, rp["XU"])
    np.testing.assert_array_equal(merit_c, rp["final_merit"])
    np.testing.assert_array_equal(next_c, s.sim_forward(x0, np.array([1.0, -2.0, 0.5, 0.1, -0.1, 0.05], np.float32), c(0.01)))
    assert meta[0] > 0 and int(meta[1]) == 4 and int(meta[2]) == rp["ls_num_iters"] and T(meta[3]) == rp["ls_step_size"][-1][0]


REF_EXAMPLE = "/root/reference/examples/bsqp.cu"


@pytest.mark.skipif(not os.path.exists(REF_EXAMPLE), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("f64", [False, True])
def test_the_reference_example_itself_compiles_after_renames(tmp_path, f64):
    """SURVEY 8(b)4 literally: the reference's examples/bsqp.cu, read at test time (never committed), with ONLY its cuda* -> hip* renames and
    its three gato includes pointed at include/bsqp.hpp, compiles and links against this library -- `T` (settings.h:7-11), `gpuErrchk`
    (utils/cuda.cuh:7-19), the 15-scalar constructor, ProblemInputs / SQPStats and the device-pointer solve are all the header's."""
    import re
    src = open(REF_EXAMPLE).read()
    src = re.sub(r'#include "bsqp/bsqp\.cuh"', '#include <hip/hip_runtime.h>\n#include "bsqp.hpp"', src)
    src = re.sub(r'#include "(types|utils/cuda)\.cuh"\n', "", src)
    src = re.sub(r"\bcuda([A-Z])", r"hip\1", src)
    assert "cuda" not in src and ".cuh" not in src
    cpp = tmp_path / "reference_example.cpp"
    cpp.write_text(src)
    exe = str(tmp_path / "reference_example")
    flags = ["-DUSE_DOUBLES", "-DGATO_DOUBLE"] if f64 else []
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", "-DPLANT_INDY7", "-DKNOT_POINTS=16", "-Wno-unused-result",
                           "-I", os.path.join(ROOT, "include"), str(cpp), "-o", exe, "-L" + LIBDIR, "-lgato_hip_f64" if f64 else "-lgato_hip",
                           "-Wl,-rpath," + LIBDIR] + flags)
    assert os.path.exists(exe)
