#!/usr/bin/env python3
"""Generate the rigid-body model tables for the two plants (indy7, iiwa14).

Inputs are the physical parameters of the arms (joint origins, link inertials, joint limits) transcribed
from the robots' URDF descriptions (reference copies: examples/indy7_description/indy7.urdf:84-248,
examples/iiwa_description/iiwa14.urdf:70-337).  Outputs:

  oracle/robot_tables.h                 plain-C tables for the CPU oracle
  gato_amd/csrc/robot_models.hpp        constexpr traits for the HIP kernels

Model conventions (Featherstone spatial algebra, what the reference's generated GRiD code evaluates,
SURVEY.md Appendix B): every joint is revolute about its local z axis, parent(k) = k-1,
  X_k(q) = [E 0; -E r~ E],  E = Ez(q_k) * E0_k,  E0_k = R0_k^T (joint-origin rotation, parent<-child),
  Xhom_k(q) = [R0_k Rz(q_k) | r_k],
  I_k = spatial inertia of link k about its joint-frame origin (6x6, [angular; linear] ordering).
Quirk 15: the reference's indy7 table drops every mass-COM coupling term (I = blkdiag(I_urdf, m 1));
its iiwa14 table has the proper COM-shifted inertias, with the fixed 'contact' link merged into link 7.
`tests/test_robot_tables.py` proves both against the literals of the reference's `init_XImats`.
"""
import math
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

H = math.pi / 2


def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    R[np.abs(R) < 1e-6] = 0.0
    R[np.abs(R - 1) < 1e-6] = 1.0
    R[np.abs(R + 1) < 1e-6] = -1.0
    return R


def skew(c):
    return np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]])


def spatial_inertia(bodies, ignore_com):
    """bodies: list of (mass, com, (ixx,ixy,ixz,iyy,iyz,izz)) rigidly attached to the joint frame."""
    I6 = np.zeros((6, 6))
    for m, c, (ixx, ixy, ixz, iyy, iyz, izz) in bodies:
        Ic = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        c = np.zeros(3) if ignore_com else np.asarray(c, float)
        C = skew(c)
        I6[:3, :3] += Ic + m * (C @ C.T)
        I6[:3, 3:] += m * C
        I6[3:, :3] += m * C.T
        I6[3:, 3:] += m * np.eye(3)
    return I6


INDY7 = dict(
    name="indy7", nq=6, ignore_com=True, barrier_mode=0,
    joints=[  # (xyz, rpy)
        ((0, 0, 0.0775), (0, 0, 0)),
        ((0, -0.109, 0.222), (H, H, 0)),
        ((-0.45, 0, -0.0305), (0, 0, 0)),
        ((-0.267, 0, -0.075), (-H, 0, H)),
        ((0, -0.114, 0.083), (H, H, 0)),
        ((-0.168, 0, 0.069), (-H, 0, H)),
    ],
    links=[
        [(11.44444535, (-0.00023749, -0.04310313, 0.13245396), (0.35065005, 0.00011931, -0.00037553, 0.304798, -0.10984447, 0.06003147))],
        [(5.84766553, (-0.29616699, 2.254e-05, 0.04483069), (0.03599743, -4.693e-05, -0.05240346, 0.72293306, 1.76e-06, 0.70024119))],
        [(2.68206064, (-0.16804016, 0.00021421, -0.07000383), (0.0161721, -0.00011817, 0.03341882, 0.11364055, -4.371e-05, 0.10022522))],
        [(2.12987371, (-0.00026847, -0.0709844, 0.07649128), (0.02798891, 3.893e-05, -4.768e-05, 0.01443076, -0.01266296, 0.01496211))],
        [(2.22412271, (-0.09796232, -0.00023114, 0.06445892), (0.01105297, 5.517e-05, -0.01481977, 0.03698291, -3.74e-05, 0.02754795))],
        [(0.38254932, (8.147e-05, -0.00046556, 0.03079097), (0.00078982, -3.4e-07, 8.3e-07, 0.00079764, -5.08e-06, 0.00058319))],
    ],
    # limits as the plant header states them (gato/dynamics/indy7/indy7_plant.cuh:66-97), before the 0.1 margin
    q_lim=[3.0543, 3.0543, 3.0543, 3.0543, 3.0543, 3.7520],
    v_lim=[2.61, 2.61, 2.61, 3.14, 3.14, 3.14],
    u_lim=[431.97, 431.97, 197.23, 79.79, 79.79, 79.79],
)

IIWA14 = dict(
    name="iiwa14", nq=7, ignore_com=False, barrier_mode=1,
    joints=[
        ((0, 0, 0.1575), (0, 0, 0)),
        ((0, 0, 0.2025), (2.3561944901923457, -1.5707962635746238, 2.3561944901923457)),
        ((0.2045, 0, 0), (1.5707963267948948, -4.371139000186238e-8, 1.5707963705062866)),
        ((0, 0, 0.2155), (1.5707963705062866, 0, 0)),
        ((0, 0.1845, 0), (-1.5707963705062866, 0, 0)),
        ((0, -0.0607, 0.2155), (2.3561944901923457, -1.5707962635746238, 2.3561944901923457)),
        ((0.081, 0, 0.0607), (1.5707963267948948, -4.371139000186238e-8, 1.5707963705062866)),
    ],
    links=[
        [(3.94781, (0, 0, 0), (0.00455, 0, 0, 0.00454, -0.00001, 0.00029))],
        [(4.50275, (0.0003, 0.059, 0.042), (0.00032, 0, 0, 0.00010, 0, 0.00042))],
        [(2.45520, (0, 0.03, 0.13), (0.00223, -0.00005, 0.00007, 0.00219, 0.00007, 0.00073))],
        [(2.61155, (0, 0.067, 0.034), (0.03844, 0.00088, -0.00112, 0.01144, -0.00111, 0.04958))],
        [(3.41000, (0.0001, 0.021, 0.076), (0.00277, -0.00001, 0.00001, 0.00284, 0, 0.00012))],
        [(3.38795, (0, 0.0006, 0.0004), (0.00050, -0.00005, -0.00003, 0.00281, -0.00004, 0.00232))],
        # link 7 + the fixed 'contact' body 0.04 m up its z axis (iiwa14.urdf:322-337)
        [(0.35432, (0, 0, 0.02), (0.00795, 0.00022, -0.00029, 0.01083, -0.00029, 0.00294)),
         (0.057, (0, 0, 0.04), (0.0000354, 0, 0, 0.0000354, 0, 0.0000354))],
    ],
    q_lim=[2.96706, 2.09440, 2.96706, 2.09440, 2.96706, 2.09440, 3.05433],
    v_lim=[1.48353, 1.48353, 1.74533, 1.30900, 2.26893, 2.35619, 2.35619],
    u_lim=[320.0, 320.0, 176.0, 176.0, 110.0, 40.0, 40.0],
)


def build(model):
    nq = model["nq"]
    E0 = np.zeros((nq, 3, 3))
    r = np.zeros((nq, 3))
    I6 = np.zeros((nq, 6, 6))
    for k, (xyz, rpy) in enumerate(model["joints"]):
        E0[k] = rpy_to_R(*rpy).T
        r[k] = xyz
        I6[k] = spatial_inertia(model["links"][k], model["ignore_com"])
    # limits tightened by JOINT_LIMIT_MARGIN = -0.1 exactly as the float expression in the plant header evaluates:
    # lo = (float)(-L - (double)(float)(-0.1)), hi = (float)(L + (double)(float)(-0.1))
    margin = float(np.float32(-0.1))
    lim = {}
    for key in ("q_lim", "v_lim", "u_lim"):
        lim[key] = np.array([[np.float32(-L - margin), np.float32(L + margin)] for L in model[key]], dtype=np.float32)
    return dict(name=model["name"], nq=nq, E0=E0, r=r, I=I6, barrier_mode=model["barrier_mode"], **lim)


def fmt(x):
    s = repr(float(np.float32(x)))
    if "e" not in s and "." not in s and "inf" not in s:
        s += ".0"
    return s + "f"


def carr(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return "{" + ", ".join(fmt(v) for v in a) + "}"
    return "{" + ", ".join(carr(x) for x in a) + "}"


def emit_c(models, path):
    out = ["/* GENERATED by tools/gen_robot_models.py -- do not edit. */",
           "#ifndef GATO_ORACLE_ROBOT_TABLES_H", "#define GATO_ORACLE_ROBOT_TABLES_H", "",
           "#define ORC_MAX_NQ 7", "",
           "typedef struct {",
           "    const char* name; int nq; int barrier_mode;",
           "    float E0[ORC_MAX_NQ][9];   /* row-major 3x3, E0 = R0^T */",
           "    float r[ORC_MAX_NQ][3];    /* joint origin in the parent frame */",
           "    float I[ORC_MAX_NQ][36];   /* col-major 6x6 spatial inertia */",
           "    float q_lim[ORC_MAX_NQ][2], v_lim[ORC_MAX_NQ][2], u_lim[ORC_MAX_NQ][2];",
           "} OrcModel;", ""]
    for m in models:
        nq = m["nq"]
        pad = 7 - nq
        E0 = [m["E0"][k].reshape(9) for k in range(nq)] + [np.zeros(9)] * pad
        r = list(m["r"]) + [np.zeros(3)] * pad
        I6 = [m["I"][k].T.reshape(36) for k in range(nq)] + [np.zeros(36)] * pad  # col-major
        ql = list(m["q_lim"]) + [np.zeros(2)] * pad
        vl = list(m["v_lim"]) + [np.zeros(2)] * pad
        ul = list(m["u_lim"]) + [np.zeros(2)] * pad
        out.append("static const OrcModel ORC_MODEL_%s = {" % m["name"].upper())
        out.append('    "%s", %d, %d,' % (m["name"], nq, m["barrier_mode"]))
        for arr in (E0, r, I6, ql, vl, ul):
            out.append("    " + carr(np.array(arr)) + ",")
        out.append("};")
        out.append("")
    out.append("#endif")
    open(path, "w").write("\n".join(out) + "\n")


def emit_hpp(models, path):
    out = ["// GENERATED by tools/gen_robot_models.py -- do not edit.",
           "// constexpr plant traits for the HIP kernels: every table is a compile-time constant so that the fully",
           "// unrolled per-lane recursions fold the 0 / +-1 entries of E0 and the structural zeros of I.",
           "#pragma once", "", "namespace gato {", ""]
    for m in models:
        nq = m["nq"]
        out.append("struct %s {" % m["name"].capitalize())
        out.append("    static constexpr int NQ = %d;" % nq)
        out.append("    static constexpr int BARRIER_MODE = %d;  // 0: indy7 clamp + outer-product Hessian, 1: iiwa14 signed clamp + diagonal Hessian" % m["barrier_mode"])
        out.append("    static constexpr float E0[%d][3][3] = %s;" % (nq, carr(m["E0"])))
        out.append("    static constexpr float R[%d][3] = %s;" % (nq, carr(m["r"])))
        out.append("    static constexpr float I[%d][6][6] = %s;  // [row][col]" % (nq, carr(m["I"])))
        out.append("    static constexpr float Q_LIM[%d][2] = %s;" % (nq, carr(m["q_lim"])))
        out.append("    static constexpr float V_LIM[%d][2] = %s;" % (nq, carr(m["v_lim"])))
        out.append("    static constexpr float U_LIM[%d][2] = %s;" % (nq, carr(m["u_lim"])))
        out.append("};")
        out.append("")
    out.append("}  // namespace gato")
    open(path, "w").write("\n".join(out) + "\n")


def main():
    models = [build(INDY7), build(IIWA14)]
    emit_c(models, os.path.join(ROOT, "oracle", "robot_tables.h"))
    emit_hpp(models, os.path.join(ROOT, "gato_amd", "csrc", "robot_models.hpp"))
    return models


if __name__ == "__main__":
    main()
    print("wrote oracle/robot_tables.h and gato_amd/csrc/robot_models.hpp")
