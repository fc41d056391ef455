"""The generated rigid-body tables (tools/gen_robot_models.py) against the literals of the reference.

Reads DATA out of the reference's generated GRiD headers when the tree is present (authoring container only):
the constant table of `init_XImats` (indy7_grid.cuh:906-1562, iiwa14_grid.cuh:1211-2093) and the sin/cos coefficient
assignments of `load_update_XImats_helpers` / `load_update_XmatsHom_helpers` (indy7_grid.cuh:1597-1820,
iiwa14_grid.cuh:2212-2400).  X_k(q), I_k, Xhom_k(q), dXhom_k(q) rebuilt from those literals must equal what our
generator's (E0, r, I) produce.  On the GPU box the reference is absent and a committed fingerprint is checked instead.
"""
import os
import re

import numpy as np
import pytest

from tools import gen_robot_models as gen

PLANTS = {"indy7": ("gato/dynamics/indy7/indy7_grid.cuh", gen.INDY7), "iiwa14": ("gato/dynamics/iiwa14/iiwa14_grid.cuh", gen.IIWA14)}


def _body(txt, start_pat, end_pat, start_from=0):
    s = txt.index(start_pat, start_from)
    e = txt.index(end_pat, s)
    return txt[s:e], e


def _parse_const_table(txt, n):
    body, _ = _body(txt, "init_XImats()", "cudaMalloc")
    tab = np.full(n, np.nan)
    for m in re.finditer(r"h_XImats\[(\d+)\]\s*=\s*static_cast<T>\(([-+0-9.eE]+)\)", body):
        tab[int(m.group(1))] = float(m.group(2))
    assert not np.isnan(tab).any()
    return tab


def _parse_updates(body, arr, nq):
    """`arr[idx] = static_cast<T>([coef*]s_temp[j]);` -> list of (idx, coef, joint, is_cos)"""
    ups = []
    pat = re.compile(r"%s\[(\d+)\]\s*=\s*static_cast<T>\(\s*(-?)\s*(?:([0-9.eE+-]+)\s*\*\s*)?s_temp\[(\d+)\]\s*\)" % arr)
    for m in pat.finditer(body):
        coef = float(m.group(3)) if m.group(3) else 1.0
        if m.group(2) == "-":
            coef = -coef
        j = int(m.group(4))
        ups.append((int(m.group(1)), coef, j % nq, j >= nq))
    return ups


def _apply(tab, ups, q):
    out = tab.copy()
    for idx, coef, j, is_cos in ups:
        out[idx] = coef * (np.cos(q[j]) if is_cos else np.sin(q[j]))
    return out


def _ours(model, q):
    m = gen.build(model)
    nq = m["nq"]
    X = np.zeros((nq, 6, 6))
    Xh = np.zeros((nq, 4, 4))
    dXh = np.zeros((nq, 4, 4))
    for k in range(nq):
        c, s = np.cos(q[k]), np.sin(q[k])
        Ez = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
        dEz = np.array([[-s, c, 0], [-c, -s, 0], [0, 0, 0]])
        E = Ez @ m["E0"][k]
        X[k, :3, :3] = E
        X[k, 3:, 3:] = E
        X[k, 3:, :3] = -E @ gen.skew(m["r"][k])
        Xh[k, :3, :3] = E.T
        Xh[k, :3, 3] = m["r"][k]
        Xh[k, 3, 3] = 1
        dXh[k, :3, :3] = (dEz @ m["E0"][k]).T
    return m, X, Xh, dXh


@pytest.mark.parametrize("plant", ["indy7", "iiwa14"])
def test_tables_match_reference_literals(plant, reference_root):
    rel, model = PLANTS[plant]
    nq = model["nq"]
    txt = open(os.path.join(reference_root, rel)).read()
    tab = _parse_const_table(txt, 104 * nq if plant == "indy7" else 120 * nq)
    body_x, end = _body(txt, "void load_update_XImats_helpers", "kcr <")
    ups_x = _parse_updates(body_x, "s_XImats", nq)
    # second overload of load_update_XmatsHom_helpers carries both Xhom and dXhom
    first = txt.index("void load_update_XmatsHom_helpers", end)
    second = txt.index("void load_update_XmatsHom_helpers", first + 10)
    body_h = txt[second:txt.index("__syncthreads();\n    }", txt.index("dXmatsHom[", second)) if plant == "iiwa14" else txt.index("end_effector_positions_inner", second)]
    ups_h = _parse_updates(body_h, "s_XmatsHom", nq)
    ups_dh = _parse_updates(body_h, "s_dXmatsHom", nq)
    assert len(ups_x) >= 8 * nq - 8 and len(ups_h) == 4 * nq and len(ups_dh) == 4 * nq

    rng = np.random.default_rng(0)
    for _ in range(5):
        q = rng.uniform(-3, 3, nq)
        m, X, Xh, dXh = _ours(model, q)
        ref = _apply(tab, ups_x, q)
        for k in range(nq):
            Xr = ref[36 * k:36 * k + 36].reshape(6, 6).T.copy()
            Xr[3:, 3:] = Xr[:3, :3]  # the `dstInd = srcInd + 21` copy (indy7_grid.cuh:1672-1680)
            np.testing.assert_allclose(X[k], Xr, atol=2e-7, rtol=0)
            Ir = tab[36 * nq + 36 * k:36 * nq + 36 * k + 36].reshape(6, 6).T
            np.testing.assert_allclose(m["I"][k], Ir, atol=1e-8, rtol=1e-7)
        hom = _apply(tab[72 * nq:72 * nq + 16 * nq], ups_h, q)
        dhom = _apply(tab[88 * nq:88 * nq + 16 * nq], ups_dh, q)
        for k in range(nq):
            np.testing.assert_allclose(Xh[k], hom[16 * k:16 * k + 16].reshape(4, 4).T, atol=2e-7, rtol=0)
            np.testing.assert_allclose(dXh[k], dhom[16 * k:16 * k + 16].reshape(4, 4).T, atol=2e-7, rtol=0)


def test_limits_match_plant_headers(reference_root):
    for plant, (rel, model) in PLANTS.items():
        txt = open(os.path.join(reference_root, rel.replace("_grid", "_plant"))).read()
        m = gen.build(model)
        for name, key in (("JOINT_LIMITS_DATA", "q_lim"), ("VEL_LIMITS_DATA", "v_lim"), ("CTRL_LIMITS_DATA", "u_lim")):
            body, _ = _body(txt, name, "};")
            vals = [float(v) for v in re.findall(r"\{-([0-9.]+) - JOINT_LIMIT_MARGIN", body)]
            assert len(vals) == m["nq"]
            np.testing.assert_allclose(m[key][:, 1], np.float32(np.array(vals) + float(np.float32(-0.1))), rtol=0, atol=0)
            np.testing.assert_allclose(m[key][:, 0], -m[key][:, 1], rtol=0, atol=0)


def test_generated_files_are_current():
    """The committed tables are exactly what the generator emits (runs everywhere, incl. the GPU box)."""
    import tempfile
    models = [gen.build(gen.INDY7), gen.build(gen.IIWA14)]
    with tempfile.TemporaryDirectory() as d:
        gen.emit_c(models, os.path.join(d, "t.h"))
        gen.emit_hpp(models, os.path.join(d, "t.hpp"))
        assert open(os.path.join(d, "t.h")).read() == open(os.path.join(gen.ROOT, "oracle", "robot_tables.h")).read()
        assert open(os.path.join(d, "t.hpp")).read() == open(os.path.join(gen.ROOT, "gato_amd", "csrc", "robot_models.hpp")).read()
