"""Register budgets of the hot kernels, checked at build time (hipcc -Rpass-analysis=kernel-resource-usage cross-compiles without a GPU).
Occupancy is part of the design: the fused PCG kernel holds four trajectories per CU only below 256 registers
without AGPR copies or scratch, the step kernel is resident for all 1024 workgroups only at <= 128 -- a refactor that costs a few
registers shows up as a 40 % slower launch on the GPU box (it happened: ext_vector aliases for float4 took pcgs_kernel from 36 to 340
bytes of scratch per lane, C3 416 -> 581 us per launch), so it is pinned here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gato_amd", "csrc")

# kernel (regex on the mangled name) -> (max VGPRs, max AGPRs, max scratch bytes per lane)
BUDGETS = {
    r"pcgc_kernelINS_5Indy7ELi3ELi128ELb1ELb1ELb0ELb1E": (256, 0, 0),      # C2's PCG AS LAUNCHED since round 6 (N = 32: exactly two wavefronts, block_sum<.., TWO>): four trajectories per CU
    r"pcgc_kernelINS_5Indy7ELi3ELi256ELb1ELb1ELb0ELb1E": (256, 0, 0),      # the same kernel for N = 64: fused Schur + fold, single-lane form, every thread owns rows (N a multiple of 16)
    r"pcgc_kernelINS_5Indy7ELi3ELi256ELb1ELb1ELb0ELb0E": (256, 16, 0),     # ... its masked form (N = 4, 8): a few AGPR copies since the body became a device function (round 4), no scratch
    r"pcgc_kernelINS_5Indy7ELi3ELi256ELb1ELb1ELb1ELb[01]E": (256, 0, 0),   # pair form
    r"pcgs_kernelINS_6Iiwa14ELi512ELb1E": (256, 0, 64),                 # C3's PCG (symmetric half storage, fold)
    r"pcgc_kernelINS_6Iiwa14ELi2ELi512ELb1ELb0ELb0E": (256, 0, 0),     # C5's PCG
    r"kkt_kernelINS_5Indy7E": (256, 0, 0),
    r"kkt_kernelINS_6Iiwa14E": (256, 0, 0),                            # 4 wavefronts per workgroup (columns {0,5,6} {1,3} {2,4} + costs): two workgroups per CU
    r"step_kernelINS_5Indy7ELi512E": (128, 0, 0),                        # 4 wavefronts per SIMD: every C2 workgroup resident
    r"step_kernelINS_6Iiwa14ELi(512|1024)E": (128, 0, 64),                # iiwa14: 44 bytes of scratch since the row-per-lane dz (160 with dz_knot)
    r"pcgc_kernelINS_6Iiwa14ELi2ELi512ELb1ELb0ELb0ELb1E": (240, 0, 0),   # C5's PCG as launched (FULL): 231 registers since the single-exit loop of round 6 (all window reads of a
                                                                         # product in flight at once; 204 before); 7 wavefronts on 4 SIMDs: two per SIMD = 256 is the bound that matters
    r"schur1_kernelINS_6Iiwa14ELb0E": (128, 0, 0),                       # 4 wavefronts per SIMD (LDS allows no more); with the staged block stores it needs the pin on the
                                                                         # theta columns (kernels.hpp) or the allocator takes 256 + 132
    r"btd_cr_kernelINS_\w+ELi16E": (128, 0, 0),                          # cyclic reduction, 16 wavefronts per workgroup: 1024 threads need <= 128
}


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_hot_kernels_stay_inside_their_register_budgets():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize", "-c", "-o", "/dev/null",
                          "solver.hip", "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                res[cur][key] = int(m.group(1))
    for pat, (vg, ag, sc) in BUDGETS.items():
        hits = [(k, v) for k, v in res.items() if re.search(pat, k)]
        assert hits, "kernel %s not found in the build" % pat
        for k, v in hits:
            assert v["vgpr"] <= vg and v["agpr"] <= ag and v["scratch"] <= sc, (k, v, (vg, ag, sc))
