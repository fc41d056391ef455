#!/bin/bash
# A/B of a kernel build flag: build the other library first, e.g.
#   cd gato_amd/csrc && hipcc $(CXXFLAGS) -DGATO_SCHUR1_STAGE=0 -shared -o ../../tools/exp/libgato_nostage.so solver.hip -ldl
# then: gpurun -- bash tools/schur1_ab.sh   (bit comparison of one solve per library, then alternating bench runs)
cd $GRAFT_REPO_ROOT
python tools/dump_solve.py /tmp/s1_stage.npz
GATO_HIP_LIB=$PWD/tools/exp/libgato_nostage.so python tools/dump_solve.py /tmp/s1_nostage.npz
python - <<'PY'
import numpy as np
a=np.load("/tmp/s1_stage.npz"); b=np.load("/tmp/s1_nostage.npz")
for k in a.files:
    same = a[k].tobytes()==b[k].tobytes()
    print(k, "SAME" if same else "DIFF", a[k].shape)
PY
bench() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_us_per_solve'], d['solution_ok'])"; }
for rep in 1 2; do
for lib in stage nostage; do
  if [ $lib = nostage ]; then export GATO_HIP_LIB=$PWD/tools/exp/libgato_nostage.so; else unset GATO_HIP_LIB; fi
  echo "== $lib C5"; bench --workload hparam --plant iiwa14 --knots 64 --batch 512 --steps 30 --warmup 3
  echo "== $lib C3"; bench --plant iiwa14 --knots 128 --batch 256 --steps 30 --warmup 3
done; done
unset GATO_HIP_LIB
python -m pytest tests/test_gpu_parity.py tests/test_f64_gpu.py -x -q 2>&1 | tail -3
