#!/bin/bash
# C5 / C3 PCG variants, timed with bench.py (rank 0, 1 GPU); run on the box
cd $GRAFT_REPO_ROOT
for v in default 4; do
  if [ $v = default ]; then unset GATO_PCG_VARIANT; else export GATO_PCG_VARIANT=$v; fi
  echo "== C5 variant $v"; python bench.py --workload hparam --plant iiwa14 --knots 64 --batch 512 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['stage_us_per_solve'], d['solution_ok'])"
done
unset GATO_PCG_VARIANT
echo "== C3"; python bench.py --plant iiwa14 --knots 128 --batch 256 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['stage_us_per_solve'], d['solution_ok'])"
echo "== C2"; python bench.py --steps 50 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['stage_us_per_solve'], d['solution_ok'])"
