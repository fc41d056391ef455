#!/bin/bash
# Runs on the MI355X box (via gpurun): kernel-trace stats and the PMC passes behind profiles/<tag>_* and profiles/pmc_summary.json.
#   tools/profile_round.sh r02a            -> gpurun_out/prof_<tag>_{c2,c3,c5}/..., gpurun_out/pmc_<tag>_{cfg}_{set}/...
# Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -u
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
declare -A CMD
CMD[c2]="$ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"   # the driver's own invocation (K = 20, W = 3)
CMD[c3]="$ROOT/bench.py --plant iiwa14 --knots 128 --batch 256 --steps 20 --warmup 3 --no-cpu-baseline"
CMD[c5]="$ROOT/bench.py --workload hparam --plant iiwa14 --knots 64 --batch 512 --steps 20 --warmup 3 --no-cpu-baseline"
# direct mode at a long horizon and a small batch: the cyclic-reduction kernel (the one place MFMA runs)
CMD[cr]="$ROOT/tools/gpu_time.py --plant indy7 -N 128 -B 8 --iters 1 --reps 20 --solver direct"
for cfg in c2 c3 c5 cr; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_${cfg} -o k -- python3 ${CMD[$cfg]} > $OUT/prof_${TAG}_${cfg}.log 2>&1
done
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE GRBM_COUNT")
for cfg in c2 c3 c5 cr; do
  i=0
  for set in "${SETS[@]}"; do
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_${TAG}_${cfg}_$i -o p -- python3 ${CMD[$cfg]} > $OUT/pmc_${TAG}_${cfg}_$i.log 2>&1
    i=$((i+1))
  done
done
ls $OUT | grep "${TAG}" | head -40
