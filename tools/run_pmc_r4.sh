#!/bin/bash
# Round 4: the PMC passes of round 3 (profiles/r3_cold_solve_pmc.json) over ONE cold solve at N = 144, for the role-separated K3
# (k_jacobi_solve_v2, the default) and, on the same box, the two-barrier K3 (VINTERP_K3=v1).  Counters in their own runs, no
# trace flags.  Run from the repository root on the GPU box:  bash tools/run_pmc_r4.sh
set -e
P1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES"
P2="SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY"
ROOT=$(pwd)
mkdir -p gpurun_out/pmc4
cd /tmp && export TMPDIR=/tmp
for k in v2 v1; do
  if [ $k = v1 ]; then export VINTERP_K3=v1; else unset VINTERP_K3; fi
  rocprofv3 --pmc $P1 --output-format csv -d $ROOT/gpurun_out/pmc4/${k}_a -o a -- python3 $ROOT/tools/pmc_cold.py > $ROOT/gpurun_out/pmc4/${k}_a.log 2>&1
  rocprofv3 --pmc $P2 --output-format csv -d $ROOT/gpurun_out/pmc4/${k}_b -o b -- python3 $ROOT/tools/pmc_cold.py > $ROOT/gpurun_out/pmc4/${k}_b.log 2>&1
  echo "== $k"
  python3 $ROOT/tools/pmc_json.py $ROOT/gpurun_out/pmc4/cold_solve_$k.json $(find $ROOT/gpurun_out/pmc4/${k}_a $ROOT/gpurun_out/pmc4/${k}_b -name "*counter_collection.csv")
done
