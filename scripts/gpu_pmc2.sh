#!/bin/bash
# usage: bash scripts/gpu_pmc2.sh <tag> <config> [steps]  -- instruction-mix / stall counters per kernel
TAG=$1; CFG=$2; STEPS=${3:-1}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc2_$CFG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-ref-gpu > $O/p$i.log 2>&1
  echo "pass $i rc=$? ($SET)"; tail -2 $O/p$i.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT && python scripts/pmc_summary.py $O | cut -c1-600 | head -6
