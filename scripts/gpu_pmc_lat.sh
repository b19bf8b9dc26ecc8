#!/bin/bash
# usage: bash scripts/gpu_pmc_lat.sh <tag> <config> [steps]   (on the GPU box)
# Memory-latency / translation / matrix-core counters per kernel ("TCC_EA read latency / UTCL1 misses per launch shape"): separate
# rocprofv3 --pmc passes, kernel-trace only.  EA read latency = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ (cycles a read request is outstanding at the HBM side).
TAG=$1; CFG=$2; STEPS=${3:-3}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmclat_$CFG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps $STEPS --warmup 1 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/p$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O > $O/../${CFG}_pmclat.csv; head -c 1500 $O/../${CFG}_pmclat.csv
