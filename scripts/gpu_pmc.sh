#!/bin/bash
# usage: bash scripts/gpu_pmc.sh <tag> <config> [steps] [pair]
# HBM-traffic and stall counters per kernel: separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE
# needs 3 TCC slots, WRITE_SIZE 2 -> one pass each), kernel-trace only, CSV output.
TAG=$1; CFG=$2; STEPS=${3:-3}; PAIR=${4:-}
NAME=$CFG${PAIR:+_$PAIR}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --config $CFG ${PAIR:+--pair $PAIR} --steps $STEPS --warmup 1 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/p$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O > $O/../${NAME}_pmc.csv; python scripts/make_traffic_json.py $O $NAME
