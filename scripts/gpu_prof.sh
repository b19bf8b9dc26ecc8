#!/bin/bash
# usage: bash scripts/gpu_prof.sh <tag> <config> [steps] [pair]  -- rocprofv3 kernel trace of bench.py on the GPU box
TAG=$1; CFG=$2; STEPS=${3:-10}; PAIR=${4:-}
NAME=$CFG${PAIR:+_$PAIR}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$NAME -o $NAME -- python $GRAFT_REPO_ROOT/bench.py --config $CFG ${PAIR:+--pair $PAIR} --steps $STEPS --warmup 2 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/prof_$NAME.log 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py $O/prof_$NAME/${NAME}_results.db > $O/${NAME}_kernel_stats.csv && cut -d, -f1-4,7,8 $O/${NAME}_kernel_stats.csv | head -12
