#!/bin/bash
# SGM prefetch-depth sweep (tuning aid): per-kernel average time of the three sweeps per (UH, UD, UU), rocprofv3 kernel trace
O=$GRAFT_REPO_ROOT/gpurun_out/sgmsweep; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for combo in "4 16 4" "8 16 4" "4 16 8" "4 8 4" "16 16 4"; do
  set -- $combo
  rm -rf $O/p
  MC_SGM_UH=$1 MC_SGM_UD=$2 MC_SGM_UU=$3 timeout 300 rocprofv3 --kernel-trace --stats -d $O/p -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-gpu > $O/log 2>&1
  echo "UH=$1 UD=$2 UU=$3"; python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $O/p/k_results.db | python -c "import csv,sys; [print(r[0][:60], r[1], r[3]) for r in csv.reader(sys.stdin) if 'sgm_pass' in r[0]]"
done
