#!/bin/bash
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_instantiations.py -m gpu -q -x -k "fused or nine" 2>&1 | tail -6
cd /tmp; export TMPDIR=/tmp
for c in mb_slow kitti_slow; do
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pp_$c -o pp -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_time.py $c > /tmp/log 2>&1; tail -2 /tmp/log | cut -c1-200
python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py /tmp/pp_$c/pp_results.db | cut -d'"' -f2,3 | cut -c1-60,75-130 | head -6
done
