#!/bin/bash
# tuning aid: per-kernel times of the small kernels (rocprofv3 kernel trace) for two configs
O=$GRAFT_REPO_ROOT/gpurun_out/profpost; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in kitti_fast kitti_slow; do
rm -rf $O/p
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p -o k -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-ref-gpu > $O/log 2>&1
echo $cfg; python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $O/p/k_results.db | python -c "import csv,sys; [print('  ', r[0][:70], r[1], r[3]) for r in csv.reader(sys.stdin) if 'mc::' in r[0] and 'sgm_pass' not in r[0] and 'cbca_strip' not in r[0]]"
done
