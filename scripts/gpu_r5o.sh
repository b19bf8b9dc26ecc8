#!/bin/bash
# Round 5: steps in flight of the horizontal SGM launch (4 / 8 = product / 16), per launch (rocprofv3), with the no-store / no-load ablations at 16.  Output: gpurun_out/r5o/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5o; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for L in P H16 H4 H16G1 H16G8 P H16; do
  cp $GRAFT_REPO_ROOT/gpurun_in/lib$L.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o p -- python $GRAFT_REPO_ROOT/bench.py --config kitti_fast --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/prof_$L.log 2>&1
  python3 -c "
import csv
t=[(r['Name'],float(r['AverageNs'])/1e3) for r in csv.DictReader(open('$O/prof_$L/p_kernel_stats.csv')) if 'sgm_pass' in r['Name']]
t.sort(key=lambda x: x[0]); print('$L', '  '.join('%s %.1f' % (n.split('<')[1][:1], v) for n, v in t))"
done 2>&1 | tee $O/ab.txt
cp /tmp/lib_keep.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
