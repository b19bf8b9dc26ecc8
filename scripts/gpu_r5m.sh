#!/bin/bash
# Round 5: what does a step of the SGM sweeps wait for?  Timing-only ablations (-DMC_SGM_DBG bits: 1 no volume stores, 2 no window-class loads, 4 no
# reference-class loads, 8 no cost loads, 16 no wave-wide minimum; results wrong by construction) on ONE box, per launch.  Output: gpurun_out/r5m/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for L in P G1 G2 G4 G6 G8 G16 G7 G31; do
  cp $GRAFT_REPO_ROOT/gpurun_in/lib$L.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o p -- python $GRAFT_REPO_ROOT/bench.py --config kitti_fast --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/prof_$L.log 2>&1
  python3 -c "
import csv
t=[(r['Name'],float(r['AverageNs'])/1e3) for r in csv.DictReader(open('$O/prof_$L/p_kernel_stats.csv')) if 'sgm_pass' in r['Name']]
t.sort(key=lambda x: x[0]); print('$L', '  '.join('%s %.1f' % (n.split('<')[1][:1], v) for n, v in t))"
done 2>&1 | tee $O/ablation.txt
cp /tmp/lib_keep.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
