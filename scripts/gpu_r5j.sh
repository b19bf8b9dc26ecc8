#!/bin/bash
# Round 5: what bounds StereoJoin (0.41 ms, MFMA pipe 45 % busy)?  Timing-only ablations of join_owner_kernel (-DMC_JOIN_DBG bits: 1 no volume stores, 2 no MFMAs,
# 4 no partner loads, 8 no ring writes; results are wrong by construction) on ONE box.  Output: gpurun_out/r5j/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for L in P D1 D2 D4 D8 D9 D3 D15; do
  cp $GRAFT_REPO_ROOT/gpurun_in/lib$L.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o p -- python $GRAFT_REPO_ROOT/bench.py --config kitti_fast --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/prof_$L.log 2>&1
  echo "== $L $(grep -i 'join_owner' $O/prof_$L/*kernel_stats.csv | cut -d, -f2-4,6-7)"
done 2>&1 | tee $O/ablation.txt
cp /tmp/lib_keep.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
