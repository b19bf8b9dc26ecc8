#!/bin/bash
# round 6, step g: kernel traces -- which kernels co-run with two pairs in flight; busy time against host gaps on the op-by-op route
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $O/pipe2 -o pipe2 -- python $R/bench.py --pairs-in-flight 2 --config kitti_fast > $O/pipe2.log 2>&1; echo "rc=$?"
timeout 600 rocprofv3 --kernel-trace -d $O/pipe2s -o pipe2s -- python $R/bench.py --pairs-in-flight 2 --config kitti_slow > $O/pipe2s.log 2>&1; echo "rc=$?"
timeout 600 rocprofv3 --kernel-trace -d $O/ops -o ops -- python $R/scripts/gpu_ops_route.py ops 5 > $O/ops.log 2>&1; echo "rc=$?"
timeout 600 rocprofv3 --kernel-trace -d $O/fused -o fused -- python $R/scripts/gpu_ops_route.py fused 5 > $O/fused.log 2>&1; echo "rc=$?"
cd $R
tail -1 $O/ops.log; tail -1 $O/fused.log
# the K = 2 blocks are the last thing bench.py --pairs-in-flight 2 runs: 5 blocks x 24 pairs ~ 290 ms
python scripts/rocpd_overlap.py $O/pipe2/pipe2_results.db --tail-ms 250 > $O/pipe2_overlap.txt; cat $O/pipe2_overlap.txt
python scripts/rocpd_overlap.py $O/pipe2s/pipe2s_results.db --tail-ms 500 > $O/pipe2s_overlap.txt; cat $O/pipe2s_overlap.txt
python scripts/rocpd_overlap.py $O/ops/ops_results.db --tail-ms 58 --top 30 > $O/ops_route.txt; cat $O/ops_route.txt
python scripts/rocpd_overlap.py $O/fused/fused_results.db --tail-ms 12.5 --top 20 > $O/fused_route.txt; cat $O/fused_route.txt
