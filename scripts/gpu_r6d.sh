#!/bin/bash
# round 6, step d: pairs in flight (K streams x K workspaces), then the full default line
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r6d; mkdir -p $O
timeout 600 python bench.py --pairs-in-flight 3 --config kitti_fast > $O/pipe_kitti_fast.json 2> $O/pipe_kitti_fast.err; echo "rc=$?"; tail -c 1500 $O/pipe_kitti_fast.json
timeout 600 python bench.py --pairs-in-flight 3 --config kitti_slow > $O/pipe_kitti_slow.json 2> $O/pipe_kitti_slow.err; echo "rc=$?"; tail -c 1500 $O/pipe_kitti_slow.json
timeout 900 python bench.py --pairs-in-flight 2 --config mb_slow > $O/pipe_mb_slow.json 2> $O/pipe_mb_slow.err; echo "rc=$?"; tail -c 1500 $O/pipe_mb_slow.json
