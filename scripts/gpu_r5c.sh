#!/bin/bash
# Round 5: the SGM schedule microbenchmark (scripts/microbench/bw_sgm_sched.hip) at both sizes + the -m gpu suite.  Output: gpurun_out/r5c/.
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
timeout 200 scripts/microbench/bw_sgm_sched.bin 370 1226 228 > $O/bw_sgm_sched_kitti.txt 2>&1; cat $O/bw_sgm_sched_kitti.txt
timeout 200 scripts/microbench/bw_sgm_sched.bin 1000 1500 256 > $O/bw_sgm_sched_mb.txt 2>&1; cat $O/bw_sgm_sched_mb.txt
MC_REQUIRE_REF=1 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
