#!/bin/bash
# round 2, first call: parity at the benchmarked shapes + bandwidth microbenchmarks (outputs kept under profiles/)
O=$GRAFT_REPO_ROOT/gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x > $O/fullsize.log 2>&1; echo "rc=$?" >> $O/fullsize.log; tail -15 $O/fullsize.log
timeout 120 scripts/microbench/bw_sizes.bin > $O/bw_sizes.txt 2>&1; cat $O/bw_sizes.txt
timeout 120 scripts/microbench/bw_copy.bin 2.0 > $O/bw_copy.txt 2>&1; cat $O/bw_copy.txt
timeout 120 scripts/microbench/bw_patterns.bin > $O/bw_patterns.txt 2>&1; cat $O/bw_patterns.txt
