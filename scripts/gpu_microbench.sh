#!/bin/bash
# usage (on the GPU box via gpurun): bash scripts/gpu_microbench.sh -- parity at the benchmarked shapes + the bandwidth
# microbenchmarks of scripts/microbench (build them first: hipcc --offload-arch=gfx950 -O3 x.hip -o x.bin); outputs are
# what profiles/r02_bw_*.txt hold
O=$GRAFT_REPO_ROOT/gpurun_out/microbench; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x > $O/fullsize.log 2>&1; echo "rc=$?" >> $O/fullsize.log; tail -15 $O/fullsize.log
timeout 120 scripts/microbench/bw_sizes.bin > $O/bw_sizes.txt 2>&1; cat $O/bw_sizes.txt
timeout 120 scripts/microbench/bw_copy.bin 2.0 > $O/bw_copy.txt 2>&1; cat $O/bw_copy.txt
timeout 120 scripts/microbench/bw_patterns.bin > $O/bw_patterns.txt 2>&1; cat $O/bw_patterns.txt
timeout 200 scripts/microbench/bw_cbca.bin > $O/bw_cbca.txt 2>&1; cat $O/bw_cbca.txt
