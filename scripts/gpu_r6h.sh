#!/bin/bash
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r6h; mkdir -p $O
MC_REQUIRE_REF=1 timeout 900 python -m pytest tests/test_ref_parity.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sample_pair.py -x -q -k "join or Join or predict or sample or golden" 2>&1 | tail -3
python scripts/gpu_ops_route.py ops 5 | tail -1
python scripts/gpu_ops_route.py fused 5 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/ops -o ops -- python $GRAFT_REPO_ROOT/scripts/gpu_ops_route.py ops 5 > $O/ops.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/rocpd_overlap.py $O/ops/ops_results.db --tail-ms 27 --top 30 > $O/ops_route.txt; cat $O/ops_route.txt
