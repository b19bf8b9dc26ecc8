#!/bin/bash
# Round 5: the whole-run transpose (X) against transpose4 (P = -DMC_TRANSPOSE_NO_RUNS) on ONE box; parity first.  Output: gpurun_out/r5h/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
MC_REQUIRE_REF=1 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
use() { cp gpurun_in/lib$1.so mc-cnn_amd/libmcadcensus.so; }
line() { use $1
  timeout 300 python bench.py --config $2 ${3:+--pair $3} --steps $4 --warmup 2 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/ab_$1_$2_$3.json 2>/dev/null
  python -c "
import json; j=json.loads([l for l in open('$O/ab_$1_$2_$3.json') if l.startswith('{')][-1]); print('lib$1', '$2', '$3', j['ms_per_step'], j['ms_per_step_min'], {k: round(v, 3) for k, v in j['stage_ms'].items()})"
}
for rep in 1 2 3; do for L in P X; do line $L kitti_slow "" 20; done; done 2>&1 | tee $O/ab.txt
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_X -o p -- python $GRAFT_REPO_ROOT/bench.py --config kitti_slow --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/prof_X.log 2>&1
grep -i "transpose" $O/prof_X/*kernel_stats.csv | cut -c1-220
