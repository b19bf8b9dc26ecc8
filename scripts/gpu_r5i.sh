#!/bin/bash
# Round 5: StereoJoin with the line writer and the next partner tile's loads issued between the MFMAs of a chain (J3 = -DMC_JOIN_ORDER=3) against the
# product (P) on ONE box; parity of J3 first.  Output: gpurun_out/r5i/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
use() { cp gpurun_in/lib$1.so mc-cnn_amd/libmcadcensus.so; }
use J3; MC_REQUIRE_REF=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_ref_parity.py tests/test_gpu_sample_pair.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -m gpu -x -q -k "join or Join or fast or sample or predict or golden" > $O/pytest_J3.log 2>&1; echo "pytest(J3) rc=$?"; tail -2 $O/pytest_J3.log
line() { use $1
  timeout 300 python bench.py --config $2 --steps $3 --warmup 2 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/ab_$1_$2.json 2>/dev/null
  python -c "
import json; j=json.loads([l for l in open('$O/ab_$1_$2.json') if l.startswith('{')][-1]); print('lib$1', '$2', j['ms_per_step'], j['ms_per_step_min'], {k: round(v, 3) for k, v in j['stage_ms'].items()})"
}
for rep in 1 2 3; do for L in P J3; do line $L kitti_fast 30; done; done 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for L in P J3; do
  cp $GRAFT_REPO_ROOT/gpurun_in/lib$L.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o p -- python $GRAFT_REPO_ROOT/bench.py --config kitti_fast --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/prof_$L.log 2>&1
  echo "== $L"; grep -i "join" $O/prof_$L/*kernel_stats.csv | cut -c1-200
done
cd $GRAFT_REPO_ROOT; cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
