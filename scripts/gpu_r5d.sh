#!/bin/bash
# Round 5: the SGM schedule with the down sweep beside the horizontal ones (T) against round 4's three launches (R4), on ONE box; variants of the prefetch depths.
# Parity of T first (the whole -m gpu suite).  Output: gpurun_out/r5d/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
use() { cp gpurun_in/lib$1.so mc-cnn_amd/libmcadcensus.so; }
use T; MC_REQUIRE_REF=1 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_T.log 2>&1; echo "pytest(T) rc=$?"; tail -2 $O/pytest_T.log
for L in T8 TD16 TH4; do use $L; MC_REQUIRE_REF=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sample_pair.py -m gpu -x -q -k "sgm or predict or fast or sample" > $O/pytest_$L.log 2>&1; echo "pytest($L) rc=$?"; tail -1 $O/pytest_$L.log; done
line() { # lib config pair steps
  use $1
  timeout 300 python bench.py --config $2 ${3:+--pair $3} --steps $4 --warmup 2 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/ab_$1_$2_$3.json 2>/dev/null
  python -c "
import json; j=json.loads([l for l in open('$O/ab_$1_$2_$3.json') if l.startswith('{')][-1]); print('lib$1', '$2', '$3', j['ms_per_step'], j['ms_per_step_min'], {k: round(v, 3) for k, v in j['stage_ms'].items()})"
}
for rep in 1 2; do
  for L in R4 T T8 TD16 TH4; do line $L kitti_fast "" 30; done
  for L in R4 T T8 TD16; do line $L mb_slow "" 5; done
done 2>&1 | tee $O/ab.txt
for L in R4 T; do line $L kitti_slow "" 20; done 2>&1 | tee -a $O/ab.txt
# per-kernel times of the two schedules
cd /tmp && export TMPDIR=/tmp
for L in R4 T; do
  cp $GRAFT_REPO_ROOT/gpurun_in/lib$L.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o p -- python $GRAFT_REPO_ROOT/bench.py --config kitti_fast --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/prof_$L.log 2>&1
  f=$(ls $O/prof_$L/*kernel_stats.csv 2>/dev/null | head -1); echo "== $L"; head -8 $f | cut -c1-200
done
cd $GRAFT_REPO_ROOT; cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
