#!/bin/bash
# Round 5, tile kernel A/B on ONE box (gpurun_in/lib<X>.so): P product, E per-chunk event-row mask, M taps on v_mfma_f32_4x4x1_16b_f32,
# N events as four compares then eight selects (no wait states) + E.  Parity of each new build first.  Output: gpurun_out/r5b/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
use() { cp gpurun_in/lib$1.so mc-cnn_amd/libmcadcensus.so; }
for L in ${CHECK:-M N}; do use $L; MC_REQUIRE_REF=1 timeout 500 python -m pytest tests/test_gpu_cbca_tile.py tests/test_gpu_planned_routes.py tests/test_gpu_sample_pair.py tests/test_gpu_instantiations.py -m gpu -x -q > $O/pytest_$L.log 2>&1; echo "pytest($L) rc=$?"; tail -1 $O/pytest_$L.log; done
line() { # lib config pair steps
  use $1
  timeout 300 python bench.py --config $2 ${3:+--pair $3} --steps $4 --warmup 2 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/ab_$1_$2_$3.json 2>/dev/null
  python -c "
import json; j=json.loads([l for l in open('$O/ab_$1_$2_$3.json') if l.startswith('{')][-1]); print('lib$1', '$2', '$3', j['ms_per_step'], j['ms_per_step_min'], {k: round(v, 3) for k, v in j['stage_ms'].items()})"
}
for rep in 1 2; do
  for L in ${LIBS:-P E M N}; do line $L mb_slow natural 3; done
  for L in ${LIBS:-P E M N}; do line $L kitti_slow "" 20; done
done 2>&1 | tee $O/ab.txt
for L in ${LIBS:-P E M N}; do line $L mb_slow sample 3; done 2>&1 | tee -a $O/ab.txt
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
