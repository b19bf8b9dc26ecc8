#!/bin/bash
# bench lines of several builds (gpurun_in/lib<X>.so) on one box: LIBS="P B" CFGS="kitti_slow mb_slow" bash scripts/gpu_ab_bench.sh
O=gpurun_out/abb; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
for L in $LIBS $LIBS; do
  cp gpurun_in/lib$L.so mc-cnn_amd/libmcadcensus.so
  for c in $CFGS; do
    timeout 300 python bench.py --config $c ${PAIR:+--pair $PAIR} --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/$L_$c.json 2>/dev/null
    python -c "
import json; j=json.loads([l for l in open('$O/$L_$c.json') if l.startswith('{')][-1]); print('lib$L', '$c', j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms'].items()})"
  done
done
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
