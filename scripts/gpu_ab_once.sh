#!/bin/bash
# bench lines (stage times) of several builds (gpurun_in/lib<X>.so) on one box, one round: LIBS="A B" CFG=mb_slow bash scripts/gpu_ab_once.sh
O=gpurun_out/abo; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
for L in $LIBS; do
  cp gpurun_in/lib$L.so mc-cnn_amd/libmcadcensus.so
  timeout 200 python bench.py --config ${CFG:-mb_slow} ${PAIR:+--pair $PAIR} --steps ${STEPS:-5} --warmup 1 --min-seconds 0.3 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/$L.json 2>/dev/null
  python -c "
import json; j=json.loads([l for l in open('$O/$L.json') if l.startswith('{')][-1]); print('lib$L', j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms'].items()})"
done
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
