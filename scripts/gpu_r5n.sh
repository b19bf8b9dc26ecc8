#!/bin/bash
# Round 5: cache policy of the SGM sweeps' volume stores (aux bits of the buffer instructions: 0 default, 1 sc0, 2 nt -- the product --, 3 both; L0: loads AND
# stores default) on ONE box.  Output: gpurun_out/r5n/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5n; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
use() { cp gpurun_in/lib$1.so mc-cnn_amd/libmcadcensus.so; }
line() { use $1
  timeout 300 python bench.py --config $2 --steps $3 --warmup 2 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/ab_$1_$2.json 2>/dev/null
  python -c "
import json; j=json.loads([l for l in open('$O/ab_$1_$2.json') if l.startswith('{')][-1]); print('lib$1', '$2', j['ms_per_step'], j['ms_per_step_min'], {k: round(v, 3) for k, v in j['stage_ms'].items()})"
}
for rep in 1 2; do for L in P A0 A1 A3 L0; do line $L kitti_fast 30; done; for L in P A0 L0; do line $L mb_slow 5; done; done 2>&1 | tee $O/ab.txt
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
