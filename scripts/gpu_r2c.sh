#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2c; mkdir -p $O; ulimit -c 0
run() { # name env...
  name=$1; shift
  for cfg in mb_slow kitti_slow; do
    steps=3; [ $cfg = kitti_slow ] && steps=20
    env "$@" timeout 300 python bench.py --config $cfg --steps $steps --warmup 1 --no-cpu-baseline --no-ref-gpu > $O/${name}_$cfg.json 2> $O/${name}_$cfg.err
    python - <<PY
import json
try:
    j=json.load(open("$O/${name}_$cfg.json")); print("$name $cfg", j["ms_per_step"], j["stage_ms"])
except Exception as e: print("$name $cfg failed", e); print(open("$O/${name}_$cfg.err").read()[-1500:])
PY
  done
}
run v2 MC_CBCA_SLAB_MB=0
run v2nt0 MC_CBCA_SLAB_MB=0 MC_CBCA_NT=0
run v2s96 MC_CBCA_SLAB_MB=96 MC_CBCA_NT=0
run v2s192 MC_CBCA_SLAB_MB=192 MC_CBCA_NT=0
run v1 MC_CBCA_SLAB_MB=0 MC_CBCA_V2=0
