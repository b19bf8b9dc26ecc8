#!/bin/bash
# round 2: CBCA plane-slab experiment (Infinity-Cache blocking of the iterations)
O=$GRAFT_REPO_ROOT/gpurun_out/r2b; mkdir -p $O
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
run off MC_CBCA_SLAB_MB=0
run s48 MC_CBCA_SLAB_MB=48 MC_CBCA_NT=0
run s96 MC_CBCA_SLAB_MB=96 MC_CBCA_NT=0
run s128 MC_CBCA_SLAB_MB=128 MC_CBCA_NT=0
run s96nt MC_CBCA_SLAB_MB=96 MC_CBCA_NT=1
run s24 MC_CBCA_SLAB_MB=24 MC_CBCA_NT=0
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_ref_parity.py -m gpu -q -x 2>&1 | tail -3
