#!/bin/bash
# tuning aid: CBCA at KITTI size -- MC_CBCA_ABLATE 0 = full, 4 = window form only (no row-by-row fallback), 1 = minimal supports only
for a in 0 4 1; do MC_CBCA_ABLATE=$a python bench.py --config kitti_slow --steps 5 --warmup 1 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ablate $a kitti_slow cbca', j['stage_ms']['cbca'])"; done
