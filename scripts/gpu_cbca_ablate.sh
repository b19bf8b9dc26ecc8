#!/bin/bash
# tuning aid: CBCA parity subset + the two CBCA-heavy bench configs; MC_CBCA_ABLATE=1 skips the larger supports
timeout 600 python -m pytest tests -m gpu -q -x -k "cbca or golden or predict" 2>&1 | tail -1
for cfg in mb_slow kitti_slow; do
python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$cfg', j['ms_per_step'], j['stage_ms'])"; done
