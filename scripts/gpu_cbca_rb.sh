#!/bin/bash
# tuning aid: rows per CBCA strip (MC_CBCA_RB overrides the automatic choice)
timeout 600 python -m pytest tests -m gpu -q -x -k "cbca or golden" 2>&1 | tail -1
for cfg in kitti_slow mb_slow; do python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$cfg cbca', j['stage_ms']['cbca'], j['ms_per_step'])"; done
