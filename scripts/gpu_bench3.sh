#!/bin/bash
# tuning aid: parity of the fused paths + the three single-GPU bench configs, stage times only
timeout 600 python -m pytest tests -m gpu -q -x -k "golden or predict or sgm or cbca" 2>&1 | tail -1
for cfg in kitti_fast kitti_slow mb_slow; do
python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$cfg', j['ms_per_step'], j['stage_ms'])"; done
