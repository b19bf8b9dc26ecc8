#!/bin/bash
# join kernel check: parity tests that involve the feature path + kitti_fast bench (verify on)
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/join; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "join or fast or feat or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench.py --config kitti_fast --steps 30 --warmup 3 --no-cpu-baseline --no-north-star --no-ops > $O/bench.json 2> $O/bench.err
python - <<PY
import json
for l in open("$O/bench.json"):
    if l.startswith("{"):
        j=json.loads(l); print(j["ms_per_step"], j["stage_ms"], j["roofline"]["kernels"], j["verify"]["bit_exact"], j["roofline"].get("box_copy"))
PY
