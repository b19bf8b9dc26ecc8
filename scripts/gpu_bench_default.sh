#!/bin/bash
# the driver's default line, on the GPU box: bash scripts/gpu_bench_default.sh <tag>
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
SECONDS=0; timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$? wall=${SECONDS}s"
python - <<PY
import json
l=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms", l["ms_per_step"], "roof", l["roofline"]["frac"], "verify", l["verify"]["bit_exact"] if l.get("verify") else None)
f=l.get("kitti_fast_from_images"); print("from_images", json.dumps(f)[:1500])
p=l.get("pipelined"); print("pipelined", json.dumps(p)[:2500])
k=l.get("kitti_accurate"); print("kacc", k and k.get("ms_per_pair"), k and k.get("stage_ms"))
n=l.get("north_star"); print("north", n and {a:n[a] for a in n if not isinstance(n[a],dict)})
print("line bytes", len(json.dumps(l)))
PY
