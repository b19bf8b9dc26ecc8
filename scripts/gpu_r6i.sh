#!/bin/bash
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r6i; mkdir -p $O
for c in kitti_fast kitti_slow; do
timeout 600 python bench.py --pairs-in-flight 3 --config $c > $O/pipe_$c.json 2> $O/pipe_$c.err; echo "rc=$?"
python - <<PY
import json
l=json.loads(open("$O/pipe_$c.json").read().strip().splitlines()[-1])["pipelined"]
for k,v in l["pairs_in_flight"].items(): print("$c", k, v["ms_per_pair"], v["ms_per_pair_min"], v["bit_exact_vs_alone"])
PY
done
