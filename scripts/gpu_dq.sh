#!/bin/bash
# EXPERIMENT: parity + kernel-trace timing of the deferred-queue cbca kernel
ulimit -c 0
mkdir -p gpurun_out
timeout 300 python scripts/gpu_dq.py > gpurun_out/dq_parity.log 2>&1; echo "parity rc $?"
tail -5 gpurun_out/dq_parity.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dq_prof -o dq -- python $R/scripts/gpu_dq.py --time > $R/gpurun_out/dq_time.log 2>&1; echo "time rc $?"
cd $R; grep -E "^(mb|kitti)" gpurun_out/dq_time.log
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/dq_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cbca" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
