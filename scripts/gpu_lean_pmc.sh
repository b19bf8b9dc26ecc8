#!/bin/bash
# usage (ON the GPU box): bash scripts/gpu_lean_pmc.sh <tag> rb:variant ...  -- HBM bytes and time of the lean pass per launch variant
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for SET in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/$SET -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_lean_pmc.py "$@" > $O/$SET.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/time -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_lean_pmc.py "$@" > $O/time.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
def rows(d, name):
    f = glob.glob("$O/%s/**/p_%s.csv" % (d, name), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in rows(cnt, "counter_collection"):
        if "lean_kernel" in r["Kernel_Name"] and r["Counter_Name"] == cnt:
            acc[(r["Kernel_Name"][:40], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(cnt, k, "n=%d" % len(v), "mean GB = %.3f" % (sum(v) / len(v) * (2 if cnt == "FETCH_SIZE" else 1) * 1024 / 1e9))
acc = collections.defaultdict(list)
for r in rows("time", "kernel_trace"):
    if "lean_kernel" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:40], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    print("time", k, "n=%d" % len(v), "mean us = %.1f  min %.1f" % (sum(v) / len(v), min(v)))
PY
