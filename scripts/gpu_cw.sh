#!/bin/bash
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/cw; mkdir -p $O
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o cw -- python $R/scripts/gpu_cbca_dense.py 2>&1 | grep -v "amdgpu.ids\|rocprofv3\|^W2\|^E2" | tee $O/dense.txt
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/cw/prof/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "cbca" in r["Name"]: print(r["Name"][:70], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
PY
timeout 600 python -m pytest tests/test_gpu_instantiations.py -x -q -k "window or forms or listed" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
