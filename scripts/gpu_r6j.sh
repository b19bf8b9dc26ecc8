#!/bin/bash
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r6j; mkdir -p $O
MC_REQUIRE_REF=1 timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_ref_parity.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, mc_cnn_amd as mc
from mc_cnn_amd import adcensus
x = torch.randn((2, 64, 370, 1226), device="cuda"); n = torch.empty((2, 1, 370, 1226), device="cuda"); o = torch.empty_like(x)
for _ in range(3): adcensus.Normalize_forward(x, n, o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): adcensus.Normalize_forward(x, n, o)
e1.record(); torch.cuda.synchronize()
print("normalize 2x64x370x1226: %.4f ms" % (e0.elapsed_time(e1) / 20))
PY
