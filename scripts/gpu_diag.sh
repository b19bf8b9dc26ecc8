#!/bin/bash
ulimit -c 0
step() { echo "== $*"; env "$@" 2>&1 | tail -6; }
step timeout 300 python -m pytest tests/test_ref_parity.py -m gpu -q -x -k cbca
step timeout 300 python -m pytest tests/test_ref_parity.py -m gpu -q -x -k predict
step timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not cbca"
step timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_abi.py tests/test_gpu_fc.py -m gpu -q -x
