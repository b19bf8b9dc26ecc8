#!/bin/bash
# counters of the cbca kernels on the realistic pair at 1000x1500x256 (scripts/gpu_cbca_dense.py --mb-natural)
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/pmcl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_dense.py --mb-natural > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O | grep -i "^Kernel\|list_kernel\|strip_kernel<2, 4, 1, 2, true, true\|list_build" | cut -c1-1200
