ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/pmcj; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o p1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/p2 -o p2 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops >> $O/log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O | grep -i "^Kernel\|join_owner" | cut -c1-900
