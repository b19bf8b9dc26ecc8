ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/pmcl2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_READ_sum TCC_WRITE_sum TCC_REQ_sum TCC_STREAMING_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_dense.py --mb-natural > $O/p$i.log 2>&1
  echo "pass $i rc=$?"; tail -2 $O/p$i.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O | grep -i "^Kernel\|list_kernel" | cut -c1-900
