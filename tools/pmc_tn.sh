export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT" \
           "MfmaUtil VALUBusy LdsUtil OccupancyPercent"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pt$i -- python3 tools/dev_tn_one.py > /dev/null 2>&1
  echo "== $pmc"
  python3 tools/pmc_summary.py /tmp/pt$i 6 2>&1 | grep "gemm_tn\|gemm_nt" | cut -c1-900
done
