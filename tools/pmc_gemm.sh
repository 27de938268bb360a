export TMPDIR=/tmp
i=0
for cfg in "256 8192,8192,8192" "128 8192,8192,8192" "320 27090,768,3072" "160 27090,3072,768"; do      # forced tile (tcow_gemm_args.tile), M,K,N
  set -- $cfg
  for pmc in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL" "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "MfmaUtil LdsUtil" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    TILE=$1 MKN=$2 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pg$i -- python3 tools/dev_gemm_one.py > /dev/null 2>&1
    echo "== tile=$1 MKN=$2 :: $pmc"
    python tools/pmc_summary.py /tmp/pg$i 6 2>&1 | grep gemm_nt
  done
done
