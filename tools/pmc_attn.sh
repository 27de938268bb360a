# per-kernel PMC of the attention kernels (separate passes: SQ has 8 slots, TCC 4); usage: bash tools/pmc_attn.sh <outfile>
# spatial (streaming kernels) and temporal (wave-private kernels) at the benchmark shape, tools/dev_attn_one.py as the traced program
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_attn.txt}
: > $out
i=0
for temporal in 0 1; do
  for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "MfmaUtil VALUBusy LdsUtil OccupancyPercent"; do
    i=$((i+1))
    export TEMPORAL=$temporal
    rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pa$i -- python3 tools/dev_attn_one.py > /dev/null 2>&1
    echo "== TEMPORAL=$temporal :: $pmc" >> $out
    python3 tools/pmc_summary.py /tmp/pa$i 8 2>&1 | grep attn >> $out
  done
done
