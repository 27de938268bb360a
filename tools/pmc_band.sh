# FETCH_SIZE / WRITE_SIZE of the wide-output NT GEMMs under the column-band tile orders (round 6).  usage (gpurun): bash tools/pmc_band.sh > gpurun_out/r06_pmc_band.txt
# plain 27090 x 768 -> 3072 GEMM with a bias (the fc1 shape; the GELU epilogue's two output tiles are left out so that the operand traffic shows), tile 320 and tile 160
export TMPDIR=/tmp
i=0
for tile in 320 160; do
for band in 0 1 2 3 4 6 12; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    TCOW_DEV_BAND=$band TILE=$tile MKN=27090,768,3072 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pb$i -- python3 tools/dev_gemm_one.py > /dev/null 2>&1
    echo "== tile=$tile band=$band :: $(python3 tools/pmc_summary.py /tmp/pb$i 6 2>&1 | grep gemm_nt | cut -c63-)"
  done
done
done
echo "(FETCH_SIZE in KiB per dispatch as rocprofv3 reports it: x 2.0 on gfx950 for wide coalesced reads.  algorithmic: A 41.6 MB + W 4.7 MB read, C 166 MB written)"
