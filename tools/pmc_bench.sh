# One training-step profile set of bench.py on the GPU box (run through gpurun): kernel-trace stats, MFMA / LDS utilisation and the two
# HBM-traffic passes (separate --pmc runs: FETCH_SIZE takes 3 of the 4 TCC slots).  usage: bash tools/pmc_bench.sh <tag>   e.g. r02_v3
export TMPDIR=/tmp
tag=${1:-r02}
mkdir -p gpurun_out profiles
B="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-config-legs"
rocprofv3 --kernel-trace --stats -d /tmp/pk_$tag -- $B > gpurun_out/${tag}_bench_under_trace.log 2>&1
python3 tools/prof_summary.py $(ls /tmp/pk_$tag/*/*_results.db | head -1) 45 > gpurun_out/${tag}_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pk2_$tag -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-parity --no-config-legs > /dev/null 2>&1
python3 tools/launch_count.py $(ls /tmp/pk_$tag/*/*_results.db | head -1) 6 $(ls /tmp/pk2_$tag/*/*_results.db | head -1) 4 > gpurun_out/${tag}_launch_count.txt 2>&1
B2="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-config-legs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$tag -- $B2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$tag -- $B2 > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil LdsUtil VALUBusy --kernel-trace --output-format csv -d /tmp/pu_$tag -- $B2 > /dev/null 2>&1
python3 tools/pmc_summary.py /tmp/pf_$tag 30 > gpurun_out/${tag}_pmc_fetch.txt 2>&1
python3 tools/pmc_summary.py /tmp/pw_$tag 30 > gpurun_out/${tag}_pmc_write.txt 2>&1
python3 tools/pmc_summary.py /tmp/pu_$tag 30 > gpurun_out/${tag}_pmc_util.txt 2>&1
python3 tools/pmc_traffic_json.py /tmp/pf_$tag /tmp/pw_$tag gpurun_out/${tag}_pmc_gemm_nt.json
