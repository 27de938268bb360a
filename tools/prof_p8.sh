export TMPDIR=/tmp
export TCOW_GEMM_P8=2
rm -rf /tmp/pp8
rocprofv3 --kernel-trace --stats -d /tmp/pp8 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | cut -c1-120
python3 tools/prof_summary.py $(ls /tmp/pp8/*/*_results.db | head -1) 8
