export TMPDIR=/tmp
rm -rf /tmp/pstep16
rocprofv3 --kernel-trace --stats -d /tmp/pstep16 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-config-legs --precision fp16 2>&1 | tail -1 | cut -c1-200
python3 tools/prof_summary.py $(ls /tmp/pstep16/*/*_results.db | head -1) 45
