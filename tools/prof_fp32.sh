# Kernel-time breakdown of a precision mode's training step (run through gpurun).  usage: bash tools/prof_fp32.sh [fp32|bf16x3]
export TMPDIR=/tmp
rm -rf /tmp/pf32
rocprofv3 --kernel-trace --stats -d /tmp/pf32 -- python3 bench.py --precision ${1:-fp32} --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-config-legs 2>&1 | tail -1 | cut -c1-300
python3 tools/prof_summary.py $(ls /tmp/pf32/*/*_results.db | head -1) 25
