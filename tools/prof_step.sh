# Kernel-time breakdown of the bf16 training step (run through gpurun): top kernels of 4 timed steps.
export TMPDIR=/tmp
rm -rf /tmp/pstep
rocprofv3 --kernel-trace --stats -d /tmp/pstep -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | cut -c1-200
python3 tools/prof_summary.py $(ls /tmp/pstep/*/*_results.db | head -1) ${1:-16}
