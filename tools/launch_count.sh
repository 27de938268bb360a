# Steady-state kernel launches per training step (run through gpurun): two kernel traces that differ in --steps, see tools/launch_count.py.
# usage: bash tools/launch_count.sh <out file>
export TMPDIR=/tmp
rm -rf /tmp/lc_a /tmp/lc_b
rocprofv3 --kernel-trace --stats -d /tmp/lc_a -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/lc_b -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity > /dev/null 2>&1
python3 tools/launch_count.py $(ls /tmp/lc_a/*/*_results.db | head -1) 6 $(ls /tmp/lc_b/*/*_results.db | head -1) 4 > ${1:-gpurun_out/launch_count.txt} 2>&1
