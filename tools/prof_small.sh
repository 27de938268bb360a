# Which kernels surround the tiny fill / copy launches of the bf16 step (run through gpurun).
export TMPDIR=/tmp
rm -rf /tmp/psm
rocprofv3 --kernel-trace --stats -d /tmp/psm -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1
db=$(ls /tmp/psm/*/*_results.db | head -1)
python3 tools/prof_neighbors.py $db FillFunctor | tail -16
python3 tools/prof_neighbors.py $db copyBuffer | tail -14
