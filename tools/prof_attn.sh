# Per-kernel durations of the spatial attention kernels at the benchmark shape (run through gpurun).  usage: bash tools/prof_attn.sh
export TMPDIR=/tmp
rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats -d /tmp/pa -- python3 tools/dev_attn_time.py 2>&1 | grep -E "spatial|rror" | head -3
python3 tools/prof_summary.py $(ls /tmp/pa/*/*_results.db | head -1) 8 | grep -E "attn|Name"
