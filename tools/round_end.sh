# Evidence set of a round on the GPU box (run through gpurun): bench line with the precision legs, step profiles + PMC passes of the bf16
# step, kernel breakdown of the two f32-storage modes, configs[3] / [4].   usage: bash tools/round_end.sh <tag>   e.g. r02_v3
export TMPDIR=/tmp
tag=${1:-r02}
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 5 2> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench.json
bash tools/pmc_bench.sh $tag
for m in bf16x3 fp32; do bash tools/prof_fp32.sh $m > gpurun_out/${tag}_step_${m}_kernel_stats.txt 2>&1; done
python3 tools/run_configs.py > gpurun_out/${tag}_configs_3_4.jsonl 2> gpurun_out/${tag}_configs.err
tail -c 600 gpurun_out/${tag}_bench.json
