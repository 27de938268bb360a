# round-3 GPU dev loop: microbenchmarks, attention correctness + A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 build/ubench_valu > gpurun_out/ubench_valu.txt 2>&1; echo "ubench rc=$?"
grep "^D  attn" gpurun_out/ubench_valu.txt
for r in 0 1 2; do TCOW_ATTN_RES=$r timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -2; done
for r in 0 1 2; do TCOW_ATTN_RES=$r timeout 300 python tools/dev_attn_time.py 2>&1 | tail -1; done | tee gpurun_out/attn_ab.txt
