export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or layernorm or sgemm" 2>&1 | tail -8
timeout 2400 python -m pytest tests/test_gpu_seeker.py -x -q -m gpu 2>&1 | tail -12
