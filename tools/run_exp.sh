export UMGEN_LIB_PATH=umgen_amd/libumgen_hip_ring.so
timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k linear 2>&1 | tail -3
timeout 120 python tools/gemm_bench.py | tail -6
