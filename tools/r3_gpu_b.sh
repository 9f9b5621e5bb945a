#!/bin/bash
# Round-3 GPU session B: the 256 x 256 GEMM kernel (tests, isolated bench, A/B in the whole frame), f-3 tail A/B, parity diagnostics.
mkdir -p gpurun_out
python tools/dbg/mfma_error.py > gpurun_out/r3b_mfma_error.txt 2>&1; cat gpurun_out/r3b_mfma_error.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "linear" > gpurun_out/r3b_pytest_linear.log 2>&1; tail -5 gpurun_out/r3b_pytest_linear.log
python tools/gemm_bench.py 44140 353120 > gpurun_out/r3b_gemm_bench.txt 2>&1; cat gpurun_out/r3b_gemm_bench.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_decode_engine.py tests/test_shard_gloo.py -m gpu -q -rA > gpurun_out/r3b_pytest_rest.log 2>&1
grep -E "passed|failed|FAILED|ERROR|max dev|first divergence|arg-max flips" gpurun_out/r3b_pytest_rest.log | cut -c1-700 | tail -40
run() { name=$1; shift; env "$@" > gpurun_out/r3b_$name.json 2> gpurun_out/r3b_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3b_$name.json"))
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "gemm", round(d["roofline_gemm"]["achieved"]), "attn", round(d["roofline_attn"]["achieved"]), "ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1), "eng", d["decode_engine"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3b_$name.err").read()[-800:])
PY
}
run bf16_b1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run bf16_b1_no256 UMGEN_GEMM256_MIN_TILES=100000000 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run bf16_b1_notail UMGEN_NO_TAIL=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run bf16_b1_nostage UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_nostage.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run bf16_b1_stamps UMGEN_DEBUG_TIMING=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline
grep "decode engine, group 0" gpurun_out/r3b_bf16_b1_stamps.err | tail -2
run bf16_b1_stamps_nostage UMGEN_DEBUG_TIMING=1 UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_nostage.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
grep "decode engine, group 0" gpurun_out/r3b_bf16_b1_stamps_nostage.err | tail -2
run bf16_b8 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run bf16_b6 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 6
run bf16_b16 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 16
