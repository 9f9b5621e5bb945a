#!/bin/bash
# Round-3 GPU session C: staggered 256-tile GEMM (tests + bench), decode engine with the q-first hand-off (tests, A/B, stamps)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "linear" > gpurun_out/r3c_pytest_linear.log 2>&1; tail -3 gpurun_out/r3c_pytest_linear.log
python tools/gemm_bench.py 44140 88280 176560 353120 > gpurun_out/r3c_gemm_bench.txt 2>&1; cat gpurun_out/r3c_gemm_bench.txt
UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_nostagger.so python tools/gemm_bench.py 353120 > gpurun_out/r3c_gemm_bench_nostagger.txt 2>&1; cat gpurun_out/r3c_gemm_bench_nostagger.txt
timeout 1500 python -m pytest tests/test_gpu_decode_engine.py tests/test_gpu_fullsize.py -m gpu -q -x --deselect tests/test_gpu_fullsize.py::test_16bit_teacher_forced_frame_at_production_width_lies_inside_the_oracle_ensemble > gpurun_out/r3c_pytest_engine.log 2>&1; tail -5 gpurun_out/r3c_pytest_engine.log
run() { name=$1; shift; env "$@" > gpurun_out/r3c_$name.json 2> gpurun_out/r3c_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3c_$name.json"))
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "gemm", round(d["roofline_gemm"]["achieved"]), "attn", round(d["roofline_attn"]["achieved"]), "ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1), "eng", d["decode_engine"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3c_$name.err").read()[-800:])
PY
}
run bf16_b1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run bf16_b1_qlast UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_qlast.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run bf16_b1_stamps UMGEN_DEBUG_TIMING=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline
grep "decode engine, group 0" gpurun_out/r3c_bf16_b1_stamps.err | tail -2
run bf16_b1_stamps_qlast UMGEN_DEBUG_TIMING=1 UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_qlast.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
grep "decode engine, group 0" gpurun_out/r3c_bf16_b1_stamps_qlast.err | tail -2
run bf16_b4 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 4
run bf16_b8 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
