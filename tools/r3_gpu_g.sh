#!/bin/bash
# Round-3 GPU session G: systolic schedule with RESIDENT c_proj / c_fc rows (fits the register file since the mlp remap), fp16 fma_mix,
# ensemble parity tests with the new engine arithmetic
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3g_$name.json 2> gpurun_out/r3g_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3g_$name.json"))
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "oar ms", round(d["phases_ms_per_frame"]["oar"],1), "eng", d["decode_engine"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3g_$name.err").read()[-800:])
PY
}
timeout 900 python -m pytest tests/test_gpu_decode_engine.py -x -q > gpurun_out/r3g_pytest_engine.log 2>&1; tail -3 gpurun_out/r3g_pytest_engine.log
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_syskeep.so timeout 900 python -m pytest tests/test_gpu_decode_engine.py -x -q > gpurun_out/r3g_pytest_engine_syskeep.log 2>&1; tail -3 gpurun_out/r3g_pytest_engine_syskeep.log
run new python bench.py --steps 2 --warmup 1 --no-cpu-baseline
run new_fp16 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp16
for b in 5 8 16; do
  run new_b$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
  run syskeep_b$b UMGEN_LIB_PATH=umgen_amd/libumgen_hip_syskeep.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
done
run syskeepwo_b8 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_syskeepwo.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run syskeep_b4sys UMGEN_ENGINE_SYSTOLIC=1 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_syskeep.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 4
run syskeep_b32 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_syskeep.so python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 32
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "ensemble or 16bit" > gpurun_out/r3g_pytest_parity.log 2>&1; tail -5 gpurun_out/r3g_pytest_parity.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -k "ensemble" -s > gpurun_out/r3g_pytest_fullsize.log 2>&1; grep -v "^\[umgen\]\|amdgpu.ids" gpurun_out/r3g_pytest_fullsize.log | tail -8
