#!/bin/bash
# Round-3 GPU session P: matrix-core row products per phase, second pass (sixth c_fc tile requested behind the attention: no spills with 4 K/V buffers)
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3p_$name.json 2> gpurun_out/r3p_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3p_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; oar ms", round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3p_$name.err").read()[-1200:])
PY
}
run valu python bench.py --steps 2 --warmup 1 --no-cpu-baseline
for v in m12 m14 m4 m15; do
  run $v UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
done
run valu_again python bench.py --steps 2 --warmup 1 --no-cpu-baseline
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12.so UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3p_stamps_m12.txt; grep "decode engine, group" gpurun_out/r3p_stamps_m12.txt | tail -1
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12.so timeout 600 python -m pytest tests/test_gpu_decode_engine.py -x -q -s > gpurun_out/r3p_pytest_m12.log 2>&1; grep -v "^\[umgen\]\|amdgpu.ids" gpurun_out/r3p_pytest_m12.log | tail -3
for b in 4 8 16; do
run valu_b$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
run m12_b$b UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
done
run m12_fp16 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp16
