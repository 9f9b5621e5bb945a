#!/bin/bash
# Round-3 GPU session O: matrix-core row products per phase (UMGEN_ENG_MFMA bits: 1 q|k|v, 2 c_proj, 4 c_fc, 8 mlp partial sums)
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3o_$name.json 2> gpurun_out/r3o_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3o_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; oar ms", round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3o_$name.err").read()[-1200:])
PY
}
run valu python bench.py --steps 2 --warmup 1 --no-cpu-baseline
for v in m15 m13 m12nb3 m8; do
  run $v UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
done
for v in m15 m8; do
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3o_stamps_$v.txt; grep "decode engine, group" gpurun_out/r3o_stamps_$v.txt | tail -1
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so timeout 600 python -m pytest tests/test_gpu_decode_engine.py -x -q -k "teacher or batch_invariant" > gpurun_out/r3o_pytest_$v.log 2>&1; tail -2 gpurun_out/r3o_pytest_$v.log
done
run m8_b8 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m8.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run m8_fp16 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m8.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp16
