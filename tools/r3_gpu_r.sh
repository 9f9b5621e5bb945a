#!/bin/bash
# Round-3 GPU session R: c_fc + mlp partial sums on the matrix cores with the c_fc fragments repacked (1 KB per request), 3 K/V buffers
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3r_$name.json 2> gpurun_out/r3r_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3r_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; oar ms", round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3r_$name.err").read()[-1200:])
PY
}
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12nb3.so timeout 600 python -m pytest tests/test_gpu_decode_engine.py -x -q -s > gpurun_out/r3r_pytest_m12nb3.log 2>&1; grep -v "^\[umgen\]\|amdgpu.ids" gpurun_out/r3r_pytest_m12nb3.log | tail -3
for b in 1 2 4 5 8 16; do
run valu_b$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
run m12nb3_b$b UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12nb3.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
done
run m12nb3_fp16 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12nb3.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp16
