#!/bin/bash
# Round-3 GPU session H: systolic schedule, resident rows + shared tail layers (4.5 B items per group); full -m gpu suite
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3h_$name.json 2> gpurun_out/r3h_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3h_$name.json"))
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1), "eng", d["decode_engine"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3h_$name.err").read()[-800:])
PY
}
timeout 900 python -m pytest tests/test_gpu_decode_engine.py -x -q > gpurun_out/r3h_pytest_engine.log 2>&1; tail -3 gpurun_out/r3h_pytest_engine.log
run b1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline
for b in 2 3 4; do
  run b${b} python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
  run b${b}_sys UMGEN_ENGINE_SYSTOLIC=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
done
for b in 5 6 8 12 16 32; do
  run b$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
done
run g1_b8 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_g1.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run b8_fp16 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8 --precision fp16
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r3h_pytest_gpu.log 2>&1; tail -8 gpurun_out/r3h_pytest_gpu.log
