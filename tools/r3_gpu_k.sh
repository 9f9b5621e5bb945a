#!/bin/bash
# Round-3 GPU session K: systolic layer switch with the parked mlp rows requested behind P1 / the key loop / P3's gather (lp0 = in front of everything, as before)
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3k_$name.json 2> gpurun_out/r3k_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3k_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3k_$name.err").read()[-800:])
PY
}
timeout 900 python -m pytest tests/test_gpu_decode_engine.py -x -q > gpurun_out/r3k_pytest_engine.log 2>&1; tail -3 gpurun_out/r3k_pytest_engine.log
for b in 5 6 8 10 16; do
  run b$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
  run lp0_b$b UMGEN_LIB_PATH=umgen_amd/libumgen_hip_lp0.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
done
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -k "eight_scenes or four_scenes" > gpurun_out/r3k_pytest_fullsize.log 2>&1; tail -3 gpurun_out/r3k_pytest_fullsize.log
