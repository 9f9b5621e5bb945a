#!/bin/bash
# Round-3 GPU session Q: matrix-core row products, third pass: c_fc + mlp partial sums (bits 12) with 3 / 4 K/V buffers, alternating runs for box noise
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3q_$name.json 2> gpurun_out/r3q_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3q_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; oar ms", round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3q_$name.err").read()[-1200:])
PY
}
for rep in 1 2; do
run valu_$rep python bench.py --steps 2 --warmup 1 --no-cpu-baseline
for v in m12nb3 m12 m13 m8nb3 m0nb3; do
  run ${v}_$rep UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
done
done
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12nb3.so UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3q_stamps_m12nb3.txt; grep "decode engine, group" gpurun_out/r3q_stamps_m12nb3.txt | tail -1
for b in 4 8; do
run m12nb3_b$b UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12nb3.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
done
run m12nb3_fp16 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_m12nb3.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp16
