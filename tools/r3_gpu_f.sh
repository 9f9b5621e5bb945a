#!/bin/bash
# Round-3 GPU session F: engine (mlp partial sums remapped: 4 lanes share 4 rows; more K/V buffers) A/B + rollout-vs-trace probe
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3f_$name.json 2> gpurun_out/r3f_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3f_$name.json"))
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "oar ms", round(d["phases_ms_per_frame"]["oar"],1), "eng", d["decode_engine"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3f_$name.err").read()[-800:])
PY
}
timeout 900 python -m pytest tests/test_gpu_decode_engine.py -x -q > gpurun_out/r3f_pytest_engine.log 2>&1; tail -5 gpurun_out/r3f_pytest_engine.log
for v in "" e1 nb3 nb4 nb5 poll2; do
  if [ -z "$v" ]; then run new python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  else run $v UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline; fi
done
run new_b8 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run nb4_b8 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_nb4.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run nb4_fp16 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_nb4.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp16
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_nb4.so UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3f_stamps_nb4.txt; grep "decode engine" gpurun_out/r3f_stamps_nb4.txt | tail -3
timeout 1500 python tools/dbg/closed_loop_probe2.py bf16 fp32 > gpurun_out/r3f_probe2.txt 2>&1; grep -v "^\[umgen\]\|amdgpu.ids" gpurun_out/r3f_probe2.txt | tail -30
