#!/bin/bash
# Round-3 GPU session E: engine micro-optimisations (4 lanes per key, transposed row sums, two polls in flight) A/B + closed-loop probe
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3e_$name.json 2> gpurun_out/r3e_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3e_$name.json"))
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "oar ms", round(d["phases_ms_per_frame"]["oar"],1), "eng", d["decode_engine"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3e_$name.err").read()[-800:])
PY
}
timeout 900 python -m pytest tests/test_gpu_decode_engine.py -x -q > gpurun_out/r3e_pytest_engine.log 2>&1; tail -5 gpurun_out/r3e_pytest_engine.log
for v in "" lpk8 nopoll2 notr nb3 stag3 stag12; do
  if [ -z "$v" ]; then run new python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  else run $v UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline; fi
done
run new_again python bench.py --steps 2 --warmup 1 --no-cpu-baseline
run new_b8 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run lpk8_b8 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_lpk8.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3e_stamps_new.txt; tail -25 gpurun_out/r3e_stamps_new.txt
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_lpk8.so UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3e_stamps_lpk8.txt; tail -25 gpurun_out/r3e_stamps_lpk8.txt
timeout 1500 python tools/dbg/closed_loop_probe.py fp32 fp16 bf16 > gpurun_out/r3e_probe.txt 2>&1; tail -30 gpurun_out/r3e_probe.txt
