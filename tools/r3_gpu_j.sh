#!/bin/bash
# Round-3 GPU session J: systolic schedule with the K/V rows of the item touched into the XCD's L2 at item start; per-item stamps at 8 scenes
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3j_$name.json 2> gpurun_out/r3j_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3j_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3j_$name.err").read()[-800:])
PY
}
for b in 8 16; do
  run b$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
  run systouch_b$b UMGEN_LIB_PATH=umgen_amd/libumgen_hip_systouch.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
done
UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 8 > /dev/null 2> gpurun_out/r3j_stamps_b8.txt; grep "decode engine" gpurun_out/r3j_stamps_b8.txt | tail -3
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_systouch.so UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 8 > /dev/null 2> gpurun_out/r3j_stamps_b8_touch.txt; grep "decode engine" gpurun_out/r3j_stamps_b8_touch.txt | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fp16_mode_refuses" > gpurun_out/r3j_pytest.log 2>&1; tail -3 gpurun_out/r3j_pytest.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp16 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('fp16 weight load', d['weight_load_s'], 's;', d['value'], 'tok/s')"
