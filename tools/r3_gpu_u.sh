#!/bin/bash
# Round-3 GPU session U: attention on the matrix cores (UMGEN_ENG_MFMA bit 16: q . K^T and P . V as MFMAs with hi / lo rows, dim-major V cache)
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3u_$name.json 2> gpurun_out/r3u_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3u_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; oar ms", round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3u_$name.err").read()[-1500:])
PY
}
for v in a28n2 a28n1; do
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so timeout 600 python -m pytest tests/test_gpu_decode_engine.py -x -q -s > gpurun_out/r3u_pytest_$v.log 2>&1; grep -v "^\[umgen\]\|amdgpu.ids" gpurun_out/r3u_pytest_$v.log | tail -4 | cut -c1-300
done
run default python bench.py --steps 2 --warmup 1 --no-cpu-baseline
for v in a28n2 a28n1; do
run $v UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
done
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_a28n2.so UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3u_stamps_a28n2.txt; grep "decode engine, group" gpurun_out/r3u_stamps_a28n2.txt | tail -1
run a28n2_b8 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_a28n2.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
