#!/bin/bash
# Round-3 GPU session N: the decode engine's row dot products on the matrix cores (-DUMGEN_ENG_MFMA=1, libumgen_hip_mfma.so) vs the VALU form
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3n_$name.json 2> gpurun_out/r3n_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3n_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3n_$name.err").read()[-1200:])
PY
}
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_mfma.so timeout 900 python -m pytest tests/test_gpu_decode_engine.py -x -q -s > gpurun_out/r3n_pytest_engine.log 2>&1; grep -v "^\[umgen\]\|amdgpu.ids" gpurun_out/r3n_pytest_engine.log | tail -12
run valu python bench.py --steps 2 --warmup 1 --no-cpu-baseline
run mfma UMGEN_LIB_PATH=umgen_amd/libumgen_hip_mfma.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
run mfma_fp16 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_mfma.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp16
run mfma_b8 UMGEN_LIB_PATH=umgen_amd/libumgen_hip_mfma.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_mfma.so UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3n_stamps_mfma.txt; grep "decode engine, group" gpurun_out/r3n_stamps_mfma.txt | tail -1
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_mfma.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -s -k "ensemble or 16bit or eight_scenes" > gpurun_out/r3n_pytest_parity.log 2>&1; grep -v "^\[umgen\]\|amdgpu.ids" gpurun_out/r3n_pytest_parity.log | tail -8 | cut -c1-600
