#!/bin/bash
# Round-3 GPU session I: sampler (threshold selection of the k-th largest logit) + systolic q|k|v rows requested one item ahead
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3i_$name.json 2> gpurun_out/r3i_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3i_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "step", round(st["avg_step_us"], 1), "us; ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3i_$name.err").read()[-800:])
PY
}
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k sampler > gpurun_out/r3i_pytest_sampler.log 2>&1; tail -5 gpurun_out/r3i_pytest_sampler.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decode_engine.py -x -q > gpurun_out/r3i_pytest_parity.log 2>&1; tail -5 gpurun_out/r3i_pytest_parity.log
run b1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline
run b1_oldsampler UMGEN_LIB_PATH=umgen_amd/libumgen_hip_g1.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline
for b in 8 16; do
  run b$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b
  for v in wqa0 wqa_nb2 wqa_nb1; do run ${v}_b$b UMGEN_LIB_PATH=umgen_amd/libumgen_hip_$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b; done
done
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_wqa0.so timeout 600 python -m pytest tests/test_gpu_decode_engine.py -x -q -k batch > gpurun_out/r3i_pytest_engine_wqa0.log 2>&1; tail -2 gpurun_out/r3i_pytest_engine_wqa0.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_i -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1; f=$(find /tmp/prof_i -name "*kernel_stats.csv" | head -1); grep -i "sample\|gemv" "$f" | cut -c1-200
