#!/bin/bash
# Round-3 GPU session L: the map / box stacks on side streams beside the TAR stack (UMGEN_CONCURRENT_STACKS, default on) vs one behind the other
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" > gpurun_out/r3l_$name.json 2> gpurun_out/r3l_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3l_$name.json"))
    st = d["roofline"]["step"]
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us; ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1), "gemm", round(d["roofline_gemm"]["achieved"]), "attn", round(d["roofline_attn"]["achieved"]))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3l_$name.err").read()[-800:])
PY
}
for b in 1 4 8; do
  run conc_b$b python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b
  run seq_b$b UMGEN_CONCURRENT_STACKS=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b
done
run conc_fp16 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision fp16
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3l_pytest.log 2>&1; tail -4 gpurun_out/r3l_pytest.log
