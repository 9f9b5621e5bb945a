#!/bin/bash
# Round-3 GPU session D: VQ decoders (f-4), fast-GELU GEMM epilogues, engine traces for the parity analysis, bench lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vq.py -q -k "small or wrappers" > gpurun_out/r3d_pytest_vq.log 2>&1; tail -15 gpurun_out/r3d_pytest_vq.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "linear" > gpurun_out/r3d_pytest_linear.log 2>&1; tail -3 gpurun_out/r3d_pytest_linear.log
python tools/gemm_bench.py 44140 353120 > gpurun_out/r3d_gemm_bench.txt 2>&1; cat gpurun_out/r3d_gemm_bench.txt
python tools/dbg/dump_trace.py tiny full_width > gpurun_out/r3d_dump.log 2>&1; tail -3 gpurun_out/r3d_dump.log
run() { name=$1; shift; env "$@" > gpurun_out/r3d_$name.json 2> gpurun_out/r3d_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3d_$name.json"))
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "gemm", round(d["roofline_gemm"]["achieved"]), "attn", round(d["roofline_attn"]["achieved"]), "ego/tar/oar ms", round(d["phases_ms_per_frame"]["ego"],1), round(d["phases_ms_per_frame"]["tar"],1), round(d["phases_ms_per_frame"]["oar"],1), "eng", d["decode_engine"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3d_$name.err").read()[-800:])
PY
}
run bf16_b1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run bf16_b1_no256 UMGEN_GEMM256_MIN_TILES=100000000 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run fp16_b1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision fp16
run bf16_b8 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
