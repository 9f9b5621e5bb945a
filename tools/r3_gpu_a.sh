#!/bin/bash
# Round-3 GPU session A: whole -m gpu suite, quick bench lines (bf16 / fp16 at one scene, 8 scenes with and without the systolic
# schedule in its three residency variants), CU-mask probe outputs.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3a_pytest.log 2>&1
tail -25 gpurun_out/r3a_pytest.log
run() { name=$1; shift; env "$@" > gpurun_out/r3a_$name.json 2> gpurun_out/r3a_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3a_$name.json"))
    print("$name", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 1), "ms/frame; engine", round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 4), "gemm", round(d["roofline_gemm"]["achieved"]), "attn", round(d["roofline_attn"]["achieved"]), d["phases_ms_per_frame"]["ego"], d["phases_ms_per_frame"]["tar"], d["phases_ms_per_frame"]["oar"], "eng", d["decode_engine"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r3a_$name.err").read()[-800:])
PY
}
run bf16_b1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run fp16_b1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision fp16
run bf16_b8 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run bf16_b8_sys UMGEN_ENGINE_SYSTOLIC=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run bf16_b8_sys_keepwo UMGEN_ENGINE_SYSTOLIC=1 UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_keepwo.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run bf16_b8_sys_keepboth UMGEN_ENGINE_SYSTOLIC=1 UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_keepboth.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8
run bf16_b5_sys UMGEN_ENGINE_SYSTOLIC=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 5
run bf16_b5 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 5
{ echo "# tools/micro/cumask_probe on MI355X: which XCDs do the workgroups of a 1024-block launch on a CU-masked stream land on?";
  timeout 30 tools/micro/cumask_probe contig 64; timeout 30 tools/micro/cumask_probe contig 128; timeout 30 tools/micro/cumask_probe contig 192;
  timeout 30 tools/micro/cumask_probe 0xFF; timeout 30 tools/micro/cumask_probe 0xF0; echo "exit code of the 0xF0 run: $?"; } > gpurun_out/r3_cumask_probe.txt 2>&1
cat gpurun_out/r3_cumask_probe.txt
