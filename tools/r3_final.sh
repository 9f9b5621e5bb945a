#!/bin/bash
# Round-3 final pass: every measurement of tools/round3_measure.sh with the shipped library, the engine's per-item stamps, the two large
# batches, smoke() and the whole -m gpu suite
bash tools/round3_measure.sh
UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r03_engine_stamps.txt; grep "decode engine" gpurun_out/r03_engine_stamps.txt | tail -3
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 12 > gpurun_out/r03_bench_b12.json 2>/dev/null
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 32 > gpurun_out/r03_bench_b32.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_pytest_gpu.log 2>&1; tail -3 gpurun_out/r03_pytest_gpu.log
