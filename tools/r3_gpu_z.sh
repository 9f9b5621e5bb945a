#!/bin/bash
# Round-3 GPU session Z: x-edge chunks per consumer group, placed by a calibration ping-pong (UMGEN_ENGINE_GX_CALIB=0: first candidate everywhere)
mkdir -p gpurun_out
one() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value'],1), 'tok/s; engine', round(d['roofline']['avg_launch_us'],1), 'us')"; }
timeout 900 python -m pytest tests/test_gpu_decode_engine.py -x -q 2>&1 | tail -2
UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep "x-edge"
for r in 1 2; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | one calib
UMGEN_ENGINE_GX_CALIB=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | one nocalib
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_prev.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | one prev
done
for b in 4 8 16; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | one calib_b$b
UMGEN_LIB_PATH=umgen_amd/libumgen_hip_prev.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | one prev_b$b
done
