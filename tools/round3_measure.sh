#!/bin/bash
# Round-3 measurements (run on the GPU box through gpurun; outputs land in gpurun_out/ and are copied into profiles/ by hand):
#   1. the default bench (30 frames, CPU baseline, extra profiled frame)                    -> r03_bench_full.json
#   2. rocprofv3 kernel statistics of the same command at 3 frames                           -> r03_rocprofv3_kernel_stats_bench_steps3.csv
#   3. rocprofv3 --pmc FETCH_SIZE of the decode engine at the MEAN KV length (steps 1101..1104) -> r03_pmc_fetch_size_engine.csv
#   4. SQ / TA / TCP counters of the round-3 GEMM (256-tile, fc shape, 8 scenes' rows) and of the spatial attention kernel
#   5. the other configurations quoted in DESIGN.md: fp16, fp32, 4 / 8 scenes per GPU, five-launch decode layer, 2x width
#   6. closed-loop agreement of the 16-bit modes with fp32 mode over a 30-frame greedy rollout
mkdir -p gpurun_out
python bench.py > gpurun_out/r03_bench_full.json 2> gpurun_out/r03_bench_full.err
tail -c 300 gpurun_out/r03_bench_full.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 3 --warmup 0 --no-cpu-baseline > /tmp/prof_bench.json 2>/tmp/prof.err
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$f" /root/repo/gpurun_out/r03_rocprofv3_kernel_stats_bench_steps3.csv
cp /tmp/prof_bench.json /root/repo/gpurun_out/r03_rocprofv3_kernel_stats_bench_steps3_bench.json
head -10 "$f" | cut -c1-180
rm -rf /tmp/pmc
UMGEN_DEBUG_OAR_STEPS=1101:1105 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python /root/repo/tools/pmc_summary.py "$f" > /root/repo/gpurun_out/r03_pmc_fetch_size_engine.csv && head -4 /root/repo/gpurun_out/r03_pmc_fetch_size_engine.csv
cd /root/repo
bash tools/gemm_pmc.sh > gpurun_out/r03_pmc_gemm_fc_353120x3072x768.txt 2>&1; tail -12 gpurun_out/r03_pmc_gemm_fc_353120x3072x768.txt | cut -c1-300
bash tools/attn_pmc.sh > gpurun_out/r03_pmc_attn_spatial_F20_S2207_H16.txt 2>&1; tail -8 gpurun_out/r03_pmc_attn_spatial_F20_S2207_H16.txt | cut -c1-300
cd /root/repo
python bench.py --steps 10 --warmup 1 --no-cpu-baseline --precision fp16 > gpurun_out/r03_bench_fp16.json 2> gpurun_out/r03_bench_fp16.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp32 > gpurun_out/r03_bench_fp32.json 2> gpurun_out/r03_bench_fp32.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 8 > gpurun_out/r03_bench_b8.json 2> gpurun_out/r03_bench_b8.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 4 > gpurun_out/r03_bench_b4.json 2> gpurun_out/r03_bench_b4.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 4 --task control > gpurun_out/r03_bench_control_b4.json 2> gpurun_out/r03_bench_control_b4.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 6 > gpurun_out/r03_bench_b6.json 2> gpurun_out/r03_bench_b6.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 16 > gpurun_out/r03_bench_b16.json 2> gpurun_out/r03_bench_b16.err
UMGEN_DECODE_ENGINE=0 UMGEN_OVERLAP=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_launches_plain.json 2> gpurun_out/r03_bench_launches_plain.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --config wide2x > gpurun_out/r03_bench_wide2x.json 2> gpurun_out/r03_bench_wide2x.err
for f in full fp16 fp32 b4 b6 b8 b16 launches_plain wide2x; do python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r03_bench_$f.json"))
    print("$f", round(d["value"], 1), "scene-tokens/s", round(d["ms_per_step"], 1), "ms/frame; layer kernel(s)", round(d["roofline"]["avg_launch_us"], 1), "us, frac", round(d["roofline"]["frac"], 4), "gemm", round(d["roofline_gemm"]["achieved"]), "attn", round(d["roofline_attn"]["achieved"]), d["phases_ms_per_frame"])
except Exception as e:
    print("$f FAILED", e)
PY
done
python tools/closed_loop.py --frames 30 > gpurun_out/r03_closed_loop.log 2>&1; tail -4 gpurun_out/r03_closed_loop.log | cut -c1-600
