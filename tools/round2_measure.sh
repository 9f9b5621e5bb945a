#!/bin/bash
# Round-2 measurements (run on the GPU box through gpurun; outputs land in gpurun_out/ and are copied into profiles/ by hand):
#   1. the default bench (30 frames, CPU baseline, extra profiled frame)          -> bench_r02_full.json
#   2. rocprofv3 kernel statistics of the same command at 3 frames               -> r02_kernel_stats_steps3.csv
#   3. rocprofv3 --pmc FETCH_SIZE, bounded to 4 decode steps at KV length 2000     -> r02_pmc_fetch_size_engine.csv
#   4. the same rollout in the other configurations that are quoted in DESIGN.md: fp32 parity mode, 8 and 4 scenes per GPU,
#      the five-launch decode layer with the overlapped TAR pass (round-1 production path), the 2x-width stress configuration
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r02_full.json 2> gpurun_out/bench_r02_full.err
tail -c 400 gpurun_out/bench_r02_full.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 3 --warmup 0 --no-cpu-baseline > /tmp/prof_bench.json 2>/tmp/prof.err
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$f" /root/repo/gpurun_out/r02_kernel_stats_steps3.csv
cp /tmp/prof_bench.json /root/repo/gpurun_out/r02_kernel_stats_steps3_bench.json
head -8 "$f" | cut -c1-200
rm -rf /tmp/pmc
UMGEN_DEBUG_OAR_STEPS=2000:2004 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python /root/repo/tools/pmc_summary.py "$f" > /root/repo/gpurun_out/r02_pmc_fetch_size_engine.csv && head -6 /root/repo/gpurun_out/r02_pmc_fetch_size_engine.csv
cd /root/repo
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision fp32 > gpurun_out/bench_r02_fp32.json 2> gpurun_out/bench_r02_fp32.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 8 > gpurun_out/bench_r02_b8.json 2> gpurun_out/bench_r02_b8.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 4 > gpurun_out/bench_r02_b4.json 2> gpurun_out/bench_r02_b4.err
UMGEN_OVERLAP=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r02_launches_overlap.json 2> gpurun_out/bench_r02_launches_overlap.err
UMGEN_DECODE_ENGINE=0 UMGEN_OVERLAP=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r02_launches_plain.json 2> gpurun_out/bench_r02_launches_plain.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --config wide2x > gpurun_out/bench_r02_wide2x.json 2> gpurun_out/bench_r02_wide2x.err
for f in fp32 b8 b4 launches_overlap launches_plain wide2x; do python - <<PY
import json
d = json.load(open("gpurun_out/bench_r02_$f.json"))
print("$f", round(d["value"], 1), "scene-tokens/s", round(d["ms_per_step"], 1), "ms/frame; layer kernel(s)", round(d["roofline"]["avg_launch_us"], 1), "us, frac", round(d["roofline"]["frac"], 4), d["phases_ms_per_frame"])
PY
done
