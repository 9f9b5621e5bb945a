# round-end measurements: the default bench (30 frames + CPU baseline) and the rocprofv3 kernel statistics of the same command
# at 3 frames; outputs land in gpurun_out/ and are copied into profiles/ by hand
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 600 gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 3 --warmup 0 --no-cpu-baseline > /tmp/prof_bench.json 2>/tmp/prof.err
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$f" /root/repo/gpurun_out/kernel_stats_steps3.csv
cp /tmp/prof_bench.json /root/repo/gpurun_out/kernel_stats_steps3_bench.json
head -12 "$f" | cut -c1-160
