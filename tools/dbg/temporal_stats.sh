timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "temporal" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/prof_bench.json 2>/tmp/prof.err
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
grep -i "temporal\|attn_spatial\|layernorm" "$f" | cut -c1-200
