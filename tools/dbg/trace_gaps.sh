#!/bin/bash
# kernel trace of a short one-scene bench -> gpurun_out/ktrace.csv (analysed by tools/dbg/trace_gaps.py)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python /root/repo/bench.py --steps 3 --warmup 0 --no-cpu-baseline > /tmp/kt_bench.json 2>/tmp/kt.err
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open("/root/repo/gpurun_out/ktrace_small.csv", "w")
out.write("name,start,end\n")
for r in rows:
    n = r["Kernel_Name"]
    short = "engine" if "oar_engine_kernel" in n else ("sampler" if "sample_token" in n else ("gemv" if "gemv_ln" in n else ("fixed" if "fixed_token" in n else n[:40].replace(",", ";"))))
    out.write(f"{short},{r['Start_Timestamp']},{r['End_Timestamp']}\n")
PY
ls -la /root/repo/gpurun_out/ktrace_small.csv
