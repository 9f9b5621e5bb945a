cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptr
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptr -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/ptr.log 2>&1
f=$(find /tmp/ptr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name in ("sample_token_kernel", "gemv_ln_kernel<unsigned short, 1, 2, 2, false>", "fixed_token_kernel", "oar_engine_kernel<false>"):
    rs = sorted([r for r in rows if name in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    d = [dur(r) for r in rs]
    if not d: continue
    n = len(d)
    print(name, "n", n, "mean %.1f" % (sum(d) / n))
    if "sample_token" in name:
        per = n // 2   # two frames were run (timed + profiled)
        f = d[:2196]
        print("  map   mean %.1f" % (sum(f[:1024]) / 1024), " bbox mean %.1f" % (sum(f[1024:1684]) / 660), " image mean %.1f" % (sum(f[1684:2196]) / 512))
# gaps between consecutive kernels of one step: engine end -> head start, head end -> sampler start, sampler end -> next engine start
ks = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
gaps = {"eng->head": [], "head->samp": [], "samp->eng": []}
for a, b in zip(ks, ks[1:]):
    ga = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    an, bn = a["Kernel_Name"], b["Kernel_Name"]
    if "oar_engine_kernel" in an and "gemv_ln" in bn: gaps["eng->head"].append(ga / 1e3)
    if "gemv_ln" in an and "sample_token" in bn: gaps["head->samp"].append(ga / 1e3)
    if "sample_token" in an and "oar_engine_kernel" in bn: gaps["samp->eng"].append(ga / 1e3)
for k, v in gaps.items():
    if v: print("gap", k, "n", len(v), "mean %.2f us" % (sum(v) / len(v)))
PY
