"""Per-frame statistics of the decode step's kernels out of gpurun_out/ktrace_small.csv (tools/dbg/trace_gaps.sh: rocprofv3 --kernel-trace of
bench.py --steps 3 --warmup 0): how long is the engine launch in frames whose launches carry background workers, and until which step?"""
import csv
import statistics as st
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ktrace_small.csv"
rows = [(r["name"], int(r["start"]), int(r["end"])) for r in csv.DictReader(open(path))]
eng = [i for i, r in enumerate(rows) if r[0] == "engine"]
frames, cur = [], []
for i in eng:
    if cur and rows[i][1] - rows[cur[-1]][2] > 5_000_000:
        frames.append(cur)
        cur = []
    cur.append(i)
frames.append(cur)
print("# frames 0, 1: their launches carry the background pass of the next frame; frame 2 (last of the rollout) and frame 3 (bench.py's profiled frame): no pass")
for fi, f in enumerate(frames):
    durs = [rows[i][2] - rows[i][1] for i in f if rows[i][2] - rows[i][1] > 100_000]
    steps = [rows[b][1] - rows[a][1] for a, b in zip(f[:-1], f[1:])]
    oth = {}
    for i in range(f[0], f[-1]):
        if rows[i][0] in ("gemv", "sampler"):
            oth.setdefault(rows[i][0], []).append(rows[i][2] - rows[i][1])
    print(f"frame {fi}: {len(durs)} engine launches, average {st.mean(durs) / 1e3:.1f} us (max {max(durs) / 1e3:.1f}); step (engine start to engine start) {st.mean(steps) / 1e3:.1f} us; "
          + ", ".join(f"{k} {st.mean(v) / 1e3:.2f} us" for k, v in oth.items()))
d1 = [rows[i][2] - rows[i][1] for i in frames[1] if rows[i][2] - rows[i][1] > 100_000]
d2 = [rows[i][2] - rows[i][1] for i in frames[2]]
n = min(len(d1), len(d2))
diff = [(d1[k] - d2[k]) / 1e3 for k in range(n)]
print("# engine launch of frame 1 (with workers) minus frame 2 (without) at the same decode step, us: mean / median / p90 / max per 200 steps")
for a in range(0, n, 200):
    seg = diff[a:a + 200]
    print(f"steps {a:4d}..{a + len(seg) - 1:4d}: {st.mean(seg):6.1f} {st.median(seg):6.1f} {sorted(seg)[int(len(seg) * 0.9)]:6.1f} {max(seg):6.1f}")
