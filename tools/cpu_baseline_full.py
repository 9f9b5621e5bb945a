#!/usr/bin/env python
"""CPU baseline per SURVEY.md section 8d / BASELINE.md section 3: the build's CPU restatement of the reference path
(oracle/umgen_oracle.py, PyTorch-CPU fp32; parity-checked against the imported reference) timed on the host cores for
N = 2 whole frames of the video rollout (B = 1, UMGen_Large, 20 history frames), median of 3 runs.  Per-frame cost is constant
once the window is full, so the 30-frame figure is 15 x the 2-frame time -- labelled as an extrapolation in the output.

    python tools/cpu_baseline_full.py [--runs 3] [--frames 2] [--threads 32] [--out profiles/r02_cpu_baseline_full.json]

~15 min per frame on 8 vCPUs, ~4 min on the GPU box's 2 x 64 cores: run it once per round, not inside bench.py (bench.py keeps
its bounded sample and prints this file's content next to it when it exists)."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def physical_cores():
    try:
        seen = set()
        for blk in open("/proc/cpuinfo").read().strip().split("\n\n"):
            d = dict(l.split(":", 1) for l in blk.splitlines() if ":" in l)
            seen.add((d.get("physical id\t", d.get("physical id", "0")).strip(), d.get("core id\t\t", d.get("core id", "0")).strip()))
        return len(seen)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--config", default="large", choices=["large", "tiny"])
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_baseline_full.json"))
    args = ap.parse_args()
    import torch

    from oracle.umgen_oracle import OracleUMGen
    from umgen_amd.config import SEQ_LEN, large_config, tiny_config
    from umgen_amd.synth import synthetic_scene
    from umgen_amd.weights import synthetic_state_dict

    torch.set_num_threads(args.threads)
    cfg = large_config() if args.config == "large" else tiny_config()
    T = min(20, cfg.max_frame_len - 1)
    o = OracleUMGen(cfg, synthetic_state_dict(cfg, seed=0))
    scene = synthetic_scene(0, n_frames=T)
    times = []
    for r in range(args.runs):
        t0 = time.perf_counter()
        o.inference(args.frames, T, scene, input_cond_frames=T, seed=1000)
        times.append(time.perf_counter() - t0)
        print(f"run {r}: {times[-1]:.1f} s for {args.frames} frames", flush=True)
    med = statistics.median(times)
    res = {"protocol": f"oracle/umgen_oracle.py fp32, UMGen_{args.config}, video, B=1, T={T}, new_frames={args.frames}, median of {args.runs}",
           "seconds_per_run": times, "median_s": med, "sec_per_frame": med / args.frames,
           "scene_tokens_per_s": args.frames * SEQ_LEN / med,
           "thirty_frame_rollout_s_extrapolated": 30 * med / args.frames,
           "torch_num_threads": torch.get_num_threads(), "logical_cpus": os.cpu_count(), "physical_cores": physical_cores()}
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
