#!/usr/bin/env python
"""bench.py -- UMGen_Large next-scene rollout throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one generated frame of the video rollout for the rank's scene batch: ego net + 3 TAR stacks over the
20-frame history window + the 2206-step OAR decode loop (one pass of the hot path, UMGen._inference).  The default
K = 30 steps is BASELINE.json configs[1]: ``UMGen_Large --infer_task video --set_num_new_frames 30, batch=1, bf16``.
Each rank rolls out its own independent scenes (weak scaling, no data-path collective); the sampled tokens are
all-gathered once at the end (RCCL) inside the timed region.  Weights are random-init (PyTorch-default-like) of the
UMGen_Large architecture and inputs are synthetic tokenized_origin_scenes-shaped tokens: no checkpoint/dataset offline.

Rank 0 prints ONE JSON line with the contract fields plus ``roofline`` (dominant kernel) and ``cpu_baseline``.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from umgen_amd.config import MOD_ORDER, SEQ_LEN, large_config, tiny_config, wide2x_config  # noqa: E402
from umgen_amd.synth import synthetic_scene  # noqa: E402
from umgen_amd.weights import expected_keys, synth_tensor  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA peak


def pmc_traffic_per_step(cfg):
    """HBM bytes per decode step from the committed rocprofv3 PMC pass (profiles/r01_pmc_fetch_size.csv: mean FETCH_SIZE [KB]
    per launch of each kernel at KV length ~2000; x2 = the gfx950 correction of MI355X_MICROARCH.md for wide streaming reads).
    PMC serialises every dispatch, so it cannot be collected inside the timed region; null when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_size.csv")
    if not os.path.exists(path):
        return None
    per_kernel = {}
    for line in open(path).read().splitlines()[1:]:
        name, _, rest = line.rpartition('",')
        f = rest.split(",")
        per_kernel[name.strip('"')] = float(f[2])
    def mean(sub):
        for k, v in per_kernel.items():
            if sub in k:
                return v
        return 0.0
    L = cfg.n_oar_layer
    kb = L * (2 * mean("gemv_ln_kernel<unsigned short, 1") + mean("attn_partial_kernel<unsigned short>") +
              mean("gemv_resid_kernel<unsigned short, 1, 6, false>") + mean("gemv_resid_kernel<unsigned short, 1, 2, true>"))
    kb += mean("gemv_ln_kernel<unsigned short, 1")      # head GEMV
    return kb * 2.0 * 1024.0


def cpu_baseline(cfg_name: str, threads: int):
    """The CPU oracle ("port" of the reference path) timed on this box's host cores on a BOUNDED sample:
    one full-size BlockTAR per stack sequence length (S = 1031, 1693, 2207; T = 20) and 16 OAR decode steps at
    KV length 1100, scaled by the per-frame block / step counts of UMGen_Large."""
    import torch

    from oracle.umgen_oracle import OracleUMGen
    from umgen_amd.weights import synthetic_state_dict

    full = large_config() if cfg_name == "large" else tiny_config()
    one = type(full)(**{**full.__dict__, "n_ego_tar_layer": 1, "n_ego_ca_layer": 1, "n_map_tar_layer": 1,
                        "n_box_tar_layer": 1, "n_tar_layer": 1, "n_oar_layer": 1})
    torch.set_num_threads(threads)
    o = OracleUMGen(one, synthetic_state_dict(one, seed=0))
    E, T = full.n_embd, 20
    t_blk = {}
    with torch.no_grad():
        for S in (1031, 1693, 2207):
            x = torch.randn(1, T, S, E)
            t0 = time.perf_counter()
            o._block_tar(x, "transformer.TAR.0")
            t_blk[S] = time.perf_counter() - t0
        kv = (torch.randn(1, 1100, E), torch.randn(1, 1100, E))
        x = torch.randn(1, 1, E)
        t0 = time.perf_counter()
        for _ in range(16):
            o._block_oar(x, "transformer.OAR.0", kv)
        t_oar = (time.perf_counter() - t0) / 16
    frame_s = (t_blk[2207] * (full.n_ego_tar_layer + full.n_tar_layer) + t_blk[1031] * full.n_map_tar_layer
               + t_blk[1693] * full.n_box_tar_layer + t_oar * full.n_oar_layer * 2206)
    return {"value": SEQ_LEN / frame_s, "unit": "scene-tokens/s", "cores": threads, "kind": "port",
            "sample": ("oracle/umgen_oracle.py (PyTorch-CPU fp32): 1 BlockTAR at S=1031/1693/2207 x T=20 "
                       f"({t_blk[1031]:.1f}/{t_blk[1693]:.1f}/{t_blk[2207]:.1f} s) + 16 BlockOAR steps at L=1100 "
                       f"({t_oar * 1e3:.2f} ms/step), scaled by UMGen_Large block/step counts -> {frame_s:.0f} s/frame")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="scenes per GPU (configs[1] is batch=1)")
    ap.add_argument("--config", default="large", choices=["large", "tiny", "wide2x"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--history", type=int, default=20, help="history frames T (configs[4]: 40 = doubled context)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from umgen_amd.engine import Engine

    cfg = {"large": large_config, "tiny": tiny_config, "wide2x": wide2x_config}[args.config]()
    if os.environ.get("UMGEN_BENCH_OAR_LAYERS"):      # experiment knob (not a bench configuration)
        cfg.n_oar_layer = int(os.environ["UMGEN_BENCH_OAR_LAYERS"])
    T = min(args.history, cfg.max_frame_len - 1)
    B = args.batch
    eng = Engine(cfg, precision=args.precision, max_batch=B, max_cond_frames=T, device=local_rank, use_graphs=not args.no_graphs)
    t_load = time.perf_counter()
    for key, shape in expected_keys(cfg).items():      # random-init weights, streamed one tensor at a time
        eng.load_tensor(key, synth_tensor(key, shape, seed=0))
    eng.finalize()
    t_load = time.perf_counter() - t_load
    scenes = [synthetic_scene(rank * B + i, n_frames=T) for i in range(B)]
    tokens = {m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}
    seeds = [1000 + rank * B + i for i in range(B)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        eng.rollout(tokens, args.warmup, cond_frames=T, input_cond_frames=T, seeds=seeds)
    sync()
    t0 = time.perf_counter()
    out = eng.rollout(tokens, args.steps, cond_frames=T, input_cond_frames=T, seeds=seeds)
    if world > 1:   # the one exchange of the path: all-gather the sampled tokens (north_star)
        flat = torch.from_numpy(np.concatenate([out[m][:, T:].reshape(B, -1) for m in MOD_ORDER], axis=1).astype(np.int32)).cuda()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    tm = eng.timings()

    if rank == 0:
        # kernel-level rooflines of the prefill side: one extra frame with per-launch HIP events (the engine runs the whole
        # window in the foreground then, on all its CUs, so the launches are timed alone)
        eng.set_profiling(True)
        eng.rollout(tokens, 1, cond_frames=T, input_cond_frames=T, seeds=seeds)
        tp = eng.timings()
        eng.set_profiling(False)
        for k in ("gemm_ms", "gemm_flops", "gemm_launches", "attn_ms", "attn_flops", "attn_launches"):
            tm[k] = tp[k]
        total_scenes = B * world
        value = total_scenes * args.steps * SEQ_LEN / dt
        frames = max(1, tm["frames"])
        gemm_tfs = (tm["gemm_flops"] / (tm["gemm_ms"] * 1e-3) / 1e12) if tm["gemm_ms"] > 0 else 0.0
        attn_tfs = (tm["attn_flops"] / (tm["attn_ms"] * 1e-3) / 1e12) if tm["attn_ms"] > 0 else 0.0
        steps_total = max(1, tm["oar_steps"])
        step_us = tm["oar_ms"] * 1e3 / steps_total
        bytes_per_step = tm["oar_bytes"] / steps_total / B        # algorithmic: weights once + KV read/write
        oar_gbs = (tm["oar_bytes"] / (tm["oar_ms"] * 1e-3) / 1e9) if tm["oar_ms"] > 0 else 0.0
        res = {
            "metric": "scene_tokens_per_sec", "value": value, "unit": "scene-tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic tokenized_origin_scenes-shaped tokens; random-init weights (no checkpoint offline)",
            "config": {"workload": f"UMGen_{args.config} --infer_task video, {args.steps}-frame rollout, "
                                   f"{B} scene(s)/GPU, T={T} history frames, top-k 5/5/16 sampling, rule_constrain",
                       "scenes_per_gpu": B, "history_frames": T, "sec_per_frame": dt / args.steps},
            # dominant unit of work (~80 % of the frame): the OAR decode step = 36 x (gemv_ln, attn_partial, gemv_resid,
            # gemv_ln, gemv_resid) + head + sampler, replayed from a hipGraph; HIP events bracket the decode phase of every frame
            "roofline": {"bound": "hbm", "achieved": oar_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": oar_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic_per_step(cfg),
                         "kernel": "OAR decode step (gemv_ln_kernel x73, attn_partial_kernel x36, gemv_resid_kernel x72, sample_token_kernel)",
                         "launches": tm["oar_kernels"], "avg_launch_us": tm["oar_ms"] * 1e3 / max(1, tm["oar_kernels"]),
                         "avg_step_us": step_us, "algorithmic_bytes_per_step": bytes_per_step,
                         # the timed region runs the next frame's history slots beside the decode loop (2 of the 8 XCDs); the
                         # same loop alone on the chip, from the extra profiled frame:
                         "achieved_alone": tp["oar_bytes"] / (tp["oar_ms"] * 1e-3) / 1e9 if tp["oar_ms"] > 0 else None,
                         "avg_step_us_alone": tp["oar_ms"] * 1e3 / max(1, tp["oar_steps"])},
            "roofline_gemm": {"bound": "mfma", "achieved": gemm_tfs, "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                              "frac": gemm_tfs / MFMA_BF16_PEAK_TFS, "kernel": "gemm_bf16_glds_kernel (TAR/ego stacks)",
                              "launches": tm["gemm_launches"], "avg_launch_ms": tm["gemm_ms"] / max(1, tm["gemm_launches"])},
            "roofline_attn": {"bound": "mfma", "achieved": attn_tfs, "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                              "frac": attn_tfs / MFMA_BF16_PEAK_TFS, "kernel": "attn_spatial_mfma_kernel",
                              "launches": tm["attn_launches"], "avg_launch_ms": tm["attn_ms"] / max(1, tm["attn_launches"])},
            # foreground stream per frame; "background" = the next frame's history slots pushed through the stacks on the
            # CU-masked second stream while the decode loop runs
            "phases_ms_per_frame": {"ego": tm["ego_ms"] / frames, "tar": tm["tar_ms"] / frames, "oar": tm["oar_ms"] / frames,
                                    "background_per_pass": tm["bg_ms"] / max(1, tm["overlapped_frames"]),
                                    "overlapped_frames": tm["overlapped_frames"]},
            "prefill_ms_unoverlapped": {"ego": tp["ego_ms"], "tar": tp["tar_ms"]},
            "weight_load_s": t_load,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.config, threads=min(32, os.cpu_count() or 1))   # 32 threads is the measured optimum on the 2x64-core host (more threads are slower)
        print(json.dumps(res))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
