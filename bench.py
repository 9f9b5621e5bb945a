#!/usr/bin/env python
"""bench.py -- UMGen_Large next-scene rollout throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one generated frame of the video rollout for the rank's scene batch: ego net + 3 TAR stacks over the
20-frame history window + the 2206-step OAR decode loop (one pass of the hot path, UMGen._inference).  The default
K = 30 steps is BASELINE.json configs[1]: ``UMGen_Large --infer_task video --set_num_new_frames 30, batch=1, bf16``.
Scene i runs on rank i mod P (umgen_amd/shard.py: the same static partition + one all-gather the evaluate CLI and the tests
use): every rank rolls out its own independent scenes (weak scaling, no data-path collective) and the sampled tokens are
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
from umgen_amd.synth import synthetic_control, synthetic_given_map, synthetic_scene  # noqa: E402
from umgen_amd.weights import expected_keys, synth_tensor  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA peak


PMC_FILES = ["r06_pmc_fetch_size_engine.csv", "r05_pmc_fetch_size_engine.csv", "r04_pmc_fetch_size_engine.csv", "r03_pmc_fetch_size_engine.csv"]   # newest first
PMC_FILES_BATCH = {4: ["r06_pmc_fetch_size_engine_b4.csv"], 8: ["r06_pmc_fetch_size_engine_b8.csv"]}      # configs[2] / [3]: scenes per GPU


def pmc_traffic_per_launch(engine_on: bool, files=None, kernel="oar_engine_kernel"):
    """(HBM bytes per launch of the dominant kernel, source) from the newest committed rocprofv3 PMC pass (tools/gpu_session.sh pmc:
    `rocprofv3 --pmc FETCH_SIZE`, decode steps 1101..1104 only via UMGEN_DEBUG_OAR_STEPS, i.e. at the MEAN KV length of a frame --
    the same L the algorithmic bytes per launch are quoted at; mean FETCH_SIZE [KB] x 2 = the gfx950 correction of
    MI355X_MICROARCH.md for wide streaming reads).  PMC serialises every dispatch, so it is a separate pass and NOT a measurement of
    this run: `roofline.traffic_source` says which file the number comes from.  (None, None) when absent or the engine did not run."""
    if not engine_on:
        return None, None
    for fn in (files or PMC_FILES):
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            continue
        for line in open(path).read().splitlines()[1:]:
            name, _, rest = line.rpartition('",')
            if kernel in name:
                return float(rest.split(",")[2]) * 2.0 * 1024.0, f"profiles/{fn} (separate rocprofv3 --pmc FETCH_SIZE pass at KV length 1101..1104, x2 gfx950 correction; not measured in this run)"
    return None, None


def closed_loop_record(precision: str):
    """Greedy closed-loop token agreement of the timed precision with the fp32 engine (frames 0 / 2 / 8), from the newest committed
    tools/closed_loop.py run -- a 30-frame fp32 rollout does not fit a bench run; `source` names the file."""
    for fn in ("r05_closed_loop.json", "r04_closed_loop.json", "r03_closed_loop.json"):
        path = os.path.join(ROOT, "profiles", fn)
        if os.path.exists(path):
            try:
                d = json.load(open(path)).get(precision)
                if d:
                    ag = d.get("per_frame_token_agreement") or []
                    return {"vs": "fp32 engine, greedy, same weights / scene", "frame0": ag[0] if len(ag) > 0 else None,
                            "frame2": ag[2] if len(ag) > 2 else None, "frame8": ag[8] if len(ag) > 8 else None,
                            "first_divergence": d.get("first_divergence"), "source": f"profiles/{fn} (not measured in this run)"}
            except Exception:
                pass
    return None


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
    sample = {"value": SEQ_LEN / frame_s, "extrapolated": True,
              "what": ("oracle/umgen_oracle.py (PyTorch-CPU fp32): 1 BlockTAR at S=1031/1693/2207 x T=20 "
                       f"({t_blk[1031]:.1f}/{t_blk[1693]:.1f}/{t_blk[2207]:.1f} s) + 16 BlockOAR steps at L=1100 "
                       f"({t_oar * 1e3:.2f} ms/step), scaled by UMGen_Large block/step counts -> {frame_s:.0f} s/frame")}
    res = {"value": sample["value"], "unit": "scene-tokens/s", "cores": threads, "kind": "port", "measured_in": "this run",
           "extrapolated_from_sample": True, "sample": sample["what"]}
    for fn in ("r04_cpu_baseline_full.json", "r02_cpu_baseline_full.json"):
        fp = os.path.join(ROOT, "profiles", fn)
        if cfg_name == "large" and os.path.exists(fp):
            # whole-frame measurement of the same oracle on a GPU box's host (tools/cpu_baseline_full.py; SURVEY.md section 8d protocol):
            # ~15 min per frame, so it is a separate run, quoted beside the bounded sample this run timed -- never in place of it
            full = json.load(open(fp))
            res["whole_frame_measurement"] = {"value": full["scene_tokens_per_s"], "unit": "scene-tokens/s", "cores": full.get("torch_num_threads"),
                                              "protocol": full.get("protocol"), "source": f"profiles/{fn} (cached measurement, not this run)"}
            break
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="scenes per GPU (configs[1] is batch=1)")
    ap.add_argument("--config", default="large", choices=["large", "tiny", "wide2x"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="bf16: BASELINE.json configs[1]; fp16: the same kernels with IEEE-half operands (the reference's autocast dtype)")
    ap.add_argument("--history", type=int, default=20, help="history frames T (configs[4]: 40 = doubled context)")
    ap.add_argument("--task", default="video", choices=["video", "control", "mapgiven"],
                    help="control: ego pose + one agent slot of every new frame given (configs[2] with --batch 4); mapgiven: the map of every new "
                         "frame given (infer_oar_net's predefined-token prefix, UMGen.py:1184-1201); the bench line is video")
    ap.add_argument("--input-history", type=int, default=0,
                    help="history frames the rollout STARTS with (0 = the reference's setting: --history for video, 13 for control, "
                         "infer_fun.py:64-71 -- the control window then grows 13 -> 20 before it slides)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="dry run of the N > 1 path on ONE GPU: every rank uses cuda:0 and the one exchange runs over gloo (RCCL refuses "
                         "two ranks on a device).  Exercises the launch contract, the partition, the gather, the max-over-ranks clock and "
                         "the JSON line -- NOT a scaling measurement (the line says so); use --config tiny: two XCD-resident decode "
                         "engines cannot share a GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="world size 1 only: still create the RCCL process group (backend nccl, device_id = this GPU) and run the path's "
                         "barrier, device all-gather and max all-reduce on it -- the hardware execution of the N > 1 communication code a "
                         "one-GPU box allows (the line carries rccl_exercised)")
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

    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dist = args.force_dist and world == 1 and not args.share_gpu
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist_on = world > 1 or force_dist
    if dist_on:
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from umgen_amd.engine import Engine

    cfg = {"large": large_config, "tiny": tiny_config, "wide2x": wide2x_config}[args.config]()
    if os.environ.get("UMGEN_BENCH_OAR_LAYERS"):      # experiment knob (not a bench configuration)
        cfg.n_oar_layer = int(os.environ["UMGEN_BENCH_OAR_LAYERS"])
    T = min(args.history, cfg.max_frame_len - 1)
    T_in = min(T, args.input_history if args.input_history > 0 else (13 if args.task == "control" else T))
    B = args.batch
    eng = Engine(cfg, precision=args.precision, max_batch=B, max_cond_frames=T, device=local_rank, use_graphs=not args.no_graphs)
    t_load = time.perf_counter()
    for key, shape in expected_keys(cfg).items():      # random-init weights, streamed one tensor at a time
        eng.load_tensor(key, synth_tensor(key, shape, seed=0))
    eng.finalize()
    t_load = time.perf_counter() - t_load
    from umgen_amd.shard import scene_partition, scene_seed, sharded_rollout

    n_scenes = B * world                     # scene i -> rank i mod P (shard.py): B scenes per rank, weak scaling
    scenes = [synthetic_scene(i, n_frames=T) for i in range(n_scenes)]
    mine = scene_partition(n_scenes, world, rank)
    tokens = {m: np.concatenate([scenes[i][m] for i in mine]) for m in MOD_ORDER}
    seeds = [scene_seed(1000, i) for i in mine]

    def control_of(ids, new_frames):
        if args.task == "mapgiven":
            return {"init_tokens": {"map": np.concatenate([synthetic_given_map(i, n_frames=new_frames)["map"] for i in ids])}}
        if args.task != "control":
            return {}
        ctl = [synthetic_control(i, n_frames=new_frames) for i in ids]
        return {"init_tokens": {k: np.concatenate([c[k] for c in ctl]) for k in ("pose", "bbox3d")}, "control_test": True}

    def rollout_fn(toks, seeds, new_frames, scene_ids=None):
        return eng.rollout(toks, new_frames, cond_frames=T, input_cond_frames=T_in, seeds=seeds,
                           **control_of(scene_ids if scene_ids is not None else mine, new_frames))

    gdev = "cpu" if (not dist_on or args.share_gpu) else "cuda"      # where the one all-gather of the path runs (RCCL: device buffers)

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        rollout_fn(tokens, seeds, args.warmup)
    sync()
    t0 = time.perf_counter()
    # the timed region: every rank's rollouts + the one exchange of the path (all-gather of the sampled tokens, north_star)
    out = sharded_rollout(rollout_fn, scenes, base_seed=1000, batch=B, device=gdev, pass_ids=True, force_collective=force_dist,
                          new_frames=args.steps)
    assert out["map"].shape[0] == n_scenes
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=gdev if dist_on else "cuda")
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    tm = eng.timings()

    if rank == 0:
        # kernel-level rooflines of the prefill side: one extra frame with per-launch HIP events (the engine runs the whole
        # window in the foreground then, on all its CUs, so the launches are timed alone)
        eng.set_profiling(True)
        eng.rollout(tokens, 1, cond_frames=T, input_cond_frames=T, seeds=seeds, **control_of(mine, 1))
        tp = eng.timings()
        eng.set_profiling(False)
        for k in ("gemm_ms", "gemm_flops", "gemm_launches", "attn_ms", "attn_flops", "attn_launches", "layers_ms", "layers_launches"):
            tm[k] = tp[k]
        total_scenes = B * world
        value = total_scenes * args.steps * SEQ_LEN / dt
        frames = max(1, tm["frames"])
        gemm_tfs = (tm["gemm_flops"] / (tm["gemm_ms"] * 1e-3) / 1e12) if tm["gemm_ms"] > 0 else 0.0
        attn_tfs = (tm["attn_flops"] / (tm["attn_ms"] * 1e-3) / 1e12) if tm["attn_ms"] > 0 else 0.0
        steps_total = max(1, tm["oar_steps"])
        step_us = tm["oar_ms"] * 1e3 / steps_total
        engine_on = bool(tm["decode_engine"])
        # dominant kernel: the decode step's layer kernel(s).  Algorithmic bytes per step = the OAR weights once + the KV rows of
        # every scene (the head / sampler launches of the step are not part of this kernel).  Its duration is
        # measured per launch with HIP events on the decode stream in the extra profiled frame (eager launches, nothing else runs).
        wsz = 4 if args.precision == "fp32" else 2
        # AR heads: one GEMV per sampled step; head_tar_bbox3d: ONE GEMM per frame (tar_head_logits) -- per step, frame average
        head_bytes = (cfg.map_vocab_size * 1024 + cfg.bbox3d_vocab_size * (660 + 1) + cfg.img_vocab_size * 512) * cfg.n_embd * wsz / 2206.0
        bytes_per_step = tm["oar_bytes"] / steps_total        # whole batch
        layer_bytes = bytes_per_step - head_bytes
        layers_us = tp["layers_ms"] * 1e3 / max(1, tp["layers_launches"])
        ach = layer_bytes / (layers_us * 1e-6) / 1e9 if layers_us > 0 else 0.0
        oar_gbs = (tm["oar_bytes"] / (tm["oar_ms"] * 1e-3) / 1e9) if tm["oar_ms"] > 0 else 0.0
        traffic, traffic_src = pmc_traffic_per_launch(engine_on and (B == 1 or B in PMC_FILES_BATCH) and args.config == "large", PMC_FILES_BATCH.get(B))
        lanes = int(tm.get("decode_lanes", 0))
        if int(tm["decode_engine"]) == 3:
            kname = ("umgen::oar_engine_wide_kernel (chip-wide decode engine of the 2x-width layers: the 36 BlockOAR layers of a decode step of one scene "
                     "in one launch of 256 workgroups)")
            if B == 1 and args.config == "wide2x":
                traffic, traffic_src = pmc_traffic_per_launch(True, ["r05_pmc_fetch_size_wide_engine.csv"], "oar_engine_wide_kernel")
        elif engine_on:
            kname = "umgen::oar_engine_kernel (XCD-resident decode engine: the 36 BlockOAR layers of a decode step in one launch)"
        elif tm.get("decode_batched"):
            kname = ("batched decode layer (rows_mfma_kernel x144 + attn_decode_batched_kernel x36 per step and lane; scenes as the MFMA's B-columns)" +
                     (f", {lanes} decode lanes on their own streams: avg_launch_us = lane 0's whole step (layers + head + sampler) in the profiled frame, "
                      "all lanes running" if lanes > 1 else ""))
        else:
            kname = "OAR decode layers as launches (gemv_ln_kernel x72, attn_partial_kernel x36, gemv_resid_kernel x72 per step)"
        res = {
            "metric": "scene_tokens_per_sec", "value": value, "unit": "scene-tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic tokenized_origin_scenes-shaped tokens; random-init weights (no checkpoint offline)",
            "config": {"workload": f"UMGen_{args.config} --infer_task {args.task}, {args.steps}-frame rollout, "
                                   f"{B} scene(s)/GPU, T={T} history frames" + (f" (starting from {T_in})" if T_in != T else "") +
                                   ", top-k 5/5/16 sampling, rule_constrain",
                       "scenes_per_gpu": B, "history_frames": T, "input_history_frames": T_in, "sec_per_frame": dt / args.steps},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": kname,
                         "avg_launch_us": layers_us, "launches_timed": tp["layers_launches"],
                         "algorithmic_bytes_per_launch": layer_bytes, "scenes_per_launch": B,
                         # the whole decode step (layer kernel + head GEMV + sampler, replayed from a hipGraph), timed region:
                         "step": {"avg_step_us": step_us, "algorithmic_bytes_per_step": bytes_per_step, "achieved": oar_gbs,
                                  "frac": oar_gbs / HBM_PEAK_GBS, "kernels_per_step": tm["oar_kernels"] / steps_total}},
            "roofline_gemm": {"bound": "mfma", "achieved": gemm_tfs, "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                              "frac": gemm_tfs / MFMA_BF16_PEAK_TFS, "kernel": ("gemm16_256_kernel (256 x 256 tiles; launches under 300 tiles: gemm_bf16_pers_kernel / gemm_bf16_glds_kernel), TAR / ego stacks"
                                         if args.precision != "fp32" else "gemm_f32_mfma_kernel (v_mfma_f32_32x32x2_f32), TAR / ego stacks"),
                              "launches": tm["gemm_launches"], "avg_launch_ms": tm["gemm_ms"] / max(1, tm["gemm_launches"]),
                              "clock_note": "peak is the nominal 2.4 GHz figure; inside these kernels' k-loops the chip sustains 1.46-1.71 GHz "
                                            "(profiles/r04_gemm_stamps_clock.txt, measurement build; not measured in this run)"},
            "roofline_attn": {"bound": "mfma", "achieved": attn_tfs, "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                              "frac": attn_tfs / MFMA_BF16_PEAK_TFS, "kernel": "attn_spatial_mfma_kernel",
                              "launches": tm["attn_launches"], "avg_launch_ms": tm["attn_ms"] / max(1, tm["attn_launches"])},
            # foreground stream per frame; "background" = the next frame's history slots pushed through the stacks on the CU-masked
            # second stream while the decode loop runs (only with UMGEN_OVERLAP=1, i.e. without the engine)
            "phases_ms_per_frame": {"ego": tm["ego_ms"] / frames, "tar": tm["tar_ms"] / frames, "oar": tm["oar_ms"] / frames,
                                    "background_per_pass": tm["bg_ms"] / max(1, tm["overlapped_frames"]),
                                    "overlapped_frames": tm["overlapped_frames"],
                                    # frames that pushed only their new last slot through the stacks (growing window, slot caches: f-3)
                                    "frames_reusing_slot_caches": tm["overlapped_frames"]},
            "prefill_ms_unoverlapped": {"ego": tp["ego_ms"], "tar": tp["tar_ms"]},
            "weight_load_s": t_load,
            "closed_loop": closed_loop_record(args.precision) if args.config == "large" else None,
            "decode_engine": int(tm["decode_engine"]), "engine_fallback": int(tm["engine_fallback"]), "decode_batched": int(tm.get("decode_batched", 0)),
            "decode_lanes": lanes,
        }
        if force_dist:
            res["rccl_exercised"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "collective_device": gdev,
                                     "calls": "init_process_group(nccl, device_id), barrier, all_reduce(MAX) x2, all_gather of the int32 token buffer -- inside the timed region"}
        if args.share_gpu:
            res["dry_run"] = f"{world} ranks SHARING cuda:0 over gloo: exercises the N > 1 code path only; value is not a scaling measurement"
        if tm["engine_fallback"]:
            print("bench.py: WARNING -- the XCD-resident decode engine was NOT used (census failed at umgen_create): this line measures the "
                  "five-launch decode layer, not the production path", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            # 32 threads is the measured optimum on the 2x64-core host (more threads are slower): "cores" = threads actually used
            res["cpu_baseline"] = cpu_baseline(args.config, threads=min(32, os.cpu_count() or 1))
        print(json.dumps(res))
    eng.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
