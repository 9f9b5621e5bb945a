"""Build-owned counterpart of projects/tools/evaluate.py + the token-saving half of projects/tools/model_pl.py.

    python -m umgen_amd.evaluate --infer_task video --set_num_new_frames 30 --ckpt_dir data/weights/UMGen_Large.pt \
        --data_test_root data/tokenized_scenes --output_path output/UMGen/

Flags and resolution follow evaluate.py:28-133 and infer_fun.py:56-159 (set_inference_setting / set_model_config):
``video`` -> input_cond_frames 20, new frames = --set_num_new_frames; ``control`` -> input_cond_frames 13, 30 frames
(infer_fun.py:64-71); ``--model_scale larger`` -> 36 TAR layers; sampler "topk", k = 5 / 5 / 16.  The artefact is the
one parity is judged on: ``<output_path>/saved_token/<name>_tokens.pkl`` = pickle of dict mod -> np.int64
[1, T_out, S_mod] (model_pl.py:350-355), skipped when it already exists (model_pl.py:215-216).
VAE decoding / video rendering (model_pl.py:254ff) stay on the reference side.

Scenes are read from ``*.pkl`` clips in the reference's raw ``tokenized_origin_scenes`` schema (tokenised on the fly exactly as
``NuPlanTokenDataset`` + ``transforms_val`` do, see scene_io.py), from ``*.npz`` / ``*.pkl`` files that already hold the token
dict handed to ``inference`` (pose [T,3] / map [T,1024] / bbox3d [T,660] / image [T,512], optional control_pose /
control_bbox3d), or generated with ``--synthetic N``.  Under ``torchrun`` the scenes are sharded over ranks (umgen_amd/shard.py).
"""
from __future__ import annotations

import argparse
import glob
import os
import pickle
import sys

import numpy as np

from .config import MOD_ORDER, large_config, tiny_config
from .scene_io import is_raw_scene, save_tokens, scene_tokens
from .synth import synthetic_control, synthetic_scene
from .weights import expected_keys, synth_tensor


def str2bool(v):   # evaluate.py declares these flags with type=bool (True for ANY non-empty string); keep it parseable
    return str(v).lower() not in ("0", "false", "no", "")


def build_parser():
    p = argparse.ArgumentParser("umgen_amd.evaluate")
    p.add_argument("--pred_task", type=str, default="pose_map_bbox3d_image")
    p.add_argument("--ckpt_dir", type=str, default="data/weights/UMGen_Large.pt")
    p.add_argument("--model_scale", type=str, default="larger", choices=["larger", "stander", "debug"])
    p.add_argument("--infer_task", type=str, default="control")
    p.add_argument("--rule_constrain", type=str2bool, default=True)
    p.add_argument("--set_num_new_frames", type=int, default=10)
    p.add_argument("--debug", type=str2bool, default=False, help="skip the checkpoint and use random-init weights")
    p.add_argument("--output_path", default="output/UMGen/")
    p.add_argument("--launcher", type=str, default=None)
    # sampler (config.py:442-463 infer_task_config)
    p.add_argument("--top_k", type=int, default=5)
    p.add_argument("--top_p", type=float, default=0.4)
    p.add_argument("--sample_method", type=str, default="topk")
    # build-side additions
    p.add_argument("--data_test_root", type=str, default=None, help="directory of token-dict scenes (*.npz / *.pkl)")
    p.add_argument("--synthetic", type=int, default=0, help="generate N synthetic scenes instead of reading files")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"],
                   help="bf16: production; fp16: the reference's own autocast arithmetic; fp32: exact parity mode")
    p.add_argument("--batch", type=int, default=1, help="scenes rolled out together per GPU")
    p.add_argument("--seed", type=int, default=0)
    return p


def resolve(args):
    """infer_fun.set_inference_setting / set_model_config value resolution."""
    if args.infer_task == "video":
        new_frames, input_cond = args.set_num_new_frames, 20
    elif "control" in args.infer_task:
        new_frames, input_cond = 30, 13
    else:
        new_frames, input_cond = -1, 20
    if args.model_scale == "debug":
        cfg = tiny_config()
    else:
        cfg = large_config(n_tar_layer=36 if args.model_scale == "larger" else 24)
    cfg.top_k, cfg.p, cfg.sample_method, cfg.rule_constrain = args.top_k, args.top_p, args.sample_method, bool(args.rule_constrain)
    return cfg, new_frames, input_cond


def load_scene(path, block_size=42, sampling_gap=4, start_index=10):
    d = dict(np.load(path)) if path.endswith(".npz") else pickle.load(open(path, "rb"))
    if is_raw_scene(d):        # a tokenized_origin_scenes clip: what NuPlanTokenDataset + transforms_val turn it into (scene_io.py)
        d = scene_tokens(d, block_size, sampling_gap, start_index)
    if "dataset_token" in d:   # control pickle layout (model_pl.py:137-171)
        ctl = d.get("control_dict", {})
        d = dict(d["dataset_token"], **{f"control_{k}": v for k, v in ctl.items()})
    sc = {m: np.asarray(d[m]).reshape(1, -1, np.asarray(d[m]).shape[-1]).astype(np.int64) for m in MOD_ORDER}
    ctl = {k[8:]: np.asarray(v).reshape(1, -1, np.asarray(v).shape[-1]).astype(np.int64) for k, v in d.items() if k.startswith("control_")}
    return sc, (ctl or None)


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    import torch.distributed as dist

    from .engine import Engine
    from .shard import scene_partition, sharded_rollout

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:   # one process per GPU: bind the device BEFORE the RCCL communicator exists
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cfg, new_frames, input_cond = resolve(args)
    control = "control" in args.infer_task
    T_hist = min(20, cfg.max_frame_len - 1)
    if args.synthetic:
        scenes = [(f"synthetic_{i:04d}", synthetic_scene(i, n_frames=min(input_cond, T_hist)),
                   synthetic_control(i, n_frames=new_frames) if control else None) for i in range(args.synthetic)]
    else:
        files = sorted(glob.glob(os.path.join(args.data_test_root or "", "*.npz")) + glob.glob(os.path.join(args.data_test_root or "", "*.pkl")))
        if not files:
            sys.exit("no scenes: pass --data_test_root <dir with *.npz|*.pkl token dicts> or --synthetic N")
        # dataset_block_size = set_num_new_frames + cond_frames (infer_fun.py:176-186), sampling_gap 4, start_index 10 (evaluate.py:157)
        block = (args.set_num_new_frames if args.infer_task == "video" else max(new_frames, 0)) + 20
        scenes = [(os.path.basename(f)[:-4],) + load_scene(f, block_size=block) for f in files]
    save_dir = os.path.join(args.output_path, "saved_token")
    os.makedirs(save_dir, exist_ok=True)
    # skip-if-exists stays per scene (model_pl.py:215-216); every rank takes the same decision before anybody writes
    todo = []
    for sid, (name, _, _) in enumerate(scenes):
        if os.path.exists(os.path.join(save_dir, name + "_tokens.pkl")):
            if rank == 0:
                print(name, " has been processed")
        else:
            todo.append(sid)
    if world > 1:
        dist.barrier()
    if not todo:
        if world > 1:
            dist.destroy_process_group()
        return
    eng = Engine(cfg, precision=args.precision, max_batch=args.batch, max_cond_frames=T_hist, device=local_rank)
    if args.debug:
        for key, shape in expected_keys(cfg).items():
            eng.load_tensor(key, synth_tensor(key, shape, seed=0))
    else:
        ckpt = torch.load(args.ckpt_dir, map_location="cpu")
        sd = ckpt["model_state"] if "model_state" in ckpt else ckpt          # infer_fun.py:43-50
        for k, v in sd["module"].items():
            eng.load_tensor(k, v.float().numpy() if v.dtype != torch.bfloat16 else v.view(torch.int16).numpy().view(np.uint16))
    eng.finalize()
    # Scenes that can share a batch: same history length, same number of new frames, same kind of control tokens.  Each group goes
    # through umgen_amd.shard.sharded_rollout -- THE multi-scene / multi-GPU implementation (scene i -> rank i mod P, `--batch`
    # scenes per engine call, one all-gather of the sampled tokens at the end) that bench.py and the tests use too.
    groups = {}
    for sid in todo:
        _, toks, ctl = scenes[sid]
        icf = min(input_cond, toks["pose"].shape[1])
        nf = new_frames if new_frames >= 0 else toks["pose"].shape[1] - icf
        key = (icf, nf, tuple(sorted((k, v.shape[1:]) for k, v in ctl.items())) if ctl else None)
        groups.setdefault(key, []).append(sid)
    for (icf, nf, _), sids in groups.items():
        def rollout_fn(toks, seeds, scene_ids):
            ctls = [scenes[s][2] for s in scene_ids]
            init = {k: np.concatenate([c[k] for c in ctls]) for k in ctls[0]} if ctls[0] else None
            return eng.rollout(toks, nf, cond_frames=T_hist, input_cond_frames=icf, init_tokens=init,
                               control_test=control and init is not None and "bbox3d" in init, seeds=seeds)
        out = sharded_rollout(rollout_fn, [scenes[s][1] for s in sids], base_seed=args.seed, batch=args.batch,
                              device="cuda" if world > 1 else "cpu", scene_ids=sids, pass_ids=True)
        for pos in scene_partition(len(sids), world, rank):                   # every rank writes the scenes it rolled out
            name = scenes[sids[pos]][0]
            print("saved", save_tokens({m: out[m][pos:pos + 1] for m in MOD_ORDER}, args.output_path, name))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
