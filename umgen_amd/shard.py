"""Multi-GPU sharding of independent scene rollouts.

Scenes never interact (the reference rolls out one scene at a time), so the path is partitioned statically:
scene ``i`` -> rank ``i mod P``; weights are replicated; there is NO per-step communication.  The only exchange is one
all-gather of the sampled tokens at the end (RCCL over xGMI on the GPU box; gloo in the CPU tests).
Per-scene RNG seeds are keyed by scene id, so results are invariant to P and to the per-rank batch size.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .config import CONTENT_LEN, MOD_ORDER


def scene_partition(n_scenes: int, world: int, rank: int) -> List[int]:
    return list(range(rank, n_scenes, world))


def scene_seed(base_seed: int, scene_id: int) -> int:
    return int(base_seed) + int(scene_id)


def pack_tokens(out: Dict[str, np.ndarray]) -> np.ndarray:
    """mod -> [B, T, S_mod]  ->  int32 [B, T, 2199] (pose | map | bbox3d | image)."""
    return np.concatenate([np.asarray(out[m]) for m in MOD_ORDER], axis=-1).astype(np.int32)


def unpack_tokens(flat: np.ndarray) -> Dict[str, np.ndarray]:
    res, o = {}, 0
    for m in MOD_ORDER:
        res[m] = np.ascontiguousarray(flat[..., o:o + CONTENT_LEN[m]]).astype(np.int64)
        o += CONTENT_LEN[m]
    return res


def gather_scene_tokens(local_out: Dict[str, np.ndarray], local_ids: Sequence[int], n_scenes: int,
                        device: str = "cpu", force_collective: bool = False) -> Dict[str, np.ndarray]:
    """All ranks receive mod -> int64 [n_scenes, T, S_mod] in scene-id order.  One all_gather of an int32 buffer
    (ranks with fewer scenes pad to the largest per-rank count).  ``force_collective``: run the all_reduce / all_gather even in a
    one-rank process group (bench.py --force-dist: the only way to execute the RCCL calls of this path on a one-GPU box)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        return {m: np.asarray(local_out[m]).astype(np.int64) for m in MOD_ORDER}
    per_rank = (n_scenes + world - 1) // world
    flat = pack_tokens(local_out) if len(local_ids) else None
    T = flat.shape[1] if flat is not None else 0
    tt = torch.tensor([T], dtype=torch.int64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    T = int(tt.item())
    buf = torch.full((per_rank, T, sum(CONTENT_LEN.values())), -1, dtype=torch.int32, device=device)
    if flat is not None:
        buf[:len(local_ids)] = torch.from_numpy(flat).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    full = np.empty((n_scenes, T, buf.shape[-1]), dtype=np.int32)
    for r in range(world):
        ids = scene_partition(n_scenes, world, r)
        if ids:
            full[ids] = parts[r][:len(ids)].cpu().numpy()
    return unpack_tokens(full)


def sharded_rollout(rollout_fn, scenes: Sequence[Dict[str, np.ndarray]], base_seed: int, batch: int = 1,
                    device: str = "cpu", scene_ids: Optional[Sequence[int]] = None, pass_ids: bool = False,
                    force_collective: bool = False, **kw) -> Dict[str, np.ndarray]:
    """Runs ``rollout_fn(tokens[B,...], seeds=[...], **kw)`` on this rank's scenes in batches of ``batch`` and gathers all scenes.

    ``scene_ids``: global ids of the entries of ``scenes`` (default 0..n-1) -- the RNG seeds are keyed by them, so a filtered
    scene list (the CLI's skip-if-exists rule) draws the same tokens as the full one.  ``pass_ids``: also hand the chunk's
    global ids to ``rollout_fn`` (per-scene control tokens)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    gids = list(scene_ids) if scene_ids is not None else list(range(len(scenes)))
    assert len(gids) == len(scenes)
    ids = scene_partition(len(scenes), world, rank)
    outs = []
    for i in range(0, len(ids), batch):
        chunk = ids[i:i + batch]
        toks = {m: np.concatenate([scenes[s][m] for s in chunk]) for m in MOD_ORDER}
        extra = {"scene_ids": [gids[s] for s in chunk]} if pass_ids else {}
        outs.append(rollout_fn(toks, seeds=[scene_seed(base_seed, gids[s]) for s in chunk], **extra, **kw))
    local = {m: np.concatenate([o[m] for o in outs]) for m in MOD_ORDER} if outs else {m: np.zeros((0, 0, CONTENT_LEN[m]), np.int64) for m in MOD_ORDER}
    return gather_scene_tokens(local, ids, len(scenes), device=device, force_collective=force_collective)
