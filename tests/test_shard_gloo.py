"""world_size-2 gloo test of the N>1 path: static scene partition + ONE all-gather of sampled tokens must reproduce
the unsharded result exactly (per-scene seeds keyed by scene id => invariant to the number of ranks)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from umgen_amd.config import CONTENT_LEN, MOD_ORDER
from umgen_amd.shard import scene_partition, sharded_rollout


def fake_rollout(tokens, seeds, new_frames):
    """Deterministic stand-in for Engine.rollout: tokens of scene s, frame t depend only on (seed, t)."""
    B, T = tokens["pose"].shape[:2]
    out = {}
    for m in MOD_ORDER:
        new = np.stack([np.stack([np.random.default_rng([seeds[b], t, len(m)]).integers(0, 1000, CONTENT_LEN[m])
                                  for t in range(new_frames)]) for b in range(B)])
        out[m] = np.concatenate([tokens[m], new], axis=1).astype(np.int64)
    return out


def scenes(n):
    return [{m: np.full((1, 2, CONTENT_LEN[m]), i, dtype=np.int64) for m in MOD_ORDER} for i in range(n)]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = sharded_rollout(fake_rollout, scenes(n), base_seed=100, batch=2, new_frames=3)
    if rank == 0:
        q.put({m: res[m] for m in MOD_ORDER})
    dist.destroy_process_group()


def test_partition_is_static_round_robin():
    assert scene_partition(5, 2, 0) == [0, 2, 4] and scene_partition(5, 2, 1) == [1, 3]


def test_two_rank_gloo_sharded_equals_unsharded():
    n = 5   # uneven: rank 0 gets 3 scenes, rank 1 gets 2 (exercises the padded all-gather)
    ref = sharded_rollout(fake_rollout, scenes(n), base_seed=100, batch=1, new_frames=3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for m in MOD_ORDER:
        assert got[m].shape == (n, 5, CONTENT_LEN[m])
        np.testing.assert_array_equal(got[m], ref[m])
