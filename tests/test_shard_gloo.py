"""world_size-2 gloo test of the N>1 path: static scene partition + ONE all-gather of sampled tokens must reproduce
the unsharded result exactly (per-scene seeds keyed by scene id => invariant to the number of ranks)."""
import os
import socket

import numpy as np
import pytest
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


def _worker(rank, world, port, n, q, batch=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = sharded_rollout(fake_rollout, scenes(n), base_seed=100, batch=batch, new_frames=3)
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


@pytest.mark.parametrize("n", [64, 61])
def test_eight_rank_gloo_partition_of_configs3_equals_unsharded(n):
    """BASELINE.json configs[3]: 64 independent scenes over 8 ranks, 8 per rank as ONE engine batch (scene i -> rank i mod 8), one
    all-gather at the end -- and an uneven 61-scene case (ranks 5..7 own 7 scenes: the padded gather and a short last batch).
    Sharded over 8 gloo ranks == the unsharded one-scene-at-a-time result, token for token (per-scene seeds keyed by scene id)."""
    ref = sharded_rollout(fake_rollout, scenes(n), base_seed=100, batch=1, new_frames=3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, n, q, 8)) for r in range(8)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [len(scene_partition(n, 8, r)) for r in range(8)] == [n // 8 + (1 if r < n % 8 else 0) for r in range(8)]
    for m in MOD_ORDER:
        assert got[m].shape == (n, 5, CONTENT_LEN[m])
        np.testing.assert_array_equal(got[m], ref[m])


# ---------------------------------------------------------------------------------------------------------------------
# the same path with the REAL engine (-m gpu): two ranks share cuda:0, each owns its scenes, one all-gather at the end
# ---------------------------------------------------------------------------------------------------------------------
def _real_scenes(n):
    from umgen_amd.synth import synthetic_scene
    return [synthetic_scene(60 + i, n_frames=2) for i in range(n)]


def _real_rollout(n, batch):
    """umgen_amd.shard.sharded_rollout over Engine.rollout on this process's GPU (what bench.py / the evaluate CLI run per rank)."""
    from umgen_amd.config import tiny_config
    from umgen_amd.engine import Engine
    from umgen_amd.weights import synthetic_state_dict

    cfg = tiny_config()
    e = Engine(cfg, precision="fp32", max_batch=batch, max_cond_frames=3, device=0)
    e.load_state_dict(synthetic_state_dict(cfg, seed=4))
    e.finalize()

    def fn(toks, seeds, new_frames):
        return e.rollout(toks, new_frames, cond_frames=3, input_cond_frames=2, seeds=seeds)

    res = sharded_rollout(fn, _real_scenes(n), base_seed=500, batch=batch, new_frames=1)
    e.close()
    return res


def _real_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)      # two ranks on ONE device: RCCL refuses that, gloo carries the gather
    res = _real_rollout(n, batch=2)
    if rank == 0:
        q.put({m: res[m] for m in MOD_ORDER})
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_with_the_real_engine_equal_the_unsharded_rollout():
    """sharded (P = 2, scene i -> rank i mod 2, per-rank batches of 2) == unsharded (P = 1, one scene at a
    time), token for token, with the real HIP engine as rollout_fn.  Per-scene seeds are keyed by scene id, so the result does
    not depend on P or on the batch composition."""
    n = 5
    ref = _real_rollout(n, batch=1)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_real_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for m in MOD_ORDER:
        assert got[m].shape == (n, 3, CONTENT_LEN[m])
        np.testing.assert_array_equal(got[m], ref[m], err_msg=m)


@pytest.mark.gpu
def test_bench_multi_rank_contract_dry_run_on_one_gpu():
    """The driver's N > 1 launch of bench.py (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), dry-run on the
    one GPU a test box has: 2 ranks share cuda:0 over gloo (--share-gpu), 2 scenes per rank.  Checks the contract of the JSON line
    (rank 0 only, n_gpus, whole-job value = all ranks' scenes / max-over-ranks time, scaling 'weak') and that the sharded rollout +
    gather inside the timed region runs end to end."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--config", "tiny", "--history", "3",
           "--share-gpu", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["scenes_per_gpu"] == 2 and "dry_run" in d
    assert abs(d["value"] - 4 * 2207 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]      # 2 ranks x 2 scenes x 1 frame over the max-over-ranks time


@pytest.mark.gpu
def test_bench_force_dist_runs_the_rccl_calls_on_one_gpu():
    """SURVEY.md section 8e on hardware, as far as a one-GPU box allows: `bench.py --force-dist` creates the RCCL
    process group (backend nccl, device_id = cuda:0, world size 1) and runs the N > 1 path's barrier, the DEVICE all-gather of the int32
    token buffer and the max-over-ranks all-reduce of shard.py / bench.py inside the timed region."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--batch", "2", "--config", "tiny",
           "--history", "3", "--force-dist", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_exercised"]["backend"] == "nccl" and d["rccl_exercised"]["collective_device"] == "cuda"
    assert abs(d["value"] - 2 * 2207 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
