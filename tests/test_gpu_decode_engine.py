"""-m gpu: the XCD-resident decode engine (umgen_amd/csrc/oar_engine.hip, one launch per decode step) against the five-launch
decode layer (gemv.hip) it replaces and against the CPU oracle.

Both forms have the same rounding points (bf16 weights, bf16 K/V cache, fp32 everything else); they differ only in fp32
summation order (row partition over lanes is the same, the attention key partition is not), so teacher-forced logits agree
to ~1e-4 and the bar written here is the north-star's 1e-3.
"""
import os

import numpy as np
import pytest

from umgen_amd.config import MOD_ORDER, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_scene
from umgen_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu


def make(cfg, sd, engine_on, max_batch=1, graphs=True):
    """engine_on: the XCD-resident engine for EVERY batch size (its systolic / rounds schedules included: the batched decode layer that
    takes 24 and more scenes by default is switched off), else the five-launch layer."""
    old = {k: os.environ.get(k) for k in ("UMGEN_DECODE_ENGINE", "UMGEN_DECODE_BATCHED")}
    os.environ["UMGEN_DECODE_ENGINE"] = "1" if engine_on else "0"
    os.environ["UMGEN_DECODE_BATCHED"] = "0"
    try:
        e = Engine(cfg, precision="bf16", max_batch=max_batch, max_cond_frames=4, use_graphs=graphs)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    e.load_state_dict(sd)
    e.finalize()
    return e


@pytest.fixture(scope="module")
def setup():
    cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=5, rule_constrain=False)     # 5 layers: uneven over 2 / 4 / 8 groups
    sd = synthetic_state_dict(cfg, seed=21)
    return cfg, sd


def test_engine_logits_match_the_five_launch_path_under_teacher_forcing(setup):
    cfg, sd = setup
    scene = synthetic_scene(31, n_frames=2)
    window = {m: scene[m][0] for m in MOD_ORDER}
    ref = make(cfg, sd, engine_on=False)
    toks_ref, tr_ref = ref.frame(window, frame_idx=0, seed=3, trace=True)
    ref.close()
    eng = make(cfg, sd, engine_on=True)
    toks, tr = eng.frame(window, frame_idx=0, seed=3, trace=True, forced=toks_ref)
    worst = 0.0
    for m in ("map", "bbox3d", "image"):
        d = np.abs(tr[f"logits_{m}"] - tr_ref[f"logits_{m}"]).max()
        worst = max(worst, float(d))
        np.testing.assert_allclose(tr[f"logits_{m}"], tr_ref[f"logits_{m}"], atol=1e-3, rtol=0, err_msg=m)
    print(f"engine vs launches: max |dlogit| = {worst:.2e}, sampled != forced: {tr['counters']['sampled_ne_forced']}")
    assert tr["counters"]["sampled_ne_forced"] <= 4, tr["counters"]
    # graph replay == eager launches of the same engine, token for token
    out_g = eng.rollout(scene, 1, cond_frames=3, input_cond_frames=2, seeds=[5])
    eng.close()
    eager = make(cfg, sd, engine_on=True, graphs=False)
    out_e = eager.rollout(scene, 1, cond_frames=3, input_cond_frames=2, seeds=[5])
    eager.close()
    for m in MOD_ORDER:
        np.testing.assert_array_equal(out_g[m], out_e[m], err_msg=m)


@pytest.mark.parametrize("B,engine_on", [(2, True), (3, True), (4, True), (8, True), (8, False), (7, True), (7, False), (10, True),
                                         (12, True), (16, True), (32, True), (40, True)])
def test_engine_is_batch_invariant(setup, B, engine_on):
    """Scenes never interact: a batch of B scenes (B = 2: 4 XCDs per scene, 3 / 4: disjoint group sets, 5 .. 32: the systolic schedule
    -- 8 scenes: zero slack, 12 / 16 / 32: the throughput-bound forms the bench quotes, shared tail layers split over two groups --,
    40: more scenes than one systolic launch carries, rounds of whole-scene groups) gives exactly the B one-scene results, whatever
    group runs which layer."""
    cfg, sd = setup
    scenes = [synthetic_scene(40 + i, n_frames=2) for i in range(B)]
    seeds = [100 + i for i in range(B)]
    e = make(cfg, sd, engine_on=engine_on, max_batch=B)
    single = [e.rollout(scenes[i], 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[i]]) for i in range(B)]
    both = e.rollout({m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}, 1, cond_frames=3, input_cond_frames=2, seeds=seeds)
    e.close()
    bad = []
    for i in range(B):
        for m in MOD_ORDER:
            d = np.argwhere(both[m][i:i + 1] != single[i][m])
            if len(d):
                bad.append((i, m, len(d), d[:3].tolist()))
    assert not bad, bad


# ---------------------------------------------------------------------------------------------------------------------
# the batched decode layer (csrc/decode_batched.hip): 24 and more scenes per call, the scenes as the MFMA's B-columns
# ---------------------------------------------------------------------------------------------------------------------
def make_batched(cfg, sd, threshold, max_batch=1, precision="bf16"):
    """UMGEN_DECODE_BATCHED=<threshold>: batches of at least that many scenes run the batched decode layer (1: every call does)."""
    old = {k: os.environ.get(k) for k in ("UMGEN_DECODE_BATCHED",)}
    os.environ["UMGEN_DECODE_BATCHED"] = str(threshold)
    try:
        e = Engine(cfg, precision=precision, max_batch=max_batch, max_cond_frames=4)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    e.load_state_dict(sd)
    e.finalize()
    return e


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_batched_decode_layer_logits_match_the_five_launch_path_under_teacher_forcing(setup, precision):
    """Same rounding points as the five-launch layer and the engine (16-bit weights and K/V cache, fp32 activations -- here as hi + lo
    16-bit pairs on the matrix cores, 2^-17 / 2^-22 relative), another fp32 summation order: teacher-forced logits within the
    north-star's 1e-3 (observed ~1e-4), at most a handful of sampled tokens on the other side of a near-tie."""
    cfg, sd = setup
    scene = synthetic_scene(31, n_frames=2)
    window = {m: scene[m][0] for m in MOD_ORDER}
    old = os.environ.get("UMGEN_DECODE_ENGINE")
    os.environ["UMGEN_DECODE_ENGINE"] = "0"
    try:
        ref = make_batched(cfg, sd, 0, precision=precision)          # 0: never batched -> five launches per layer
    finally:
        if old is None:
            del os.environ["UMGEN_DECODE_ENGINE"]
        else:
            os.environ["UMGEN_DECODE_ENGINE"] = old
    toks_ref, tr_ref = ref.frame(window, frame_idx=0, seed=3, trace=True)
    assert ref.timings()["decode_engine"] == 0 and ref.timings()["decode_batched"] == 0
    ref.close()
    e = make_batched(cfg, sd, 1, precision=precision)
    toks, tr = e.frame(window, frame_idx=0, seed=3, trace=True, forced=toks_ref)
    assert e.timings()["decode_batched"] == 1 and e.timings()["decode_engine"] == 0
    e.close()
    worst = 0.0
    for m in ("map", "bbox3d", "image"):
        worst = max(worst, float(np.abs(tr[f"logits_{m}"] - tr_ref[f"logits_{m}"]).max()))
        np.testing.assert_allclose(tr[f"logits_{m}"], tr_ref[f"logits_{m}"], atol=1e-3, rtol=0, err_msg=m)
    print(f"batched decode layer vs launches ({precision}): max |dlogit| = {worst:.2e}, sampled != forced: {tr['counters']['sampled_ne_forced']}")
    assert tr["counters"]["sampled_ne_forced"] <= 4, tr["counters"]


@pytest.mark.parametrize("B", [2, 17, 33, 64])
def test_batched_decode_layer_is_batch_invariant(setup, B):
    """A scene's outputs are a function of its own MFMA column and their summation order does not depend on how many columns are in
    flight: a batch of B scenes (1 .. 4 column blocks of 16, ragged last block) == the B one-scene runs of the same path, bit for bit."""
    cfg, sd = setup
    scenes = [synthetic_scene(40 + i, n_frames=2) for i in range(B)]
    seeds = [100 + i for i in range(B)]
    e = make_batched(cfg, sd, 1, max_batch=B)
    single = [e.rollout(scenes[i], 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[i]]) for i in range(min(B, 20))]
    both = e.rollout({m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}, 1, cond_frames=3, input_cond_frames=2, seeds=seeds)
    assert e.timings()["decode_batched"] == 1
    # the scenes beyond the first 20: one-scene runs of a few of them (every column block, incl. the ragged last one)
    extra = sorted(set(range(B)) & {21, 31, 32, 47, 48, 63})
    single_extra = {i: e.rollout(scenes[i], 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[i]]) for i in extra}
    e.close()
    bad = []
    for i, ref in list(enumerate(single)) + sorted(single_extra.items()):
        for m in MOD_ORDER:
            d = np.argwhere(both[m][i:i + 1] != ref[m])
            if len(d):
                bad.append((i, m, len(d), d[:3].tolist()))
    assert not bad, bad


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_decode_lanes_equal_the_single_stream_batched_layer(setup, precision):
    """Decode lanes (engine.hip DecLane): the batch cut into sub-batches that decode on their own streams, forked behind the TAR stacks and
    joined at the end of the frame.  Scenes do not interact in the decode loop, so 1, 3 (default at 33 scenes: lanes of at most 16), 4 and 8 lanes emit the same
    tokens bit for bit -- over two frames, so that the second frame starts from what the lanes of the first one wrote."""
    cfg, sd = setup
    B = 33
    scenes = [synthetic_scene(70 + i, n_frames=2) for i in range(B)]
    toks = {m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}
    seeds = [500 + i for i in range(B)]
    outs = {}
    for lanes in ("1", None, "4", "8"):
        old = os.environ.get("UMGEN_DECODE_LANES")
        if lanes is not None:
            os.environ["UMGEN_DECODE_LANES"] = lanes
        try:
            e = make_batched(cfg, sd, 1, max_batch=B, precision=precision)
        finally:
            if lanes is not None:
                if old is None:
                    del os.environ["UMGEN_DECODE_LANES"]
                else:
                    os.environ["UMGEN_DECODE_LANES"] = old
        outs[lanes] = e.rollout(toks, 2, cond_frames=3, input_cond_frames=2, seeds=seeds)
        t = e.timings()
        assert t["decode_batched"] == 1 and t["decode_lanes"] == {"1": 1, None: 3, "4": 4, "8": 8}[lanes], t
        e.close()
    for lanes in (None, "4", "8"):
        for m in MOD_ORDER:
            np.testing.assert_array_equal(outs[lanes][m], outs["1"][m], err_msg=f"{lanes} lanes, {m}")


def test_decode_lanes_in_control_mode_with_a_growing_window(setup):
    """The lanes address every per-scene array through their base offset -- the control mask, the previous frame's boxes, the decoded-box
    buffer of the rule constraint, the seeds: a 25-scene control rollout (pose + one controlled slot per scene, window growing 2 -> 3 so that
    the second frame reuses the slot caches) on 2 lanes equals the one-stream batched layer and the scenes' own one-scene rollouts."""
    from umgen_amd.synth import synthetic_control
    cfg, sd = setup
    B = 25
    scenes = [synthetic_scene(90 + i, n_frames=2) for i in range(B)]
    inits = [synthetic_control(90 + i, n_frames=2, slot=2 + i % 5) for i in range(B)]
    cat = lambda ds: {k: np.concatenate([d[k] for d in ds]) for k in ds[0]}
    seeds = [700 + i for i in range(B)]
    kw = dict(cond_frames=3, input_cond_frames=2, control_test=True)
    outs = {}
    for lanes in ("1", None):
        old = os.environ.get("UMGEN_DECODE_LANES")
        if lanes is not None:
            os.environ["UMGEN_DECODE_LANES"] = lanes
        try:
            e = make_batched(cfg, sd, 1, max_batch=B)
        finally:
            if lanes is not None:
                if old is None:
                    del os.environ["UMGEN_DECODE_LANES"]
                else:
                    os.environ["UMGEN_DECODE_LANES"] = old
        outs[lanes] = e.rollout(cat(scenes), 2, init_tokens=cat(inits), seeds=seeds, **kw)
        assert e.timings()["decode_lanes"] == (1 if lanes else 2)
        if lanes is None:
            single = {i: e.rollout(scenes[i], 2, init_tokens=inits[i], seeds=[seeds[i]], **kw) for i in (0, 12, 13, 24)}
        e.close()
    for m in MOD_ORDER:
        np.testing.assert_array_equal(outs[None][m], outs["1"][m], err_msg=m)
        for i, ref in single.items():
            np.testing.assert_array_equal(outs[None][m][i:i + 1], ref[m], err_msg=f"scene {i} {m}")


def test_default_path_selection_by_batch_size(setup):
    """Up to 23 scenes per call the XCD-resident engine takes the decode step (its systolic schedule from 5 on), from 24 on the batched
    layer (measured crossover, profiles/r04_lanes_sweep.txt; UMGEN_DECODE_BATCHED moves the threshold): umgen_timings says which ran.  (Round 5's
    multi-scene engine and chain launch measured behind both -- profiles/r05_ms_experiments.txt, r05_decode_chain.txt -- and left the library in round 6.)"""
    cfg, sd = setup
    e = Engine(cfg, precision="bf16", max_batch=24, max_cond_frames=4)
    e.load_state_dict(sd)
    e.finalize()
    scenes = [synthetic_scene(40 + i, n_frames=2) for i in range(24)]
    e.rollout({m: np.concatenate([s[m] for s in scenes[:23]]) for m in MOD_ORDER}, 1, cond_frames=3, input_cond_frames=2, seeds=list(range(23)))
    t = e.timings()
    assert t["decode_engine"] == 1 and t["decode_batched"] == 0
    e.rollout({m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}, 1, cond_frames=3, input_cond_frames=2, seeds=list(range(24)))
    t = e.timings()
    assert t["decode_engine"] == 0 and t["decode_batched"] == 1 and t["decode_lanes"] == 2
    e.close()


# ---------------------------------------------------------------------------------------------------------------------
# the overlapped pass on the decode engine's idle XCDs (csrc/bg_worker.h): engines for one scene per call
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("task", ["video", "control", "mapgiven"])
def test_background_workers_reproduce_the_foreground_engine(task, precision):
    """Production width, 10 BlockOAR layers / 2 blocks per stack (`deep`): a rollout whose later frames find slots 0 .. T - 2 of their window already pushed
    through the four stacks by the background workers -- recorded op list, virtual launches of the stand-alone kernels' bodies on the XCDs the one-scene
    engine leaves idle, resumed from decode step to decode step, the rest drained behind the frame -- equals, token for token, the rollout of an engine
    that computes every window in the foreground (UMGEN_BG_ENGINE=0).  Sliding window (video), growing window with control tokens, and a given map
    (whose prefix pass runs in the foreground in both engines, ahead of the workers' first op)."""
    from tests.golden.make_full_width_golden import config as width_config
    from umgen_amd.synth import synthetic_control, synthetic_given_map
    from umgen_amd.weights import synthetic_state_dict as ssd
    cfg = width_config("deep")
    sd = ssd(cfg, seed=5)
    T, frames = 4, 3
    scene = synthetic_scene(17, n_frames=T)
    kw = dict(cond_frames=T, input_cond_frames=T, seeds=[3])
    if task == "control":
        c = synthetic_control(17, n_frames=frames)
        kw.update(input_cond_frames=2, init_tokens={k: c[k] for k in ("pose", "bbox3d")}, control_test=True)
    elif task == "mapgiven":
        kw.update(init_tokens={"map": synthetic_given_map(17, n_frames=frames)["map"]})
    outs, tms = {}, {}
    for bg in ("0", "1"):
        old = os.environ.get("UMGEN_BG_ENGINE")
        os.environ["UMGEN_BG_ENGINE"] = bg
        try:
            e = Engine(cfg, precision=precision, max_batch=1, max_cond_frames=T)
        finally:
            if old is None:
                del os.environ["UMGEN_BG_ENGINE"]
            else:
                os.environ["UMGEN_BG_ENGINE"] = old
        e.load_state_dict(sd)
        e.finalize()
        outs[bg] = e.rollout(scene, frames, **kw)
        tms[bg] = e.timings()
        e.close()
    assert tms["1"]["decode_engine"] == 1 and tms["0"]["decode_engine"] == 1
    assert tms["1"]["overlapped_frames"] == frames - 1, tms["1"]
    if task == "mapgiven":
        assert tms["1"]["prefix_passes"] == frames and tms["0"]["prefix_passes"] == frames
    for m in MOD_ORDER:
        np.testing.assert_array_equal(outs["1"][m], outs["0"][m], err_msg=m)


def test_background_workers_with_a_long_window_and_eager_launches():
    """A window of 22 history slots (the temporal kernel's 32-slot form: all 512 threads of a worker workgroup) and eager launches instead of graph replays:
    same tokens as the foreground engine; and a frame through `umgen_frame` between two rollouts (no pass may be left pending, none is started)."""
    from tests.golden.make_full_width_golden import config as width_config
    from umgen_amd.weights import synthetic_state_dict as ssd
    cfg = width_config("deep", max_frame_len=32)
    sd = ssd(cfg, seed=6)
    T, frames = 22, 2
    scene = synthetic_scene(23, n_frames=T)
    outs = {}
    for bg, graphs in (("0", True), ("1", False)):
        old = os.environ.get("UMGEN_BG_ENGINE")
        os.environ["UMGEN_BG_ENGINE"] = bg
        try:
            e = Engine(cfg, precision="bf16", max_batch=1, max_cond_frames=T, use_graphs=graphs)
        finally:
            if old is None:
                del os.environ["UMGEN_BG_ENGINE"]
            else:
                os.environ["UMGEN_BG_ENGINE"] = old
        e.load_state_dict(sd)
        e.finalize()
        outs[bg] = e.rollout(scene, frames, cond_frames=T, input_cond_frames=T, seeds=[4])
        if bg == "1":
            assert e.timings()["overlapped_frames"] == frames - 1, e.timings()
            toks, _ = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, seed=4)
            again = e.rollout(scene, frames, cond_frames=T, input_cond_frames=T, seeds=[4])
            for m in MOD_ORDER:
                np.testing.assert_array_equal(toks[m], outs["1"][m][0, T], err_msg=f"umgen_frame {m}")
                np.testing.assert_array_equal(again[m], outs["1"][m], err_msg=f"second rollout {m}")
        e.close()
    for m in MOD_ORDER:
        np.testing.assert_array_equal(outs["1"][m], outs["0"][m], err_msg=m)
