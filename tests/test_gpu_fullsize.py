"""-m gpu: parity at the production width and at BASELINE.json's full size.

* full WIDTH (E=768, H=16, all vocabularies at their production size, S=2207) with one layer per stack: small enough
  for the CPU oracle to finish in seconds, so logits are compared directly under teacher forcing;
* full SIZE (UMGen_Large, 2.44 B parameters, 20 history frames): the oracle needs ~15 min per frame on CPU, so parity is
  carried by size-independent properties of the path: scenes never interact (a B=2 batch == the two B=1 rollouts), a
  hipGraph replay == eager launches, `umgen_frame` == the first frame of `umgen_rollout`, a frame teacher-forced with
  its own output samples exactly that output again, the history is returned untouched and every token is in range.
"""
import dataclasses
import os

import numpy as np
import pytest

from oracle.umgen_oracle import OracleUMGen
from umgen_amd.config import BBOX_PAD, CONTENT_LEN, MOD_ORDER, N_SLOTS, SLOT_LEN, large_config, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_control, synthetic_scene
from umgen_amd.weights import synthetic_items, synthetic_state_dict

pytestmark = pytest.mark.gpu


def full_width_cfg(**over):
    return tiny_config(n_embd=768, n_head=16, **over)


@pytest.fixture(scope="module")
def full_width():
    # rule_constrain off: a blanked slot leaves stale K/V rows behind in the unforced run (UMGen.py:1116-1123), which a
    # teacher-forced replay of the OUTPUT tokens cannot reproduce; the rule path has its own tests in test_gpu_parity.py
    cfg = full_width_cfg(rule_constrain=False).greedy()
    sd = synthetic_state_dict(cfg, seed=21)
    scene = synthetic_scene(31, n_frames=2)
    o = OracleUMGen(cfg, sd)
    ref = o.inference(1, 3, scene, input_cond_frames=2, trace=True, seed=0)
    return cfg, sd, scene, ref, o.trace


def test_full_width_fp32_teacher_forced_logits_vs_oracle(full_width):
    """Production width and vocabularies, fp32 parity mode: every OAR logit row of the frame within the north-star's 1e-3
    of the oracle under teacher forcing, and the engine samples the oracle's greedy tokens back."""
    cfg, sd, scene, ref, otr = full_width
    forced = {m: ref[m][0, 2] for m in MOD_ORDER}
    e = Engine(cfg, precision="fp32", max_cond_frames=4)
    e.load_state_dict(sd)
    e.finalize()
    toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, trace=True, forced=forced)
    np.testing.assert_allclose(tr["cond"], otr["cond"][0], atol=5e-4, rtol=0)
    np.testing.assert_allclose(tr["ego_logits"], otr["ego_logits"][0], atol=1e-3, rtol=0)
    for m in ("map", "bbox3d", "image"):
        np.testing.assert_allclose(tr[f"logits_{m}"], otr["logits"][0][m], atol=1e-3, rtol=0)
    # greedy arg-max of near-tied logits may legitimately differ by summation order; everything else must be identical
    assert tr["counters"]["sampled_ne_forced"] <= 2, tr["counters"]
    e.close()


def test_full_width_bf16_teacher_forced_logits_vs_oracle(full_width):
    """bf16 production mode at production width against the fp32 oracle on the same bf16-rounded weights (tolerances of
    tests/test_gpu_parity.py::test_bf16_teacher_forced_logits_vs_oracle: bf16 operands in the TAR GEMMs / attention and a
    bf16 KV cache)."""
    cfg, sd, scene, _, _ = full_width
    o = OracleUMGen(cfg, sd, weight_dtype="bf16")
    ref = o.inference(1, 3, scene, input_cond_frames=2, trace=True, seed=0)
    forced = {m: ref[m][0, 2] for m in MOD_ORDER}
    e = Engine(cfg, precision="bf16", max_cond_frames=4)
    e.load_state_dict(sd)
    e.finalize()
    toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, trace=True, forced=forced)
    np.testing.assert_allclose(tr["cond"], o.trace["cond"][0], atol=6e-2, rtol=0)
    for m in ("map", "bbox3d", "image"):
        np.testing.assert_allclose(tr[f"logits_{m}"], o.trace["logits"][0][m], atol=8e-2, rtol=0)
        agree = (tr[f"logits_{m}"].argmax(-1) == o.trace["logits"][0][m].argmax(-1)).mean()
        assert agree > 0.95, (m, agree)
    e.close()


# ---------------------------------------------------------------------------------------------------------------------
# UMGen_Large, configs[1] / configs[2] of BASELINE.json at full size
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def large():
    cfg = large_config()
    e = Engine(cfg, precision="bf16", max_batch=2, max_cond_frames=20)
    e.load_state_dict(synthetic_items(cfg, seed=0))
    e.finalize()
    yield cfg, e
    e.close()


def check_structure(cfg, scene, out, t_in, new_frames):
    vocab = {"pose": cfg.pose_vocab_size, "map": cfg.map_vocab_size, "bbox3d": cfg.bbox3d_vocab_size, "image": cfg.img_vocab_size}
    for m in MOD_ORDER:
        assert out[m].shape == (scene[m].shape[0], t_in + new_frames, CONTENT_LEN[m])
        np.testing.assert_array_equal(out[m][:, :t_in], scene[m][:, :t_in], err_msg=f"history {m}")   # UMGen.py:1581-1595
        assert out[m].min() >= 0 and out[m].max() < vocab[m], m
    # a slot blanked by the rule constraint is blanked as a whole (UMGen.py:1116-1123): category pad => 11 pads
    box = out["bbox3d"][:, t_in:].reshape(-1, N_SLOTS, SLOT_LEN)
    blank = box[..., 10] == BBOX_PAD
    assert np.all(box[blank] == BBOX_PAD)


def test_large_batch_of_two_equals_two_single_rollouts(large):
    """configs[1] shape (video, 20 history frames) with the default k = 5/5/16 sampler: scenes are independent units, so the
    B=2 batch must reproduce the two B=1 rollouts token for token (per-scene counter RNG; batch-invariant kernels)."""
    cfg, e = large
    scenes = [synthetic_scene(1000 + i, n_frames=20) for i in range(2)]
    seeds = [7, 8]
    single = [e.rollout(scenes[i], 2, cond_frames=20, seeds=[seeds[i]]) for i in range(2)]
    # frame 2 of a one-scene rollout takes the overlapped path: history slots 0..18 went through the stacks beside frame 1's decode
    assert e.timings()["overlapped_frames"] == 1
    both_in = {m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}
    both = e.rollout(both_in, 2, cond_frames=20, seeds=seeds)
    check_structure(cfg, both_in, both, 20, 2)
    for i in range(2):
        for m in MOD_ORDER:
            np.testing.assert_array_equal(both[m][i:i + 1], single[i][m], err_msg=f"scene {i} {m}")
    # determinism: the same call again gives the same tokens
    again = e.rollout(scenes[0], 2, cond_frames=20, seeds=[seeds[0]])
    for m in MOD_ORDER:
        np.testing.assert_array_equal(again[m], single[0][m])
    # and a different seed does not
    other = e.rollout(scenes[0], 1, cond_frames=20, seeds=[seeds[0] + 1])
    assert any(np.any(other[m][:, 20] != single[0][m][:, 20]) for m in ("map", "image"))


def test_large_frame_entry_point_and_self_forcing(large):
    """`umgen_frame` on the 20-frame window == frame 20 of `umgen_rollout`; teacher-forcing a frame with its own output
    samples exactly that output again (counter 5 = sampled != forced stays 0)."""
    cfg, e = large
    scene = synthetic_scene(1003, n_frames=20)
    out = e.rollout(scene, 1, cond_frames=20, seeds=[5])
    window = {m: scene[m][0] for m in MOD_ORDER}
    toks, _ = e.frame(window, frame_idx=0, seed=5)
    for m in MOD_ORDER:
        np.testing.assert_array_equal(toks[m], out[m][0, 20], err_msg=m)
    # self-forcing needs the rule constraint off: a blanked slot's pad tokens were never fed back in the run that produced
    # them (stale K/V rows, UMGen.py:1116-1123), so replaying the output tokens is a different computation after a blanking
    norule = dataclasses.replace(cfg, rule_constrain=False)
    toks1, _ = e.frame(window, frame_idx=0, seed=5, sampling=norule)
    toks2, tr = e.frame(window, frame_idx=0, seed=5, sampling=norule, forced=toks1)
    assert tr["counters"]["sampled_ne_forced"] == 0, tr["counters"]
    for m in MOD_ORDER:
        np.testing.assert_array_equal(toks2[m], toks1[m], err_msg=m)


def test_large_control_rollout_copies_control_pose_and_overlapped_graph_path_equals_plain_eager(large):
    """configs[2] shape (control, 13 history frames, one controlled agent): control pose tokens are copied verbatim into the
    output (UMGen.py:1640-1651); the production path (hipGraph replay of the decode step, next frame's history slots pushed
    through the stacks on the background stream) gives the same tokens as the plain path (eager launches, one pass)."""
    cfg, e = large
    scene = synthetic_scene(1005, n_frames=13)
    init = synthetic_control(1005, n_frames=2)
    out = e.rollout(scene, 2, cond_frames=20, input_cond_frames=13, init_tokens=init, control_test=True, seeds=[9])
    check_structure(cfg, scene, out, 13, 2)
    np.testing.assert_array_equal(out["pose"][:, 13:15], init["pose"][:, :2])
    assert e.timings()["overlapped_frames"] == 1       # growing window (13 -> 14 slots), pose given: ego prefix skipped
    os.environ["UMGEN_OVERLAP"] = "0"                  # plain path: whole window in one foreground pass, eager launches
    try:
        eager = Engine(cfg, precision="bf16", max_batch=1, max_cond_frames=20, use_graphs=False)
    finally:
        del os.environ["UMGEN_OVERLAP"]
    eager.load_state_dict(synthetic_items(cfg, seed=0))
    eager.finalize()
    out2 = eager.rollout(scene, 2, cond_frames=20, input_cond_frames=13, init_tokens=init, control_test=True, seeds=[9])
    assert eager.timings()["overlapped_frames"] == 0
    eager.close()
    for m in MOD_ORDER:
        np.testing.assert_array_equal(out2[m], out[m], err_msg=m)
