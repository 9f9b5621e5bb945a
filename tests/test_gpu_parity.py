"""-m gpu: the HIP engine (through the C ABI) against the CPU oracle and the golden vectors recorded from the reference.

Bars (BASELINE.json north_star): token sequences bit-exact under greedy decoding; logits within the tolerance
written in each test under teacher forcing.
"""
import os

import numpy as np
import pytest
import torch

from umgen_amd.config import MOD_ORDER, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import golden_init_tokens, synthetic_control, synthetic_scene
from umgen_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
LOGIT_POS = {"map": [0, 1, 511, 1023], "bbox3d": [0, 9, 10, 11, 330, 659], "image": [0, 255, 511]}
COND_ROWS = [0, 1, 4, 5, 6, 500, 1030, 1031, 1032, 1042, 1692, 1693, 1694, 2000, 2206]


@pytest.fixture(scope="module")
def oc():
    """Oracle outputs of the tiny-config cases, recorded by tests/golden/make_oracle_cases.py (the oracle is not re-run on the GPU
    box; tests/test_oracle.py keeps the vectors current)."""
    return np.load(os.path.join(GOLD, "oracle_cases.npz"))


def make_engine(cfg, seed, precision, max_batch=1):
    e = Engine(cfg, precision=precision, max_batch=max_batch, max_cond_frames=4)
    e.load_state_dict(synthetic_state_dict(cfg, seed=seed))
    e.finalize()
    return e


@pytest.mark.parametrize("name", ["tiny_video_greedy", "tiny_control_greedy", "tiny_boxctl_greedy", "tiny_mapgiven_greedy", "tiny_mapboxgiven_greedy"])
def test_fp32_greedy_rollout_is_token_exact_vs_reference_golden(name):
    """fp32 parity mode: the whole rollout (ego net, 3 TAR stacks, 2206-step OAR loop, rule constraint, control with pose +
    bbox3d tokens and with bbox3d tokens alone, the map -- or the map and the boxes -- of every new frame GIVEN as init_tokens -- infer_oar_net's predefined-token
    prefix, UMGen.py:1184-1201) reproduces the token sequences recorded from the reference itself, bit for bit."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    cfg = tiny_config().greedy()
    e = make_engine(cfg, ws, "fp32")
    scene = synthetic_scene(sid, n_frames=icf)
    init = golden_init_tokens(sid, nf, ctl)
    out = e.rollout(scene, nf, cond_frames=cf, input_cond_frames=icf, init_tokens=init, control_test=ctl in (1, 2), seeds=[0])
    for m in MOD_ORDER:
        np.testing.assert_array_equal(out[m], g[f"out_{m}"].astype(np.int64), err_msg=m)
    e.close()


@pytest.mark.parametrize("name", ["tiny_mapgiven_greedy", "tiny_mapboxgiven_greedy"])
def test_given_token_prefix_as_one_pass_is_token_exact(name, monkeypatch):
    """The predefined-token prefix of infer_oar_net (UMGen.py:1184-1201, 1234-1237: the reference pushes the given map -- or map and boxes --
    through the 36 layers in ONE forward pass) as one pass here too (engine.hip run_prefix_prefill: the given positions as the rows of the
    stacks' GEMMs + S x S attention with the causal mask, every layer's k | v rows into the decode cache; the step loop starts at the last given
    position): the rollouts recorded from the reference bit for bit in fp32 mode, on EVERY new frame (UMGEN_OVERLAP=0: no background TAR pass in
    the stacks' buffers), equal to the step-by-step replay of rounds 1-4 (UMGEN_PREFIX_PASS=0); in bf16 the two forms agree up to near-ties."""
    monkeypatch.setenv("UMGEN_OVERLAP", "0")
    g = np.load(os.path.join(GOLD, name + ".npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    cfg = tiny_config().greedy()
    scene = synthetic_scene(sid, n_frames=icf)
    init = golden_init_tokens(sid, nf, ctl)
    outs = {}
    for precision in ("fp32", "bf16"):
        for passes in (True, False):
            if passes:
                monkeypatch.delenv("UMGEN_PREFIX_PASS", raising=False)
            else:
                monkeypatch.setenv("UMGEN_PREFIX_PASS", "0")
            e = make_engine(cfg, ws, precision)
            outs[precision, passes] = e.rollout(scene, nf, cond_frames=cf, input_cond_frames=icf, init_tokens=init, seeds=[0])
            assert e.timings()["prefix_passes"] == (nf if passes else 0)
            e.close()
    for m in MOD_ORDER:
        np.testing.assert_array_equal(outs["fp32", True][m], g[f"out_{m}"].astype(np.int64), err_msg=m)
        np.testing.assert_array_equal(outs["fp32", False][m], g[f"out_{m}"].astype(np.int64), err_msg=m)
    agree = np.mean([np.mean(outs["bf16", True][m][:, icf:] == outs["bf16", False][m][:, icf:]) for m in ("bbox3d", "image")])
    print(f"{name}: bf16 greedy token agreement of the prefix pass with the step-by-step replay {agree:.4f}")
    assert agree > 0.97, agree


@pytest.mark.parametrize("name", ["tiny_control_greedy", "tiny_boxctl_greedy", "tiny_grow_control_greedy", "tiny_grow_boxctl_greedy"])
@pytest.mark.parametrize("precision", ["fp32"])
def test_growing_window_slot_reuse_in_the_foreground_is_token_exact(name, precision, monkeypatch):
    """SURVEY.md section 8 row f-3, the production form: while the control-mode window still grows (2 -> 6 history frames here, 13 -> 20
    in configs[2]; infer_fun.py:64-71, UMGen.py:1600-1603) a frame leaves the temporal k | v rows of all its slots in the per-layer slot
    caches and the next frame pushes only its new last slot through the ego / map / box / TAR stacks -- on the ONE stream the decode
    engine uses (UMGEN_OVERLAP=0: no masked background stream).  The rollouts recorded from the reference itself (pose + bbox3d control
    and bbox3d-only control, where the ego stack is cached too) must come out bit for bit, and the reuse must really have happened."""
    monkeypatch.setenv("UMGEN_OVERLAP", "0")
    g = np.load(os.path.join(GOLD, name + ".npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    cfg = tiny_config().greedy()
    e = Engine(cfg, precision=precision, max_batch=1, max_cond_frames=cf)
    e.load_state_dict(synthetic_state_dict(cfg, seed=ws))
    e.finalize()
    scene = synthetic_scene(sid, n_frames=icf)
    init = golden_init_tokens(sid, nf, ctl)
    out = e.rollout(scene, nf, cond_frames=cf, input_cond_frames=icf, init_tokens=init, control_test=bool(ctl), seeds=[0])
    reused = e.timings()["overlapped_frames"]
    for m in MOD_ORDER:
        np.testing.assert_array_equal(out[m], g[f"out_{m}"].astype(np.int64), err_msg=m)
    assert reused == min(nf - 1, cf - icf), f"{reused} frames reused the slot caches, expected {min(nf - 1, cf - icf)}"
    # the same rollout with the reuse switched off (every frame recomputes its whole window, like the reference) is the same tokens
    monkeypatch.setenv("UMGEN_GROW_CACHE", "0")
    e2 = Engine(cfg, precision=precision, max_batch=1, max_cond_frames=cf)
    e2.load_state_dict(synthetic_state_dict(cfg, seed=ws))
    e2.finalize()
    out2 = e2.rollout(scene, nf, cond_frames=cf, input_cond_frames=icf, init_tokens=init, control_test=bool(ctl), seeds=[0])
    assert e2.timings()["overlapped_frames"] == 0
    for m in MOD_ORDER:
        np.testing.assert_array_equal(out2[m], out[m], err_msg=m)
    e.close()
    e2.close()


def test_fp32_first_frame_activations_match_reference_golden():
    g = np.load(os.path.join(GOLD, "tiny_video_greedy.npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    cfg = tiny_config().greedy()
    e = make_engine(cfg, ws, "fp32")
    scene = synthetic_scene(sid, n_frames=icf)
    toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, trace=True)
    np.testing.assert_allclose(tr["cond"][COND_ROWS], g["cond_rows"][0], atol=2e-4, rtol=0)
    np.testing.assert_allclose(tr["ego_logits"], g["ego_logits"][0], atol=2e-4, rtol=0)
    for m, pos in LOGIT_POS.items():
        np.testing.assert_allclose(tr[f"logits_{m}"][pos], g[f"logits_{m}"], atol=1e-3, rtol=0)   # north-star: 1e-3 logit tolerance
    for m in MOD_ORDER:
        np.testing.assert_array_equal(toks[m], g[f"out_{m}"][0, icf].astype(np.int64))
    e.close()


def check_inside_ensemble(ens, tr, cond_rows, logit_pos, what):
    """The bound of the 16-bit modes (tests/golden/make_ensemble.py): the engine is one more summation order of the rounding-aware
    oracle, so it must lie within 2 x the spread of the oracle's own accumulation-order ensemble around the ensemble centre, and
    an arg-max may only differ where the centre's top-2 gap is inside that noise (4 x spread: both candidates move by up to 2 x)."""
    worst = {}
    for q, got, ref in [("cond", tr["cond"][cond_rows], ens["cond_center"]), ("ego", tr["ego_logits"], ens["ego_center"])] + \
                       [(m, tr[f"logits_{m}"][pos], ens[f"{m}_center"]) for m, pos in logit_pos.items()]:
        spread = float(ens[f"{q}_spread"])
        d = float(np.abs(got - ref).max())
        worst[q] = (d, spread, d / spread)
        assert d <= 2.0 * spread, f"{what} {q}: max |engine - ensemble centre| = {d:.3e} > 2 x ensemble spread {spread:.3e}"
    flips = {}
    for m in logit_pos:
        am = tr[f"logits_{m}"].argmax(-1)
        f = np.nonzero(am != ens[f"{m}_argmax"].astype(np.int64))[0]
        gap = float(ens[f"{m}_gap"][f].max()) if len(f) else 0.0
        flips[m] = (len(f), gap)
        assert gap <= 4.0 * float(ens[f"{m}_spread"]), f"{what} {m}: arg-max flip at a position whose top-2 gap {gap:.3e} is outside the noise"
    print(f"{what}: (max dev, ensemble spread, ratio) {worst}; arg-max flips (count, largest centre gap) {flips}")
    return worst


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_16bit_teacher_forced_frame_lies_inside_the_oracle_ensemble(precision, oc):
    """bf16 (the bench mode) and fp16 (the reference's own autocast dtype, UMGen.py:1604-1605) under teacher forcing against the
    ACCUMULATION-ORDER ENSEMBLE of the rounding-aware oracle (weight_dtype="<prec>_engine": a 16-bit round trip wherever the
    engine stores 16 bits -- LN outputs, q|k|v, the online-softmax probabilities of the spatial attention, attention / GELU
    outputs in the TAR stacks, the K/V cache).  The bars are not fitted to the engine: they are twice the spread the oracle
    shows against ITSELF when its fp32 sums run in 8 other orders.  fp16 additionally meets an absolute 2e-3 on the logits."""
    g = np.load(os.path.join(GOLD, "tiny_video_greedy.npz"))
    ens = np.load(os.path.join(GOLD, f"ensemble_tiny_{precision}_engine.npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    cfg = tiny_config().greedy()
    scene = synthetic_scene(sid, n_frames=icf)
    forced = {m: ens[f"tok_{m}"].astype(np.int64) for m in MOD_ORDER}
    e = make_engine(cfg, ws, precision)
    toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, trace=True, forced=forced)
    check_inside_ensemble(ens, tr, COND_ROWS, LOGIT_POS, f"tiny {precision}")
    if precision == "fp16":
        for m, pos in LOGIT_POS.items():
            np.testing.assert_allclose(tr[f"logits_{m}"][pos], ens[f"{m}_center"], atol=2e-3, rtol=0, err_msg=m)
    # free-running greedy rollout (bf16): token-exact up to the first near-tie of the recorded oracle run; report where
    if precision == "bf16":
        out = e.rollout(scene, 1, cond_frames=cf, input_cond_frames=icf, seeds=[0])
        for m in ("map", "bbox3d", "image"):
            d = np.nonzero(out[m][0, icf] != oc[f"bf16_free_{m}"].astype(np.int64))[0]
            if len(d):
                gap = float(oc[f"bf16_free_gap_{m}"][d[0]])
                print(f"bf16 greedy rollout: first divergence at {m}[{d[0]}], oracle top-2 gap {gap:.2e}")
                assert gap < 4.0 * float(ens[f"{m}_spread"]) + 1e-3, (m, d[0])
                break
    e.close()


def test_fp32_sampled_rollout_matches_oracle_and_is_batch_invariant(oc):
    """k = 5/5/16 sampling with the build's counter-based RNG: engine == oracle token for token (fp32 mode), and a
    B=2 batch gives exactly the two B=1 results (scenes never interact; RNG keyed by scene seed)."""
    cfg = tiny_config()
    scenes = [synthetic_scene(10 + i, n_frames=2) for i in range(2)]
    seeds = [111, 222]
    e = make_engine(cfg, 3, "fp32", max_batch=2)
    single = [e.rollout(scenes[i], 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[i]]) for i in range(2)]
    both = e.rollout({m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}, 1, cond_frames=3, input_cond_frames=2, seeds=seeds)
    for i in range(2):
        for m in MOD_ORDER:
            np.testing.assert_array_equal(single[i][m], oc[f"sampled_{i}_{m}"].astype(np.int64), err_msg=f"scene {i} {m}")
            np.testing.assert_array_equal(both[m][i:i + 1], single[i][m], err_msg=f"batch scene {i} {m}")
    e.close()


def test_dropin_model_class_through_registry_matches_golden():
    """The reference's call sequence (evaluate.py:193-214, model_pl.py:237): build_from_cfg(dict(type=UMGen, config=Namespace))
    -> load_state_dict(strict=False) -> inference(**setting) returns the golden token dict."""
    from argparse import Namespace

    from umgen_amd.model import UMGen
    from umgen_amd.registry import MODELS, build_from_cfg

    g = np.load(os.path.join(GOLD, "tiny_video_greedy.npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    c = tiny_config().greedy()
    ns = Namespace(n_embd=c.n_embd, n_head=c.n_head, n_ego_tar_layer=1, n_ego_ca_layer=1, n_map_tar_layer=1, n_box_tar_layer=1,
                   n_tar_layer=1, n_oar_layer=c.n_oar_layer, pose_vocab_size=1024, map_vocab_size=8192, bbox3d_vocab_size=1028,
                   img_vocab_size=8192, aux_vocab_size=8, n_map_embd=16, n_img_embd=16, max_frame_len=c.max_frame_len, task_num=7,
                   task_name_id={"pose_map_bbox3d_image": 6}, sample_method="topk", top_k=1, top_k_map=1, p=0.4, sfmx_temp=1.0,
                   rule_constrain=True, split_map_tar=True, split_box_tar=True, map_transform=True, box_transform=False, n_step=1,
                   sample_img=True, bias=False, merage_ar_tar=True, only_ar=False)
    model = build_from_cfg(dict(type=UMGen, config=ns, precision="fp32"), MODELS)
    model.rcfg.topk_image = 1            # the reference hard-codes topk_image on the instance (UMGen.py:103)
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(c, seed=ws).items()}
    sd["transformer.head_tar_pose.weight"] = torch.zeros(1024, c.n_embd)     # present in real checkpoints, unused by the rollout
    res = model.load_state_dict(sd, strict=False)
    assert not res.missing_keys and res.unexpected_keys == ["transformer.head_tar_pose.weight"]
    model.eval()
    scene = synthetic_scene(sid, n_frames=icf)
    out = model.inference(new_frames=nf, cond_frames=cf, pred_task="pose_map_bbox3d_image",
                          input_cond_tokens={k: torch.from_numpy(v) for k, v in scene.items()}, init_tokens=None,
                          input_cond_frames=icf, control_test=False, cond_on_par=True, infer_from_gt=False)
    for m in MOD_ORDER:
        assert out[m].dtype == np.int64
        np.testing.assert_array_equal(out[m], g[f"out_{m}"].astype(np.int64), err_msg=m)


def test_fp32_top_p_frame_matches_oracle_under_teacher_forcing(oc):
    """sample_method='topp' (UMGen.py:915-965; not the evaluate.py default): nucleus p=0.4 for pose/bbox3d/map and the
    whole distribution for image tokens (UMGen.py:1133).  With random-init weights the nucleus holds thousands of nearly
    equiprobable codes, so a 1-ulp difference in exp() moves a draw across a CDF boundary: round 1 allowed 22 such mismatches
    between numpy's and the device's expf.  Both sides now use the same bit-reproducible exp (common.h exp_det / the oracle's
    exp_det: separately rounded fp32 Horner steps), so the remaining budget only covers arg-max-level logit noise: the
    comparison is teacher-forced (no error propagation) and the logits must agree to 1e-3."""
    cfg = tiny_config()
    cfg.sample_method = "topp"
    cfg.rule_constrain = False      # no retro-active blanking: the emitted tokens ARE the sampled stream being forced
    scene = synthetic_scene(21, n_frames=2)
    forced = {m: oc[f"topp_{m}"].astype(np.int64) for m in MOD_ORDER}
    e = make_engine(cfg, 5, "fp32")
    toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, seed=77, trace=True, forced=forced, sampling=cfg)
    for m, pos in LOGIT_POS.items():
        np.testing.assert_allclose(tr[f"logits_{m}"][pos], oc[f"topp_logits_{m}"], atol=1e-3, rtol=0)
    # the logits themselves still differ by fp32 summation order (<= 1e-3), which can move a CDF boundary across u: a handful
    print("top-p sampled != forced:", tr["counters"]["sampled_ne_forced"])
    assert tr["counters"]["sampled_ne_forced"] <= 8, tr["counters"]
    e.close()


def test_fp32_sampled_frame_exercises_pad_avoid_and_matches_oracle_counters(oc):
    """k = 5 sampling: the pad-avoid resample (UMGen.py:1092-1104) and the rule constraint fire on the device exactly as
    often as in the oracle."""
    cfg = tiny_config()
    scene = synthetic_scene(10, n_frames=2)
    e = make_engine(cfg, 3, "fp32")
    toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, seed=5, trace=True)     # seed chosen so that the pad-avoid branch fires
    for m in MOD_ORDER:
        np.testing.assert_array_equal(toks[m], oc[f"padavoid_{m}"].astype(np.int64), err_msg=m)
    c = tr["counters"]
    pad_avoid, rule_checked, rule_blanked = [int(x) for x in oc["padavoid_counters"]]
    assert c["pad_avoid"] == pad_avoid and c["pad_avoid"] > 0, (c, oc["padavoid_counters"])
    assert c["rule_checked"] == rule_checked
    assert c["rule_blanked"] == rule_blanked
    e.close()


def test_edge_cases_single_history_frame_zero_new_frames_and_errors(oc):
    """Ragged / degenerate calls: T_in = 1 history frame (window grows 1 -> 2), new_frames = 0 (history returned unchanged),
    B = 3 with max_batch = 3, and loud failures on invalid arguments (no silent fallback)."""
    from umgen_amd.engine import UMGenError
    cfg = tiny_config().greedy()
    e = make_engine(cfg, 9, "fp32", max_batch=3)
    scene = synthetic_scene(30, n_frames=1)
    ref = {m: oc[f"edge_{m}"].astype(np.int64) for m in MOD_ORDER}              # oracle, window: 1 -> 2 -> slides
    out = e.rollout(scene, 2, cond_frames=2, input_cond_frames=1, seeds=[0])
    for m in MOD_ORDER:
        np.testing.assert_array_equal(out[m], ref[m], err_msg=m)
    same = e.rollout(scene, 0, cond_frames=2, input_cond_frames=1, seeds=[0])
    for m in MOD_ORDER:
        np.testing.assert_array_equal(same[m], scene[m])
    three = {m: np.concatenate([synthetic_scene(30 + i, n_frames=1)[m] for i in range(3)]) for m in MOD_ORDER}
    out3 = e.rollout(three, 1, cond_frames=2, input_cond_frames=1, seeds=[0, 0, 0])
    np.testing.assert_array_equal(out3["map"][0], out["map"][0, :2])
    with pytest.raises(UMGenError):
        e.rollout({m: np.concatenate([three[m], three[m]]) for m in MOD_ORDER}, 1, cond_frames=2, input_cond_frames=1)   # B = 6 > max_batch
    with pytest.raises(UMGenError):
        e.rollout(scene, 1, cond_frames=99, input_cond_frames=1)                                                       # window > max_cond_frames
    # malformed scenes fail at the ABI boundary instead of indexing the embedding tables out of bounds
    bad = {m: scene[m].copy() for m in MOD_ORDER}
    bad["map"][0, 0, 17] = 8192
    with pytest.raises(UMGenError, match="map token 8192"):
        e.rollout(bad, 1, cond_frames=2, input_cond_frames=1)
    bad = {m: scene[m].copy() for m in MOD_ORDER}
    bad["bbox3d"][0, 0, 5] = -1
    with pytest.raises(UMGenError, match="bbox3d token -1"):
        e.rollout(bad, 1, cond_frames=2, input_cond_frames=1)
    ctl = synthetic_control(30, n_frames=1)
    with pytest.raises(UMGenError, match="without control_test"):
        e.rollout(scene, 1, cond_frames=2, input_cond_frames=1, init_tokens={"bbox3d": ctl["bbox3d"]}, control_test=False)     # bbox3d tokens need control_test
    with pytest.raises(UMGenError, match="not supported"):       # image tokens are dropped by the reference's own decode loop (UMGen.py:1512-1520)
        e.rollout(scene, 1, cond_frames=2, input_cond_frames=1, init_tokens={"pose": ctl["pose"], "image": scene["image"]})
    with pytest.raises(UMGenError, match="given map token 8192"):
        e.rollout(scene, 1, cond_frames=2, input_cond_frames=1, init_tokens={"map": np.full((1, 1, 1024), 8192)})
    with pytest.raises(UMGenError, match="shape"):
        e.rollout(scene, 1, cond_frames=2, input_cond_frames=1, init_tokens={"pose": ctl["pose"], "bbox3d": ctl["bbox3d"][:, :, :600]})
    ctl_bad = {"pose": ctl["pose"].copy(), "bbox3d": ctl["bbox3d"].copy()}
    ctl_bad["pose"][0, 0, 1] = 1024
    with pytest.raises(UMGenError, match="control pose token 1024"):
        e.rollout(scene, 1, cond_frames=2, input_cond_frames=1, init_tokens=ctl_bad, control_test=True)
    out_again = e.rollout(scene, 2, cond_frames=2, input_cond_frames=1, seeds=[0])     # the engine is still usable after the failures
    for m in MOD_ORDER:
        np.testing.assert_array_equal(out_again[m], ref[m], err_msg=m)
    e2 = Engine(cfg, precision="fp32", max_batch=1, max_cond_frames=2)
    with pytest.raises(UMGenError):
        e2.finalize()                                                                                                   # weights missing
    with pytest.raises(UMGenError):
        e2.load_tensor("transformer.spe.weight", np.zeros((7, 7), np.float32))                                          # wrong shape
    e2.close()
    e.close()


def test_evaluate_cli_end_to_end_on_a_raw_clip(tmp_path):
    """`python -m umgen_amd.evaluate` (counterpart of evaluate.py + model_pl.test_step): raw tokenized_origin_scenes-schema clip ->
    scene_io reader -> rollout -> `<output>/saved_token/<name>_tokens.pkl` (model_pl.py:350-355); a second run skips the scene
    (model_pl.py:215-216)."""
    import pickle

    from umgen_amd import evaluate, scene_io

    data = tmp_path / "scenes"
    data.mkdir()
    with open(data / "clip_0007.pkl", "wb") as f:
        pickle.dump(scene_io.synthetic_raw_scene(7, 120, 70), f)
    out = tmp_path / "out"
    argv = ["--infer_task", "video", "--set_num_new_frames", "1", "--model_scale", "debug", "--debug", "1",
            "--data_test_root", str(data), "--output_path", str(out), "--precision", "fp32"]
    evaluate.main(argv)
    p = out / "saved_token" / "clip_0007_tokens.pkl"
    assert p.exists()
    with open(p, "rb") as f:
        toks = pickle.load(f)
    ref = scene_io.scene_tokens(scene_io.synthetic_raw_scene(7, 120, 70), block_size=21)
    for m in MOD_ORDER:   # 20 conditioning frames (video task) + 1 new one; the tiny model's window is the last 7 of them
        assert toks[m].dtype == np.int64 and toks[m].shape[:2] == (1, 21)
        np.testing.assert_array_equal(toks[m][0, :20], ref[m][:20], err_msg=m)     # history = what the reader produced
    mtime = p.stat().st_mtime_ns
    evaluate.main(argv)                                                             # "... has been processed"
    assert p.stat().st_mtime_ns == mtime


def test_evaluate_cli_batches_scenes_through_sharded_rollout(tmp_path):
    """`--synthetic 5 --batch 2`: the CLI rolls its scenes out through umgen_amd.shard.sharded_rollout in engine calls of 2 + 2 + 1
    scenes; the five pickles equal the five one-scene rollouts (per-scene seeds are keyed by scene id, the batch never enters the
    arithmetic), and a re-run with two pickles deleted regenerates exactly those two (skip-if-exists stays per scene)."""
    import pickle

    from umgen_amd import evaluate
    from umgen_amd.shard import scene_seed

    out = tmp_path / "out"
    argv = ["--infer_task", "video", "--set_num_new_frames", "1", "--model_scale", "debug", "--debug", "1", "--synthetic", "5",
            "--batch", "2", "--output_path", str(out), "--precision", "fp32", "--seed", "40"]
    evaluate.main(argv)
    cfg, new_frames, input_cond = evaluate.resolve(evaluate.build_parser().parse_args(argv))
    T_hist = min(20, cfg.max_frame_len - 1)
    e = make_engine_hist(cfg, T_hist)
    ref = []
    for i in range(5):
        sc = synthetic_scene(i, n_frames=min(input_cond, T_hist))
        ref.append(e.rollout(sc, new_frames, cond_frames=T_hist, input_cond_frames=sc["pose"].shape[1], seeds=[scene_seed(40, i)]))
    e.close()
    paths = [out / "saved_token" / f"synthetic_{i:04d}_tokens.pkl" for i in range(5)]
    for i, p in enumerate(paths):
        with open(p, "rb") as f:
            toks = pickle.load(f)
        for m in MOD_ORDER:
            assert toks[m].dtype == np.int64
            np.testing.assert_array_equal(toks[m], ref[i][m], err_msg=f"scene {i} {m}")
    keep = {p: p.stat().st_mtime_ns for p in paths}
    paths[1].unlink()
    paths[4].unlink()
    evaluate.main(argv)
    for i, p in enumerate(paths):
        if i in (1, 4):
            with open(p, "rb") as f:
                toks = pickle.load(f)
            for m in MOD_ORDER:
                np.testing.assert_array_equal(toks[m], ref[i][m], err_msg=f"regenerated scene {i} {m}")
        else:
            assert p.stat().st_mtime_ns == keep[p]


def make_engine_hist(cfg, T_hist):
    from umgen_amd.weights import expected_keys, synth_tensor
    e = Engine(cfg, precision="fp32", max_batch=1, max_cond_frames=T_hist)
    for key, shape in expected_keys(cfg).items():
        e.load_tensor(key, synth_tensor(key, shape, seed=0))
    e.finalize()
    return e


def test_long_history_window_of_39_frames_matches_oracle(oc):
    """BASELINE.json config #5 doubles the context: history windows above 32 slots take the 64-slot temporal-attention form
    (2 heads per workgroup).  fp32 greedy, 39 history frames -> the new frame is token-exact against the oracle."""
    cfg = tiny_config(max_frame_len=48).greedy()
    scene = synthetic_scene(77, n_frames=39)
    e = Engine(cfg, precision="fp32", max_batch=1, max_cond_frames=40)
    e.load_state_dict(synthetic_state_dict(cfg, seed=5))
    e.finalize()
    out = e.rollout(scene, 1, cond_frames=40, input_cond_frames=39, seeds=[0])
    for m in MOD_ORDER:
        np.testing.assert_array_equal(out[m][:, :39], scene[m], err_msg=m)
        np.testing.assert_array_equal(out[m][:, 39], oc[f"long39_{m}"].astype(np.int64), err_msg=m)
    e.close()


def test_fp16_mode_refuses_a_weight_outside_the_half_range_and_rounds_like_torch():
    """precision fp16 stores matrix weights as IEEE half: a value that would become inf is refused at load (no silent garbage), and the
    host's float -> half rounding is torch's (round to nearest even, subnormals included), checked through a rollout-independent
    route: a weight tensor made of half-way cases loads without error in fp16 mode and is refused once it holds 7e4."""
    from umgen_amd.engine import UMGenError
    cfg = tiny_config().greedy()
    e = Engine(cfg, precision="fp16", max_batch=1, max_cond_frames=4)
    sd = synthetic_state_dict(cfg, seed=1)
    key = "transformer.OAR.1.mlp.c_fc.weight"
    assert key in sd, sorted(sd)[:5]
    w = np.array(sd[key], dtype=np.float32, copy=True)
    w.flat[:4] = [65504.0, -65504.0, 6.0e-8, 1.0 + 2.0 ** -11]      # largest half, a half subnormal, a tie (rounds to even)
    e.load_tensor(key, w)
    w.flat[5] = 7.0e4
    with pytest.raises(UMGenError, match="does not fit fp16"):
        e.load_tensor(key, w)
    e.close()


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_16bit_engine_vs_the_reference_under_autocast(precision):
    """The 16-bit engines against the REFERENCE'S OWN autocast arithmetic (UMGen.py:1604-1605): tests/golden/tiny_video_autocast_{fp16,bf16}.npz
    are one greedy frame of the imported reference with its autocast region live (torch CPU autocast in that type; make_golden.py), the engine is
    teacher-forced with that run's tokens and compared on the recorded conditioning rows, ego logits and AR logit rows.  Tolerance: tests/test_oracle.py
    AUTOCAST_BARS -- fp16 4e-3 on logits (measured 2.7e-3), bf16 3e-2 (measured 1.8e-2): the reference under autocast emits 16-bit LOGITS, so half an
    ulp of the recorded numbers alone is 1e-3 .. 2e-3 (fp16) / 8e-3 .. 1.6e-2 (bf16) and the fp32 restatement sits at the same distance (2.8e-3 / 2.0e-2);
    the north-star's 1e-3 is met against the fp32 reference goldens in fp32 mode (test_fp32_first_frame_activations_match_reference_golden)."""
    from tests.test_oracle import AUTOCAST_BARS
    g = np.load(os.path.join(GOLD, f"tiny_video_autocast_{precision}.npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    cfg = tiny_config(rule_constrain=False).greedy()
    e = make_engine(cfg, ws, precision)
    scene = synthetic_scene(sid, n_frames=icf)
    forced = {m: g[f"out_{m}"][0, icf].astype(np.int64) for m in MOD_ORDER}
    toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, trace=True, forced=forced)
    e.close()
    d = {"cond": float(np.abs(tr["cond"][COND_ROWS] - g["cond_rows"][0]).max()), "ego": float(np.abs(tr["ego_logits"] - g["ego_logits"][0]).max())}
    for m, pos in LOGIT_POS.items():
        d[m] = float(np.abs(tr[f"logits_{m}"][pos] - g[f"logits_{m}"]).max())
    agree = float(np.mean([np.mean(toks[m] == forced[m]) for m in MOD_ORDER]))
    print(f"tiny {precision} engine vs the reference under autocast: max abs deviation {d}; own arg-max == the reference's token at {agree:.4f} of the positions")
    bar = AUTOCAST_BARS[precision]
    assert d["cond"] <= bar["cond"] and d["ego"] <= bar["ego"], d
    for m in LOGIT_POS:
        assert d[m] <= bar["logits"], (m, d)
