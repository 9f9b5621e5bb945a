"""Oracle outputs of the tiny-config parity cases that the -m gpu tests compare the engine with, recorded once here so that the
GPU box does not spend minutes of its (slow, shared) host CPU re-running the CPU oracle:

    python tests/golden/make_oracle_cases.py [case ...]      ->  tests/golden/oracle_cases.npz

The oracle itself is pinned on the reference (tests/test_oracle.py, tests/test_oracle_vs_reference.py); tests/test_oracle.py also
re-runs one of these cases live so that a change of the oracle cannot leave stale vectors behind."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.umgen_oracle import OracleUMGen  # noqa: E402
from umgen_amd.config import MOD_ORDER, tiny_config  # noqa: E402
from umgen_amd.synth import synthetic_scene  # noqa: E402
from umgen_amd.weights import synthetic_state_dict  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "oracle_cases.npz")
COND_ROWS = [0, 1, 4, 5, 6, 500, 1030, 1031, 1032, 1042, 1692, 1693, 1694, 2000, 2206]
LOGIT_POS = {"map": [0, 1, 511, 1023], "bbox3d": [0, 9, 10, 11, 330, 659], "image": [0, 255, 511]}


def logit_summary(out, prefix, logits):
    for m, pos in LOGIT_POS.items():
        lg = logits[m]
        srt = np.sort(lg, axis=-1)
        out[f"{prefix}_logits_{m}"] = lg[pos].astype(np.float32)
        out[f"{prefix}_argmax_{m}"] = lg.argmax(-1).astype(np.int16)
        out[f"{prefix}_gap_{m}"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)


def case_sampled(out):
    """k = 5/5/16 sampling with the counter-based RNG: two scenes, one new frame each."""
    cfg = tiny_config()
    o = OracleUMGen(cfg, synthetic_state_dict(cfg, seed=3))
    for i, seed in enumerate((111, 222)):
        ref = o.inference(1, 3, synthetic_scene(10 + i, n_frames=2), input_cond_frames=2, seed=seed)
        for m in MOD_ORDER:
            out[f"sampled_{i}_{m}"] = ref[m].astype(np.int16)


def case_bf16(out):
    """Rounding-aware oracle on the scene of the reference golden tiny_video_greedy: teacher-forced trace + free-running greedy."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "tiny_video_greedy.npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    cfg = tiny_config().greedy()
    sd = synthetic_state_dict(cfg, seed=ws)
    scene = synthetic_scene(sid, n_frames=icf)
    forced = {m: g[f"out_{m}"][0, icf].astype(np.int64)[None] for m in MOD_ORDER}
    o = OracleUMGen(cfg, sd, weight_dtype="bf16_engine")
    o.inference(1, cf, scene, input_cond_frames=icf, trace=True, forced=forced)
    out["bf16_cond_rows"] = o.trace["cond"][0][COND_ROWS].astype(np.float32)
    out["bf16_cond_rms"] = np.float32(np.sqrt((o.trace["cond"][0].astype(np.float64) ** 2).mean()))
    out["bf16_ego_logits"] = o.trace["ego_logits"][0].astype(np.float32)
    logit_summary(out, "bf16", o.trace["logits"][0])
    o2 = OracleUMGen(cfg, sd, weight_dtype="bf16_engine")
    ref = o2.inference(1, cf, scene, input_cond_frames=icf, trace=True)
    for m in MOD_ORDER:
        out[f"bf16_free_{m}"] = ref[m][0, icf].astype(np.int16)
    for m in ("map", "bbox3d", "image"):
        srt = np.sort(o2.trace["logits"][0][m], axis=-1)
        out[f"bf16_free_gap_{m}"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)


def case_topp(out):
    cfg = tiny_config()
    cfg.sample_method = "topp"
    cfg.rule_constrain = False
    o = OracleUMGen(cfg, synthetic_state_dict(cfg, seed=5))
    ref = o.inference(1, 2, synthetic_scene(21, n_frames=2), input_cond_frames=2, seed=77, trace=True)
    for m in MOD_ORDER:
        out[f"topp_{m}"] = ref[m][0, 2].astype(np.int16)
    logit_summary(out, "topp", o.trace["logits"][0])


def case_pad_avoid(out):
    cfg = tiny_config()
    o = OracleUMGen(cfg, synthetic_state_dict(cfg, seed=3))
    ref = o.inference(1, 3, synthetic_scene(10, n_frames=2), input_cond_frames=2, seed=5)
    for m in MOD_ORDER:
        out[f"padavoid_{m}"] = ref[m][0, 2].astype(np.int16)
    out["padavoid_counters"] = np.array([o.counters.get(k, 0) for k in ("pad_avoid", "rule_checked", "rule_blanked")], dtype=np.int32)


def case_edge(out):
    """T_in = 1 history frame, two new frames: the window grows 1 -> 2 and then slides."""
    cfg = tiny_config().greedy()
    ref = OracleUMGen(cfg, synthetic_state_dict(cfg, seed=9)).inference(2, 2, synthetic_scene(30, n_frames=1), input_cond_frames=1)
    for m in MOD_ORDER:
        out[f"edge_{m}"] = ref[m].astype(np.int16)


def case_long39(out):
    cfg = tiny_config(max_frame_len=48).greedy()
    ref = OracleUMGen(cfg, synthetic_state_dict(cfg, seed=5)).inference(1, 40, synthetic_scene(77, n_frames=39), input_cond_frames=39, seed=0)
    for m in MOD_ORDER:
        out[f"long39_{m}"] = ref[m][:, 39].astype(np.int16)


CASES = {"sampled": case_sampled, "bf16": case_bf16, "topp": case_topp, "pad_avoid": case_pad_avoid, "edge": case_edge, "long39": case_long39}


def main(names):
    out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
    for n in names or list(CASES):
        print("recording", n, flush=True)
        CASES[n](out)
    np.savez_compressed(PATH, **out)
    print("wrote", PATH, os.path.getsize(PATH))


if __name__ == "__main__":
    main(sys.argv[1:])
