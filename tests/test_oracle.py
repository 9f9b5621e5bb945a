"""Pins the CPU oracle (oracle/umgen_oracle.py) against golden vectors produced by RUNNING THE REFERENCE
(tests/golden/make_golden.py): token sequences under full greedy decoding must be identical, recorded
activations / logits must agree to 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle.umgen_oracle import OracleUMGen
from umgen_amd.config import tiny_config
from umgen_amd.synth import golden_init_tokens, synthetic_control, synthetic_scene
from umgen_amd.weights import synthetic_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LOGIT_POS = {"map": [0, 1, 511, 1023], "bbox3d": [0, 9, 10, 11, 330, 659], "image": [0, 255, 511]}
COND_ROWS = [0, 1, 4, 5, 6, 500, 1030, 1031, 1032, 1042, 1692, 1693, 1694, 2000, 2206]


@pytest.mark.parametrize("name", ["tiny_video_greedy", "tiny_control_greedy", "tiny_boxctl_greedy", "tiny_grow_boxctl_greedy", "tiny_grow_control_greedy", "tiny_mapgiven_greedy", "tiny_mapboxgiven_greedy"])
def test_oracle_matches_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
    cfg = tiny_config().greedy()
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    o = OracleUMGen(cfg, synthetic_state_dict(cfg, seed=ws))
    out = o.inference(nf, cf, synthetic_scene(sid, n_frames=icf), input_cond_frames=icf,
                      init_tokens=golden_init_tokens(sid, nf, ctl),
                      control_test=ctl in (1, 2), trace=True)
    for m in ("pose", "map", "bbox3d", "image"):
        np.testing.assert_array_equal(out[m], g[f"out_{m}"].astype(np.int64), err_msg=m)
    cond = np.stack(o.trace["cond"])[:, COND_ROWS]
    np.testing.assert_allclose(cond, g["cond_rows"], atol=1e-5, rtol=0)
    if "ego_logits" in g.files:
        np.testing.assert_allclose(np.stack(o.trace["ego_logits"]), g["ego_logits"], atol=1e-5, rtol=0)
    for m, pos in LOGIT_POS.items():
        if f"logits_{m}" in g.files:      # (a GIVEN modality has no logits: tiny_mapgiven_greedy's map)
            np.testing.assert_allclose(o.trace["logits"][0][m][pos], g[f"logits_{m}"], atol=1e-5, rtol=0)
    # the fixtures must exercise the host-side control flow, not just the transformer
    if "grow" not in name and "given" not in name:      # (the growing-window case is there for the window arithmetic: 2 -> 5 history frames, then it slides)
        assert o.counters.get("rule_blanked", 0) > 0 and o.counters.get("rule_free", 0) > 0
    if ctl in (1, 2):
        assert o.counters.get("control_resample", 0) > 0


def test_recorded_oracle_cases_are_current():
    """tests/golden/oracle_cases.npz (what the -m gpu tests compare the engine with instead of re-running the oracle on the GPU
    box) still is what the oracle produces: the cheapest case -- one history frame, two new frames, sliding window -- is
    re-recorded live and compared, and the sampler helper the recorded top-p / top-k cases depend on is pinned bit for bit."""
    from tests.golden.make_oracle_cases import CASES, PATH

    rec = np.load(PATH)
    live = {}
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    CASES["edge"](live)
    for k, v in live.items():
        np.testing.assert_array_equal(rec[k], v, err_msg=k)
    for name in CASES:      # every case of the generator is present in the file
        assert any(k.startswith({"pad_avoid": "padavoid"}.get(name, name) + "_") for k in rec.files), name
