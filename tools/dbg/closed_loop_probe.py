"""Self-consistency of a greedy rollout frame: every decoded map / image token must be the arg-max of the SAME engine's teacher-forced
logits (trace mode) of that frame.  A mismatch at a position whose top-2 gap is far above the mode's rounding noise means the rollout
path (hipGraph step replay) and the trace path (one step at a time) computed different logits."""
import sys
import numpy as np
sys.path.insert(0, ".")
from umgen_amd.config import MOD_ORDER, large_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_scene
from umgen_amd.weights import expected_keys, synth_tensor

cfg = large_config().greedy()
T = 20
scene = synthetic_scene(0, n_frames=T)
window = {m: scene[m][0][-T:] for m in MOD_ORDER}
traces = {}
for prec in sys.argv[1:] or ("fp32", "fp16", "bf16"):
    e = Engine(cfg, precision=prec, max_batch=1, max_cond_frames=T)
    for key, shape in expected_keys(cfg).items():
        e.load_tensor(key, synth_tensor(key, shape, seed=0))
    e.finalize()
    out = e.rollout(scene, 1, cond_frames=T, input_cond_frames=T, seeds=[0])
    toks = {m: out[m][0, T] for m in MOD_ORDER}
    _, tr = e.frame(window, frame_idx=0, seed=0, trace=True, forced=toks)
    traces[prec] = (toks, tr)
    for m in ("map", "image"):
        lg = tr[f"logits_{m}"]
        am = lg.argmax(-1)
        bad = np.nonzero(am != toks[m])[0]
        print(prec, m, "rollout token != arg-max of own trace at", bad[:10], "of", len(am))
        for i in bad[:5]:
            s = np.sort(lg[i])
            print("    idx", i, "rollout tok", int(toks[m][i]), "trace argmax", int(am[i]), "top2 gap", float(s[-1] - s[-2]), "logit of rollout tok", float(lg[i][toks[m][i]]), "max", float(s[-1]))
    e.close()
if "fp32" in traces:
    for prec in traces:
        if prec == "fp32":
            continue
        for m in ("map", "bbox3d", "image"):
            same_prefix = np.cumprod(traces[prec][0][m] == traces["fp32"][0][m]).astype(bool)
            d = np.abs(traces[prec][1][f"logits_{m}"] - traces["fp32"][1][f"logits_{m}"]).max(-1)
            n = int(same_prefix.sum())
            print(prec, m, "tokens equal to fp32's up to index", n, "max |dlogit| vs fp32 over the common prefix", float(d[:max(n, 1)].max()), "at", int(d[:max(n, 1)].argmax()))
