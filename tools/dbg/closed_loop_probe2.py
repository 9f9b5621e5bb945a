"""Where do a rollout frame and the trace path part?  (a) umgen_rollout, 1 frame; (b) umgen_frame free-running with trace;
(c) umgen_frame teacher-forced with (a)'s tokens."""
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
for prec in sys.argv[1:] or ("bf16",):
    e = Engine(cfg, precision=prec, max_batch=1, max_cond_frames=T)
    for key, shape in expected_keys(cfg).items():
        e.load_tensor(key, synth_tensor(key, shape, seed=0))
    e.finalize()
    out = e.rollout(scene, 1, cond_frames=T, input_cond_frames=T, seeds=[0])
    a = {m: out[m][0, T] for m in MOD_ORDER}
    b, trb = e.frame(window, frame_idx=0, seed=0, trace=True)
    c, trc = e.frame(window, frame_idx=0, seed=0, trace=True, forced=a)
    for m in MOD_ORDER:
        ne = np.nonzero(a[m] != b[m])[0]
        print(prec, m, "rollout vs free frame(): differ at", ne[:8], "n", len(ne), "| forced frame() returns rollout tokens:", bool((c[m] == a[m]).all()))
    for m in ("map", "bbox3d", "image"):
        am = trb[f"logits_{m}"].argmax(-1)
        print(prec, m, "free frame(): token != arg-max of its own trace at", np.nonzero(am != b[m])[0][:8], "| max |logits free - forced|", float(np.abs(trb[f"logits_{m}"] - trc[f"logits_{m}"]).max()))
    print(prec, "counters free", trb["counters"], "forced", trc["counters"])
    print(prec, "cond max |free - forced|", float(np.abs(trb["cond"] - trc["cond"]).max()))
    e.close()
