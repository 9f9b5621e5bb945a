"""configs[2] of BASELINE.json at full size: UMGen_Large, --infer_task control (13 history frames, control pose + one controlled
agent per scene), 4 scenes per GPU.  Prints scene-tokens/s for a short rollout and checks it against four one-scene rollouts."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from umgen_amd.config import MOD_ORDER, large_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_control, synthetic_scene
from umgen_amd.weights import synthetic_items

cfg = large_config()
B, NEW = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 3
e = Engine(cfg, precision="bf16", max_batch=B, max_cond_frames=20)
e.load_state_dict(synthetic_items(cfg, seed=0)); e.finalize()
scenes = [synthetic_scene(2000 + i, n_frames=13) for i in range(B)]
ctl = [synthetic_control(2000 + i, n_frames=NEW) for i in range(B)]
tok = {m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}
init = {k: np.concatenate([c[k] for c in ctl]) for k in ("pose", "bbox3d")}
seeds = [11, 12, 13, 14]
e.rollout(tok, 1, cond_frames=20, input_cond_frames=13, init_tokens=init, control_test=True, seeds=seeds)   # warm-up (graphs)
t0 = time.perf_counter()
out = e.rollout(tok, NEW, cond_frames=20, input_cond_frames=13, init_tokens=init, control_test=True, seeds=seeds)
dt = time.perf_counter() - t0
print(f"control, B={B}, {NEW} frames: {B * NEW * 2207 / dt:.0f} scene-tokens/s, {dt / NEW:.2f} s/frame (4 scenes)", e.timings()["overlapped_frames"])
for i in range(B):
    one = e.rollout(scenes[i], NEW, cond_frames=20, input_cond_frames=13, init_tokens=ctl[i], control_test=True, seeds=[seeds[i]])
    for m in MOD_ORDER:
        assert np.array_equal(one[m], out[m][i:i + 1]), (i, m)
print("batch of 4 == four single rollouts (token-exact)")
e.close()
