import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from umgen_amd.config import MOD_ORDER, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_scene
from umgen_amd.weights import synthetic_state_dict

cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=5, rule_constrain=False)
sd = synthetic_state_dict(cfg, seed=21)
B = 8
scenes = [synthetic_scene(40 + i, n_frames=2) for i in range(B)]
seeds = [100 + i for i in range(B)]
e = Engine(cfg, precision="bf16", max_batch=B, max_cond_frames=4)
e.load_state_dict(sd); e.finalize()
def cat(idx): return {m: np.concatenate([scenes[i][m] for i in idx]) for m in MOD_ORDER}
s6a = e.rollout(scenes[6], 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[6]])
s6b = e.rollout(scenes[6], 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[6]])
print("single twice equal:", all(np.array_equal(s6a[m], s6b[m]) for m in MOD_ORDER))
b1 = e.rollout(cat(range(8)), 1, cond_frames=3, input_cond_frames=2, seeds=seeds)
b2 = e.rollout(cat(range(8)), 1, cond_frames=3, input_cond_frames=2, seeds=seeds)
print("batch twice equal:", all(np.array_equal(b1[m], b2[m]) for m in MOD_ORDER))
for m in MOD_ORDER:
    d = np.argwhere(b1[m][6:7] != s6a[m])
    if len(d): print("batch vs single", m, d[:5].tolist(), b1[m][6:7][tuple(d[0])], s6a[m][tuple(d[0])], "n=", len(d))
perm = [6, 1, 2, 3, 4, 5, 0, 7]
b3 = e.rollout(cat(perm), 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[i] for i in perm])
for m in MOD_ORDER:
    d = np.argwhere(b3[m][0:1] != s6a[m])
    if len(d): print("perm slot0 (scene 6) vs single", m, d[:5].tolist(), "n=", len(d))
    d = np.argwhere(b3[m][6:7] != b1[m][0:1])
    if len(d): print("perm slot6 (scene 0) vs batch slot 0", m, d[:5].tolist(), "n=", len(d))
# two scenes: D = 4
b4 = e.rollout(cat([6, 7]), 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[6], seeds[7]])
for m in MOD_ORDER:
    d = np.argwhere(b4[m][0:1] != s6a[m])
    if len(d): print("B=2 slot0 (scene 6) vs single", m, d[:5].tolist(), "n=", len(d))
e.close()
os.environ["UMGEN_DECODE_ENGINE"] = "0"
e0 = Engine(cfg, precision="bf16", max_batch=B, max_cond_frames=4)
e0.load_state_dict(sd); e0.finalize()
r6 = e0.rollout(scenes[6], 1, cond_frames=3, input_cond_frames=2, seeds=[seeds[6]])
print("engine-off single tok809:", r6["map"][0, 2, 809], " engine single:", s6a["map"][0, 2, 809], " engine batch:", b1["map"][6, 2, 809])
for name, o in (("engine single", s6a), ("engine batch slot6", {m: b1[m][6:7] for m in MOD_ORDER})):
    n = {m: int((o[m][:, 2] != r6[m][:, 2]).sum()) for m in MOD_ORDER}
    first = {m: (np.argwhere(o[m][0, 2] != r6[m][0, 2])[:3].ravel().tolist()) for m in MOD_ORDER}
    print(name, "vs engine-off: mismatches", n, first)
e0.close()
