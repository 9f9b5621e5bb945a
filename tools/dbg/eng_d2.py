import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from umgen_amd.config import MOD_ORDER, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_scene
from umgen_amd.weights import synthetic_state_dict

cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=5, rule_constrain=False)
sd = synthetic_state_dict(cfg, seed=21)
s6, s7 = synthetic_scene(46, n_frames=2), synthetic_scene(47, n_frames=2)
two = {m: np.concatenate([s6[m], s7[m]]) for m in MOD_ORDER}
def run(tag, B, graphs, D=None, engine="1"):
    os.environ["UMGEN_DECODE_ENGINE"] = engine
    if D: os.environ["UMGEN_DEBUG_ENGINE_D"] = str(D)
    else: os.environ.pop("UMGEN_DEBUG_ENGINE_D", None)
    e = Engine(cfg, precision="bf16", max_batch=B, max_cond_frames=4, use_graphs=graphs)
    e.load_state_dict(sd); e.finalize()
    if B == 1: o = e.rollout(s6, 1, cond_frames=3, input_cond_frames=2, seeds=[106])
    else: o = e.rollout(two, 1, cond_frames=3, input_cond_frames=2, seeds=[106, 107])
    e.close()
    print(tag, "tok809 =", o["map"][0, 2, 809], flush=True)
    return o
ref = run("B=1 D=8 graphs", 1, True)
run("B=1 D=4 graphs", 1, True, D=4)
run("B=1 D=1 graphs", 1, True, D=1)
run("B=1 D=8 eager ", 1, False)
run("B=2 D=4 graphs", 2, True)
run("B=2 D=4 eager ", 2, False)
run("B=2 launches graphs", 2, True, engine="0")
run("B=2 D=2 graphs", 2, True, D=2)
