import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
B = int(sys.argv[1]); out = sys.argv[2]
os.environ["UMGEN_DEBUG_DUMP_X"] = out
from umgen_amd.config import MOD_ORDER, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_scene
from umgen_amd.weights import synthetic_state_dict
cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=5, rule_constrain=False)
sd = synthetic_state_dict(cfg, seed=21)
s6, s7 = synthetic_scene(46, n_frames=2), synthetic_scene(47, n_frames=2)
two = {m: np.concatenate([s6[m], s7[m]]) for m in MOD_ORDER}
e = Engine(cfg, precision="bf16", max_batch=B, max_cond_frames=4, use_graphs=False)
e.load_state_dict(sd); e.finalize()
o = e.rollout(s6 if B == 1 else two, 1, cond_frames=3, input_cond_frames=2, seeds=[106, 107][:B])
print("B", B, "tok809", o["map"][0, 2, 809])
e.close()
