"""decode_chain_kernel (UMGEN_DECODE_CHAIN=1) against the five launches per layer of the batched decode layer: tokens must be equal bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from umgen_amd.config import MOD_ORDER, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_scene
from umgen_amd.weights import synthetic_state_dict

B = int(os.environ.get("B", "33"))
cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=int(os.environ.get("LAYERS", "5")), rule_constrain=False)
sd = synthetic_state_dict(cfg, seed=21)
scenes = [synthetic_scene(40 + i, n_frames=2) for i in range(B)]
batch = {m: np.concatenate([s[m] for s in scenes]) for m in MOD_ORDER}
outs = {}
for chain in ("0", "1"):
    os.environ["UMGEN_DECODE_CHAIN"] = chain
    os.environ["UMGEN_DECODE_BATCHED"] = "1"
    os.environ["UMGEN_DECODE_MS"] = "0"
    e = Engine(cfg, precision="bf16", max_batch=B, max_cond_frames=4)
    e.load_state_dict(sd)
    e.finalize()
    t = time.perf_counter()
    outs[chain] = e.rollout(batch, 1, cond_frames=3, input_cond_frames=2, seeds=list(range(B)))
    dt = time.perf_counter() - t
    tm = e.timings()
    print(f"chain={chain}: frame {dt * 1e3:.1f} ms, decode_batched {tm['decode_batched']} lanes {tm['decode_lanes']} oar_ms {tm['oar_ms']:.1f}", flush=True)
    e.close()
bad = 0
for m in MOD_ORDER:
    d = int((outs["0"][m] != outs["1"][m]).sum())
    bad += d
    print(m, "tokens differing:", d, "of", outs["0"][m].size)
print("EQUAL" if bad == 0 else "DIFFERENT")
