"""Dumps the engine's teacher-forced trace (all conditioning rows, ego logits, OAR logit rows) of the ensemble test cases to
gpurun_out/trace_<width>_<precision>.npz, for offline comparison with oracle variants (which modelling assumption explains a gap)."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from tests.golden.make_ensemble import case
from umgen_amd.config import MOD_ORDER
from umgen_amd.engine import Engine

os.makedirs("gpurun_out", exist_ok=True)
for width in sys.argv[1:] or ["tiny", "full_width"]:
    cfg, sd, scene, cf, icf, forced, cond_rows, logit_pos = case(width)
    for prec in ("bf16", "fp16", "fp32"):
        e = Engine(cfg, precision=prec, max_cond_frames=4)
        e.load_state_dict(sd)
        e.finalize()
        toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, trace=True, forced={m: forced[m][0] for m in MOD_ORDER})
        e.close()
        np.savez_compressed(f"gpurun_out/trace_{width}_{prec}.npz", cond=tr["cond"].astype(np.float32), ego=tr["ego_logits"],
                            **{f"logits_{m}": tr[f"logits_{m}"][pos] for m, pos in logit_pos.items()})
        print("dumped", width, prec, flush=True)
