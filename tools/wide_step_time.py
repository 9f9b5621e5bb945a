#!/usr/bin/env python3
"""Wall time of one 2x-width decode step through the BlockOAR layers (umgen_dbg_oar_step: upload x, the layers, download x) at a few KV lengths, for the
chip-wide engine (oar_engine_wide.hip) and the five-launch layer -- the A/B harness of the engine's measurement builds (UMGEN_LIB_PATH=...; their results
are garbage, only the time means something).  ~40 us of the time is the call's own upload / synchronise / download."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umgen_amd.config import wide2x_config  # noqa: E402
from umgen_amd.engine import Engine  # noqa: E402
from umgen_amd.weights import expected_keys, synth_tensor  # noqa: E402

cfg = wide2x_config()
eng = Engine(cfg, precision=os.environ.get("PRECISION", "bf16"), max_batch=1, max_cond_frames=2)
for key, shape in expected_keys(cfg).items():
    eng.load_tensor(key, synth_tensor(key, shape, seed=0))
eng.finalize()
rng = np.random.default_rng(0)
x = rng.standard_normal((1, cfg.n_embd)).astype(np.float32)
modes = [(3, "chip-wide engine"), (0, "five launches")] if not os.environ.get("ENGINE_ONLY") else [(3, "chip-wide engine")]
for L in [int(v) for v in os.environ.get("LS", "64,1100,2200").split(",")]:
    for mode, name in modes:
        for _ in range(3):
            eng.dbg_oar_step(x, L, mode)
        n = 30
        t = time.perf_counter()
        for _ in range(n):
            eng.dbg_oar_step(x, L, mode)
        dt = (time.perf_counter() - t) / n * 1e6
        print(f"KV length {L:5d}  {name:18s} {dt:8.1f} us per step  ({dt / cfg.n_oar_layer:6.2f} us per layer)")
eng.close()
