#!/usr/bin/env python
"""Timing line of the VQ decoders: umgen_vq_decode of the two production
configurations (image: 16 x 32 tokens -> 3 x 256 x 512, map: 32 x 32 tokens -> 5 x 256 x 256), 20 frames per call like the reference
(decode_map.py:110-183), fp32 on the matrix cores (v_mfma_f32_32x32x2_f32) and -- UMGEN_FP32_MFMA=0 in a second process -- on the VALU
FMA-chain kernel.   python tools/vq_time.py  ->  one JSON line"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden.make_vq_golden import FULL, SEED  # noqa: E402
from umgen_amd.vq import VQDecoder, decoder_keys, synth_vq_tensor  # noqa: E402

res = {"mode": "valu" if os.environ.get("UMGEN_FP32_MFMA") == "0" else "mfma_f32_32x32x2"}
for name, cfg in FULL.items():
    d = VQDecoder(cfg)
    d.load_state_dict({k: synth_vq_tensor(k, s, SEED) for k, s in decoder_keys(cfg).items()})
    th, tw = cfg["token_hw"]
    codes = np.random.default_rng(1).integers(0, cfg["n_embed"], size=(20, th, tw)).astype(np.int64)
    d.decode_code(codes[:1])
    t0 = time.perf_counter()
    out = d.decode_code(codes)
    dt = time.perf_counter() - t0
    d.close()
    res[name] = {"frames": 20, "seconds": dt, "ms_per_frame": dt * 1e3 / 20, "out_shape": list(out.shape)}
print(json.dumps(res))
