#!/usr/bin/env python
"""Isolated timing of the batched decode layer's five launches (csrc/decode_batched.hip) through umgen_dbg_batched_layer_bench:
   python tools/batched_bench.py [L]   ->  per M (scenes) the average microseconds of q|k|v, attention, c_proj, c_fc, mlp c_proj and
of the five back to back, plus the HBM roofline the layer's algorithmic bytes would need at 8 TB/s."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umgen_amd import _lib  # noqa: E402

lib = _lib.load_library()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
for prec, name in ((1, "bf16"), (2, "fp16")):
    for M in (1, 8, 16, 32, 48, 64):
        us = np.zeros(6, np.float32)
        rc = lib.umgen_dbg_batched_layer_bench(prec, M, L, 200, us.ctypes.data_as(C.POINTER(C.c_float)))
        assert rc == 0, rc
        bytes_layer = 12 * 768 * 768 * 2 + M * (L + 1) * 2 * 768 * 2
        print(f"{name} M={M:2d} L={L}: qkv {us[0]:.1f} attn {us[1]:.1f} proj {us[2]:.1f} fc {us[3]:.1f} proj2 {us[4]:.1f} | sequence {us[5]:.1f} us "
              f"(sum {us[:5].sum():.1f}) | {bytes_layer / 1e6:.1f} MB per layer -> {bytes_layer / (us[5] * 1e-6) / 1e12:.2f} TB/s, roof {bytes_layer / 8e12 * 1e6:.1f} us")
