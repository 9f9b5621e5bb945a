"""Helpers shared by the -m gpu tests (ctypes calls into libumgen_hip.so's kernel-level hooks)."""
import ctypes as C

import numpy as np
import torch

from umgen_amd import _lib


def bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 ndarray -> raw bfloat16 bits (uint16), round-to-nearest-even like torch."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().view(torch.int16).numpy().view(np.uint16)


def bf16_round(a: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy()


def from_bits(b: np.ndarray) -> np.ndarray:
    return torch.from_numpy(b.view(np.int16)).view(torch.bfloat16).float().numpy()


# precision codes of the umgen_dbg_* hooks: 0 = fp32, 1 = bf16 (raw bits), 2 = fp16 (IEEE half bits)
def bits16(a: np.ndarray, code: int) -> np.ndarray:
    return bf16_bits(a) if code == 1 else np.ascontiguousarray(a, dtype=np.float32).astype(np.float16).view(np.uint16)


def round16(a: np.ndarray, code: int) -> np.ndarray:
    return bf16_round(a) if code == 1 else np.ascontiguousarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def from_bits16(b: np.ndarray, code: int) -> np.ndarray:
    return from_bits(b) if code == 1 else b.view(np.float16).astype(np.float32)


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def check(rc):
    assert rc == 0, f"libumgen_hip debug hook failed rc={rc}"


def lib():
    return _lib.load_library()
