"""VQ decoders on the GPU: the rollout's map / image tokens back to rasters.

Mirrors, on the C ABI of include/umgen.h (``umgen_vq_*``, csrc/vqdec.hip):
  * ``NormVQModel.decode_code`` / ``indices_to_quant`` + ``decode``    projects/tokenizer/vq_model.py:88-103, 126-150
  * the two model configurations                                       vq_model.py:153-202 (image: dim16 res512 f16, map: dim16 res256 f8)
  * ``Mapdecoder.decode_maps`` / ``Imagedecoder.decode_images``        projects/tools/decode_map.py:110-183
State-dict keys are the reference checkpoint's (``decoder.*``, ``post_quant_conv.*``, ``quantize.embedding.weight``); encoder and
EMA entries are ignored like ``strict=False``.  fp32 throughout (the reference decodes outside autocast).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import numpy as np

from . import _lib

# vq_model.py:153-202
IMAGE_VQ = dict(n_embed=8192, embed_dim=16, z_channels=256, resolution=512, out_ch=3, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2,
                attn_resolutions=(32,), post_quant_ks=3, post_quant_pad=1, token_hw=(16, 32))
MAP_VQ = dict(n_embed=8192, embed_dim=16, z_channels=16, resolution=256, out_ch=5, ch=128, ch_mult=(1, 2, 2, 4), num_res_blocks=2,
              attn_resolutions=(16,), post_quant_ks=1, post_quant_pad=0, token_hw=(32, 32))


class VQError(RuntimeError):
    pass


def decoder_keys(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every state-dict entry the decode path reads, with its shape (Decoder.__init__, vq_modules.py:294-383; VQModel / NormVQModel
    __init__, vq_model.py:22-59, 126-150)."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    L = len(cfg["ch_mult"])
    ch, z = cfg["ch"], cfg["z_channels"]

    def conv(k, cin, cout, ks):
        out[k + ".weight"] = (cout, cin, ks, ks)
        out[k + ".bias"] = (cout,)

    def norm(k, c):
        out[k + ".weight"] = (c,)
        out[k + ".bias"] = (c,)

    def res(k, cin, cout):
        norm(k + ".norm1", cin)
        conv(k + ".conv1", cin, cout, 3)
        norm(k + ".norm2", cout)
        conv(k + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(k + ".nin_shortcut", cin, cout, 1)

    def attn(k, c):
        norm(k + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{k}.{n}", c, c, 1)

    out["quantize.embedding.weight"] = (cfg["n_embed"], cfg["embed_dim"])
    conv("post_quant_conv", cfg["embed_dim"], z, cfg["post_quant_ks"])
    block_in = ch * cfg["ch_mult"][L - 1]
    curr_res = cfg["resolution"] // 2 ** (L - 1)
    conv("decoder.conv_in", z, block_in, 3)
    res("decoder.mid.block_1", block_in, block_in)
    attn("decoder.mid.attn_1", block_in)
    res("decoder.mid.block_2", block_in, block_in)
    for lv in reversed(range(L)):
        block_out = ch * cfg["ch_mult"][lv]
        for b in range(cfg["num_res_blocks"] + 1):
            res(f"decoder.up.{lv}.block.{b}", block_in, block_out)
            block_in = block_out
            if curr_res in cfg["attn_resolutions"]:
                attn(f"decoder.up.{lv}.attn.{b}", block_in)
        if lv != 0:
            conv(f"decoder.up.{lv}.upsample.conv", block_in, block_in, 3)
            curr_res *= 2
    norm("decoder.norm_out", block_in)
    conv("decoder.conv_out", block_in, cfg["out_ch"], 3)
    return out


def synth_vq_tensor(key: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    """Deterministic PyTorch-default-like initialisation per key (no checkpoint offline): the golden generator loads the same
    tensors into the reference's NormVQModel."""
    import zlib
    rng = np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, zlib.crc32(key.encode())]))
    if key.endswith("embedding.weight"):
        w = rng.standard_normal(shape).astype(np.float32)
        return (w / np.linalg.norm(w, axis=1, keepdims=True)).astype(np.float32)       # l2-normalised codes (NormEMAVectorQuantizer)
    if ".norm" in key:
        return (1.0 + 0.1 * (rng.random(shape, dtype=np.float32) * 2 - 1)).astype(np.float32) if key.endswith("weight") else \
            (0.1 * (rng.random(shape, dtype=np.float32) * 2 - 1)).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    bound = 1.0 / math.sqrt(max(fan_in, 1))
    return ((rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(bound)).astype(np.float32)


class VQDecoder:
    """One decoder = one GPU.  ``cfg``: IMAGE_VQ / MAP_VQ or a dict with the same keys (the reference's ddconfig)."""

    def __init__(self, cfg: dict, device: int = 0):
        self.lib = _lib.load_library()
        self.cfg = dict(cfg)
        L = len(cfg["ch_mult"])
        c = _lib.VQConfig(n_embed=cfg["n_embed"], embed_dim=cfg["embed_dim"], z_channels=cfg["z_channels"], ch=cfg["ch"], out_ch=cfg["out_ch"],
                          n_levels=L, num_res_blocks=cfg["num_res_blocks"], n_attn_res=len(cfg["attn_resolutions"]),
                          resolution=cfg["resolution"], post_quant_ks=cfg["post_quant_ks"], post_quant_pad=cfg["post_quant_pad"],
                          token_h=cfg["token_hw"][0], token_w=cfg["token_hw"][1], device=device)
        for i, m in enumerate(cfg["ch_mult"]):
            c.ch_mult[i] = m
        for i, r in enumerate(cfg["attn_resolutions"]):
            c.attn_resolutions[i] = r
        self._h = C.c_void_p()
        rc = self.lib.umgen_vq_create(C.byref(c), C.byref(self._h))
        if rc != 0:
            msg = self.lib.umgen_vq_last_error(self._h).decode() if self._h else "umgen_vq_create failed"
            if self._h:
                self.lib.umgen_vq_destroy(self._h)
                self._h = None
            raise VQError(f"umgen_vq_create: {msg} (rc={rc})")
        self.out_hw = (cfg["token_hw"][0] << (L - 1), cfg["token_hw"][1] << (L - 1))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.umgen_vq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise VQError(f"{what}: {self.lib.umgen_vq_last_error(self._h).decode()} (rc={rc})")
        return rc

    def load_state_dict(self, sd: Iterable, strict: bool = False):
        """``sd``: the VQ checkpoint's ``state_dict`` (torch tensors or arrays).  Returns (missing, unexpected) like torch."""
        items = sd.items() if hasattr(sd, "items") else sd
        want = decoder_keys(self.cfg)
        seen, unexpected = set(), []
        for k, v in items:
            a = np.ascontiguousarray(v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v), dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            rc = self._check(self.lib.umgen_vq_load_tensor(self._h, k.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), shape, a.ndim), f"load_tensor({k})")
            (seen.add(k) if rc == 0 else unexpected.append(k))
        missing = [k for k in want if k not in seen]
        if strict and (missing or unexpected):
            raise VQError(f"missing {missing[:4]}, unexpected {unexpected[:4]}")
        if not missing:
            self._check(self.lib.umgen_vq_finalize(self._h), "finalize")
        return missing, unexpected

    def decode_code(self, code_b) -> np.ndarray:
        """NormVQModel.decode_code (vq_model.py:93-97): codes [n, h, w] -> float32 [n, out_ch, H, W]."""
        codes = np.ascontiguousarray(np.asarray(code_b), dtype=np.int64)
        th, tw = self.cfg["token_hw"]
        if codes.ndim != 3 or codes.shape[1:] != (th, tw):
            raise VQError(f"codes have shape {codes.shape}, expected (n, {th}, {tw})")
        out = np.empty((codes.shape[0], self.cfg["out_ch"]) + self.out_hw, dtype=np.float32)
        self._check(self.lib.umgen_vq_decode(self._h, codes.shape[0], codes.ctypes.data_as(C.POINTER(C.c_int64)),
                                             out.ctypes.data_as(C.POINTER(C.c_float))), "decode")
        return out


class Mapdecoder:
    """decode_map.py:110-148: map tokens [1, T, 1024] (or [T, 1024]) -> reconstructions [T, 5, 256, 256] (``decode_maps`` of the
    reference additionally projects them to RGB with a seeded random 1x1 conv for visualisation: ``to_rgb`` below)."""

    def __init__(self, ckpt=None, device: int = 0, cfg: dict = MAP_VQ):
        self.dec = VQDecoder(cfg, device=device)
        sd = _state_dict_of(ckpt)
        if sd is not None:
            missing, _ = self.dec.load_state_dict(sd)
            if missing:
                raise VQError(f"map VQ checkpoint lacks {len(missing)} decoder tensors, e.g. {missing[:3]}")

    def decode_maps(self, map_tokens, H: int = 32, W: int = 32, rgb: bool = True) -> np.ndarray:
        t = np.asarray(map_tokens)
        t = t.reshape(-1, t.shape[-1]).reshape(-1, H, W)
        rec = np.concatenate([self.dec.decode_code(t[i:i + 20]) for i in range(0, t.shape[0], 20)])   # 20 frames per call like the reference
        if not rgb:
            return rec
        return np.concatenate([to_rgb(rec[i:i + 20]) for i in range(0, rec.shape[0], 20)])


class Imagedecoder:
    """decode_map.py:151-183: image tokens [1, T, 512] -> images [T, 3, 256, 512] in [-1, 1]."""

    def __init__(self, ckpt=None, device: int = 0, cfg: dict = IMAGE_VQ):
        self.dec = VQDecoder(cfg, device=device)
        sd = _state_dict_of(ckpt)
        if sd is not None:
            missing, _ = self.dec.load_state_dict(sd)
            if missing:
                raise VQError(f"image VQ checkpoint lacks {len(missing)} decoder tensors, e.g. {missing[:3]}")

    def decode_images(self, image_tokens, H: int = 16, W: int = 32) -> np.ndarray:
        t = np.asarray(image_tokens)
        if t.ndim == 1:
            t = t[None]
        t = t.reshape(-1, t.shape[-1]).reshape(-1, H, W)
        return np.concatenate([self.dec.decode_code(t[i:i + 20]) for i in range(0, t.shape[0], 20)])


def _state_dict_of(ckpt):
    """``ckpt``: a path (VQModel.init_from_ckpt, vq_model.py:65-73: ``torch.load(path)["state_dict"]``), a loaded checkpoint dict
    holding ``state_dict``, or the state dict itself."""
    if ckpt is None:
        return None
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        import torch
        ckpt = torch.load(ckpt, map_location="cpu")
    if hasattr(ckpt, "keys") and "state_dict" in ckpt and not any(str(k).startswith("decoder.") for k in ckpt.keys()):
        ckpt = ckpt["state_dict"]
    return ckpt


def to_rgb(x: np.ndarray, seed: int = 0) -> np.ndarray:
    """decode_map.py:25-30 (visualisation only: a seeded random 1x1 convolution to 3 channels, min-max scaled to [-1, 1]; torch's
    generator is part of its definition, so it runs through torch on the host)."""
    import torch
    import torch.nn.functional as F
    torch.manual_seed(seed)
    xt = torch.from_numpy(np.ascontiguousarray(x))
    w = torch.randn(3, xt.shape[1], 1, 1).to(xt)
    y = F.conv2d(xt, weight=w)
    y = 2.0 * (y - y.min()) / (y.max() - y.min()) - 1.0
    return y.numpy()
