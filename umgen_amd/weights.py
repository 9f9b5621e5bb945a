"""State-dict key set consumed by the rollout, and a deterministic synthetic-weight generator.

Key names are exactly the reference's ``UMGen.state_dict()`` names (UMGen.py:176-261; checkpoint
layout ``ckpt["model_state"]["module"]``, infer_fun.py:43-50) so a real ``UMGen_Large.pt`` loads
unchanged.  The real checkpoint is not available offline (README.md:66-81), so tests and the bench
use ``synthetic_state_dict``: PyTorch-default-like initialisation (Linear: U(+-1/sqrt(fan_in)),
Embedding: N(0,1)) drawn from a per-key PCG64 stream -> identical weights in the imported
reference, the oracle and the HIP engine without committing any blob.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import numpy as np

from .config import RolloutConfig, SEQ_LEN

_TAR_SUBS = (("ln_1", "spatial_attn_1", "ln_2", "mlp1"),
             ("ln_3", "temporal_attn", "ln_4", "mlp2"),
             ("ln_5", "spatial_attn_2", "ln_6", "mlp3"))


def _attn_keys(prefix: str, E: int) -> Iterable[Tuple[str, Tuple[int, ...]]]:
    yield prefix + ".c_attn.weight", (3 * E, E)
    yield prefix + ".c_attn.bias", (3 * E,)
    yield prefix + ".c_proj.weight", (E, E)
    yield prefix + ".c_proj.bias", (E,)


def _mlp_keys(prefix: str, E: int) -> Iterable[Tuple[str, Tuple[int, ...]]]:
    yield prefix + ".c_fc.weight", (4 * E, E)
    yield prefix + ".c_proj.weight", (E, 4 * E)


def block_tar_keys(prefix: str, E: int):
    """module.py:296-326 (BlockTAR)."""
    for ln_a, attn, ln_b, mlp in _TAR_SUBS:
        yield f"{prefix}.{ln_a}.weight", (E,)
        yield from _attn_keys(f"{prefix}.{attn}", E)
        yield f"{prefix}.{ln_b}.weight", (E,)
        yield from _mlp_keys(f"{prefix}.{mlp}", E)


def block_oar_keys(prefix: str, E: int):
    """module.py:378-397 (BlockOAR)."""
    yield f"{prefix}.ln_1.weight", (E,)
    yield from _attn_keys(f"{prefix}.temporal_attn", E)
    yield f"{prefix}.ln_2.weight", (E,)
    yield from _mlp_keys(f"{prefix}.mlp", E)


def decoder_keys(prefix: str, E: int):
    """module.py:630-655 (Decoder) + 454-470 (FlashCrossAttention)."""
    yield f"{prefix}.ln_1.weight", (E,)
    yield from _attn_keys(f"{prefix}.self_attn", E)
    yield f"{prefix}.ln_2.weight", (E,)
    yield f"{prefix}.ln_3.weight", (E,)
    for n in ("q_attn", "k_attn", "v_attn", "c_proj"):
        yield f"{prefix}.cross_attn.{n}.weight", (E, E)
        yield f"{prefix}.cross_attn.{n}.bias", (E,)
    yield f"{prefix}.ln_4.weight", (E,)
    yield from _mlp_keys(f"{prefix}.mlp1", E)


def expected_keys(cfg: RolloutConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every state-dict entry the rollout reads (checkpoint row)."""
    E = cfg.n_embd
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    t = "transformer."
    out[t + "egoe.weight"] = (3, E)
    out[t + "axe.weight"] = (cfg.aux_vocab_size, E)
    out[t + "be.weight"] = (cfg.bbox3d_vocab_size, E)
    out[t + "tpe.weight"] = (cfg.max_frame_len, E)
    out[t + "spe.weight"] = (SEQ_LEN, E)
    out[t + "tske.weight"] = (cfg.task_num, E)
    for name, n in (("ego_tar", cfg.n_ego_tar_layer), ("map_tar", cfg.n_map_tar_layer),
                    ("box_tar", cfg.n_box_tar_layer), ("TAR", cfg.n_tar_layer)):
        for i in range(n):
            out.update(block_tar_keys(f"{t}{name}.{i}", E))
    for i in range(cfg.n_oar_layer):
        out.update(block_oar_keys(f"{t}OAR.{i}", E))
    for i in range(cfg.n_ego_ca_layer):
        out.update(decoder_keys(f"{t}ego_cross_attn.{i}", E))
    for ln in ("ln_ego_tar", "ln_ego", "ln_tar", "ln_oar", "ln_map_tar", "ln_box_tar"):
        out[t + ln + ".weight"] = (E,)
    out[t + "head_ego.weight"] = (cfg.pose_vocab_size, E)
    out[t + "head_ar_map.weight"] = (cfg.map_vocab_size, E)
    out[t + "head_ar_bbox3d.weight"] = (cfg.bbox3d_vocab_size, E)
    out[t + "head_tar_bbox3d.weight"] = (cfg.bbox3d_vocab_size, E)
    out[t + "head_ar_img.weight"] = (cfg.img_vocab_size, E)
    out["map_mlp_pre.c_fc.weight"] = (4 * E, cfg.n_map_embd)
    out["map_mlp_pre.c_proj.weight"] = (E, 4 * E)
    out["img_mlp_pre.c_fc.weight"] = (4 * E, cfg.n_img_embd)
    out["img_mlp_pre.c_proj.weight"] = (E, 4 * E)
    out["map_codebook.weight"] = (cfg.map_vocab_size, cfg.n_map_embd)
    out["img_codebook.weight"] = (cfg.img_vocab_size, cfg.n_img_embd)
    return out


# keys a checkpoint may carry that override build-computed constants (UMGen.py:257-261, strict=False)
OPTIONAL_KEYS = ("fouier_pe", "bbox3d_spatial_posi", "grid_center_posi_embedding")


def is_matrix_weight(key: str) -> bool:
    """True for the nn.Linear weights the engine stores in its precision dtype (bf16 / fp16 / fp32): every *.weight that is not an
    embedding table, a LayerNorm weight or a codebook (those stay fp32 on the device in every mode)."""
    if not key.endswith(".weight") or key.endswith("codebook.weight"):
        return False
    mod = key.split(".")[-2]
    return not (mod.startswith("ln_") or mod in ("egoe", "axe", "be", "tpe", "spe", "tske"))


def n_params(cfg: RolloutConfig) -> int:
    return sum(int(np.prod(s)) for s in expected_keys(cfg).values())


def _rng_for(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, zlib.crc32(key.encode())]))


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    rng = _rng_for(key, seed)
    last = key.rsplit(".", 2)
    if key.endswith("codebook.weight"):
        return (rng.random(shape, dtype=np.float32) * 2.0 - 1.0)
    if ".ln_" in key or key.split(".")[-2].startswith("ln_"):
        return (1.0 + 0.1 * (rng.random(shape, dtype=np.float32) * 2.0 - 1.0)).astype(np.float32)
    mod = last[-2]
    if mod in ("egoe", "axe", "be", "tpe", "spe", "tske"):
        return rng.standard_normal(shape, dtype=np.float32)
    if key.endswith(".bias"):
        fan_in = shape[0] // 3 if ".c_attn." in key else shape[0]
        # bias of Linear(fan_in=E): c_attn has 3E outputs over E inputs, the others E over E
        bound = 1.0 / math.sqrt(fan_in)
        return ((rng.random(shape, dtype=np.float32) * 2.0 - 1.0) * bound).astype(np.float32)
    bound = 1.0 / math.sqrt(shape[-1])
    return ((rng.random(shape, dtype=np.float32) * 2.0 - 1.0) * np.float32(bound)).astype(np.float32)


def synthetic_items(cfg: RolloutConfig, seed: int = 0):
    """Yields (key, float32 ndarray) one tensor at a time (keeps host RAM flat for UMGen_Large)."""
    for key, shape in expected_keys(cfg).items():
        yield key, synth_tensor(key, shape, seed)


def synthetic_state_dict(cfg: RolloutConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    return OrderedDict(synthetic_items(cfg, seed))
