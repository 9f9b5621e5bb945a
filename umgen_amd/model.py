"""Drop-in ``UMGen`` model class for the reference's callers (evaluate.py:193-220, model_pl.py:173-175, 237-239).

Same surface as projects/models/UMGen.py: ``UMGen(config)`` with the reference's argparse.Namespace config, an
``nn.Module`` (so Lightning can own/move it), ``load_state_dict(sd, strict=False)`` with the reference's key names,
``eval()``, ``.cpu()/.to()``, and ``inference(**kwargs) -> Dict[str, np.ndarray int64 [1, T_out, S_mod]]``.
Internally nothing is a torch op: tokens go to the HIP engine through the C ABI (include/umgen.h).  There is no CPU
fallback -- constructing the engine without a GPU / without libumgen_hip.so raises.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from .config import MOD_ORDER, RolloutConfig
from .engine import Engine, UMGenError
from .registry import MODELS
from .weights import OPTIONAL_KEYS, expected_keys


@MODELS.register_module()
class UMGen(nn.Module):
    def __init__(self, config, precision: str = "bf16", max_batch: int = 1, device: int = 0):
        super().__init__()
        self.config = config
        self.rcfg = config if isinstance(config, RolloutConfig) else RolloutConfig.from_namespace(config)
        self.precision = precision
        self._engine: Optional[Engine] = None
        self._engine_args = dict(precision=precision, max_batch=max_batch, device=device)
        self._loaded = False
        self.frame_idx = 0
        self.seed = 0

    # -- engine lifetime -------------------------------------------------------------------------
    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self.rcfg, max_cond_frames=min(20, self.rcfg.max_frame_len), **self._engine_args)
        return self._engine

    def load_state_dict(self, state_dict, strict: bool = False):
        """infer_fun.load_model_paramter (infer_fun.py:43-50) calls this with ckpt["module"], strict=False."""
        want = expected_keys(self.rcfg)
        eng = self.engine
        missing, unexpected = [], []
        for k in want:
            if k not in state_dict:
                missing.append(k)
        for k, v in state_dict.items():
            t = v.detach() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            if t.dtype == torch.bfloat16:
                arr = t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
            else:
                arr = t.cpu().float().contiguous().numpy()
            if not eng.load_tensor(k, arr) and k not in OPTIONAL_KEYS:
                unexpected.append(k)
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing keys {missing[:5]}..., unexpected keys {unexpected[:5]}...")
        if not missing:
            eng.finalize()
            self._loaded = True
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def to(self, *args, **kwargs):      # weights live in the engine's HBM; moving the module is a no-op
        return self

    def cpu(self):
        return self

    def cuda(self, device=None):
        return self

    # -- the hot path ----------------------------------------------------------------------------
    @torch.no_grad()
    def inference(self, new_frames: int, cond_frames: int = 1, input_cond_frames: int = -1, pred_task: str = "image",
                  input_cond_tokens: Optional[Dict[str, torch.Tensor]] = None,
                  init_tokens: Optional[Dict[str, torch.Tensor]] = None, cond_on_tar: bool = False,
                  test_map_affine: bool = False, max_objects=100, control_test=False, **kwargs) -> Dict[str, np.ndarray]:
        """UMGen.inference (UMGen.py:1542-1671).  ``cond_on_par`` / ``infer_from_gt`` are swallowed like the reference."""
        if pred_task != "pose_map_bbox3d_image":
            raise UMGenError(f"pred_task={pred_task!r}: only 'pose_map_bbox3d_image' (the evaluation task) is implemented")
        if not self._loaded:
            raise UMGenError("load_state_dict() has not provided every tensor the rollout reads")
        assert isinstance(input_cond_tokens, dict)
        toks = {m: (input_cond_tokens[m].detach().cpu().numpy() if isinstance(input_cond_tokens[m], torch.Tensor)
                    else np.asarray(input_cond_tokens[m])) for m in MOD_ORDER}
        init = None
        if init_tokens is not None:
            init = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                    for k, v in init_tokens.items() if v is not None and k in ("pose", "bbox3d")}
        B = toks["pose"].shape[0]
        seeds = kwargs.get("seeds", [self.seed + i for i in range(B)])
        return self.engine.rollout(toks, new_frames, cond_frames=cond_frames, input_cond_frames=input_cond_frames,
                                   init_tokens=init, control_test=bool(control_test), seeds=seeds)
