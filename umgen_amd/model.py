"""Drop-in ``UMGen`` model class for the reference's callers (evaluate.py:193-220, model_pl.py:173-175, 237-239).

Same surface as projects/models/UMGen.py: ``UMGen(config)`` with the reference's argparse.Namespace config, an
``nn.Module`` (so Lightning can own/move it), ``load_state_dict(sd, strict=False)`` with the reference's key names,
``eval()``, ``.cpu()/.to()``, and ``inference(**kwargs) -> Dict[str, np.ndarray int64 [1, T_out, S_mod]]``.
Internally nothing is a torch op: tokens go to the HIP engine through the C ABI (include/umgen.h).  There is no CPU
fallback -- constructing the engine without a GPU / without libumgen_hip.so raises.
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from .config import MOD_ORDER, RolloutConfig
from .engine import Engine, UMGenError
from .registry import MODELS
from .weights import OPTIONAL_KEYS, expected_keys, is_matrix_weight


@MODELS.register_module()
class UMGen(nn.Module):
    def __init__(self, config, precision: str = "bf16", max_batch: int = 1, device: Optional[int] = None):
        super().__init__()
        self.config = config
        self.rcfg = config if isinstance(config, RolloutConfig) else RolloutConfig.from_namespace(config)
        self.precision = precision
        self._engine: Optional[Engine] = None
        # one process per GPU: the harness (Lightning / torchrun) tells the rank its device through LOCAL_RANK and .to(device)
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self._engine_args = dict(precision=precision, max_batch=max_batch, device=int(device))
        self._state: Optional[Dict[str, np.ndarray]] = None      # host copy of the weights: lets the engine move / grow
        self._loaded = False
        self.frame_idx = 0
        self.seed = 0

    # -- engine lifetime -------------------------------------------------------------------------
    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self.rcfg, max_cond_frames=min(20, self.rcfg.max_frame_len), **self._engine_args)
            if self._state is not None:      # re-created on another device / with a larger batch: reload the weights
                for k, arr in self._state.items():
                    if not self._engine.load_tensor(k, arr) and k not in OPTIONAL_KEYS:
                        raise UMGenError(f"re-created engine refused state-dict entry {k!r}")
                if self._loaded:             # (a partial load_state_dict was never finalized: nothing to finalize now either)
                    self._engine.finalize()
        return self._engine

    def _host_copy(self, key: str, arr: np.ndarray) -> np.ndarray:
        """Host copy kept so that the engine can move to another GPU / grow its scene batch: nn.Linear weights in the engine's own
        16-bit storage type (what umgen_load_tensor rounds them to anyway: half the host memory), everything else as given."""
        if arr.dtype == np.float32 and is_matrix_weight(key):
            if self.precision == "bf16":
                return torch.from_numpy(arr).bfloat16().view(torch.int16).numpy().view(np.uint16)
            if self.precision == "fp16":
                return arr.astype(np.float16)
        return arr

    def _recreate(self, **changes):
        """The weights live in the engine's HBM: a new device or a larger scene batch means a new engine."""
        new = dict(self._engine_args, **changes)
        if new == self._engine_args:
            return
        if self._engine is not None:
            warnings.warn(f"umgen_amd.UMGen: engine re-created ({self._engine_args} -> {new}); the weights are uploaded again", stacklevel=3)
            self._engine.close()
            self._engine = None
        self._engine_args = new

    def load_state_dict(self, state_dict, strict: bool = False):
        """infer_fun.load_model_paramter (infer_fun.py:43-50) calls this with ckpt["module"], strict=False."""
        want = expected_keys(self.rcfg)
        eng = self.engine
        missing, unexpected = [], []
        for k in want:
            if k not in state_dict:
                missing.append(k)
        kept: Dict[str, np.ndarray] = {}
        for k, v in state_dict.items():
            t = v.detach() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            if t.dtype == torch.bfloat16:
                arr = t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
            else:
                arr = t.cpu().float().contiguous().numpy()
            if not eng.load_tensor(k, arr) and k not in OPTIONAL_KEYS:
                unexpected.append(k)
            elif k in want or k in OPTIONAL_KEYS:
                kept[k] = self._host_copy(k, arr)
        self._state = kept
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing keys {missing[:5]}..., unexpected keys {unexpected[:5]}...")
        if not missing:
            eng.finalize()
            self._loaded = True
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # The parameters live in the engine's HBM, not in nn.Parameters.  `.to(device)` / `.cuda(i)` (model_pl.py:366-368, Lightning's
    # device placement) select the engine's GPU; `.cpu()` (model_pl.py:445-447, around the VAE decode) keeps it where it is.
    @staticmethod
    def _device_index(dev) -> Optional[int]:
        if dev is None:
            return None
        if isinstance(dev, int):
            return dev
        d = torch.device(dev) if not isinstance(dev, torch.device) else dev
        if d.type != "cuda":
            return None
        return d.index if d.index is not None else int(os.environ.get("LOCAL_RANK", "0"))

    def to(self, *args, **kwargs):
        dev = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device, int)) and not isinstance(a, bool):
                dev = a
        idx = self._device_index(dev)
        if idx is not None:
            self._recreate(device=idx)
        return self

    def cpu(self):
        return self

    def cuda(self, device=None):
        idx = self._device_index(device if device is not None else "cuda")
        if idx is not None:
            self._recreate(device=idx)
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
                    for k, v in init_tokens.items() if v is not None}      # (pose / bbox3d control, given map / map + bbox3d; the engine rejects anything else)
        B = toks["pose"].shape[0]
        if B > self._engine_args["max_batch"]:      # extension over the reference (B = 1): several scenes per call
            self._recreate(max_batch=B)
        seeds = kwargs.get("seeds", [self.seed + i for i in range(B)])
        return self.engine.rollout(toks, new_frames, cond_frames=cond_frames, input_cond_frames=input_cond_frames,
                                   init_tokens=init, control_test=bool(control_test), seeds=seeds)
