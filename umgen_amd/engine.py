"""Python handle of the HIP rollout engine (thin wrapper over the C ABI; numpy in, numpy out)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .config import CONTENT_LEN, MOD_ORDER, SEQ_LEN, RolloutConfig


class UMGenError(RuntimeError):
    pass


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int64)


def _p64(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.POINTER(C.c_int64)) if a is not None else None


class Engine:
    """One engine = one GPU, one HIP stream.  ``precision``: "bf16" (production), "fp16" (the same kernels with IEEE-half operands:
    the reference's own autocast arithmetic) or "fp32" (exact parity mode)."""

    def __init__(self, cfg: RolloutConfig, precision: str = "bf16", max_batch: int = 1, max_cond_frames: int = 20,
                 device: int = 0, use_graphs: bool = True):
        self.lib = _lib.load_library()
        self.cfg = cfg
        max_cond_frames = min(max_cond_frames, cfg.max_frame_len)
        self.precision = precision
        self.max_batch = max_batch
        c = _lib.Config(
            abi_version=4, n_embd=cfg.n_embd, n_head=cfg.n_head, n_ego_tar_layer=cfg.n_ego_tar_layer,
            n_ego_ca_layer=cfg.n_ego_ca_layer, n_map_tar_layer=cfg.n_map_tar_layer, n_box_tar_layer=cfg.n_box_tar_layer,
            n_tar_layer=cfg.n_tar_layer, n_oar_layer=cfg.n_oar_layer, pose_vocab=cfg.pose_vocab_size,
            map_vocab=cfg.map_vocab_size, bbox3d_vocab=cfg.bbox3d_vocab_size, img_vocab=cfg.img_vocab_size,
            aux_vocab=cfg.aux_vocab_size, n_map_embd=cfg.n_map_embd, n_img_embd=cfg.n_img_embd,
            max_frame_len=cfg.max_frame_len, task_num=cfg.task_num, task_id=cfg.task_id,
            precision={"fp32": _lib.PREC_FP32, "bf16": _lib.PREC_BF16, "fp16": _lib.PREC_FP16}[precision], max_batch=max_batch,
            max_cond_frames=max_cond_frames, device=device, use_graphs=int(use_graphs))
        self._h = C.c_void_p()
        rc = self.lib.umgen_create(C.byref(c), C.byref(self._h))
        if rc != 0:
            msg = self.lib.umgen_last_error(self._h).decode() if self._h else "umgen_create failed"
            if self._h:
                self.lib.umgen_destroy(self._h)
                self._h = None
            raise UMGenError(f"umgen_create: {msg} (rc={rc})")

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.umgen_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc < 0:
            raise UMGenError(f"{what}: {self.lib.umgen_last_error(self._h).decode()} (rc={rc})")
        return rc

    # -- weights -------------------------------------------------------------------------------
    def load_tensor(self, key: str, arr) -> bool:
        """Returns False when the key is not consumed by the rollout (ignored, like strict=False)."""
        a = np.asarray(arr)
        if a.dtype == np.float32:
            dt = _lib.DT_F32
        elif a.dtype == np.float64:
            dt = _lib.DT_F64
        elif a.dtype == np.float16:
            dt = _lib.DT_F16
        elif a.dtype == np.uint16:      # raw bfloat16 bits
            dt = _lib.DT_BF16
        else:
            a = a.astype(np.float32)
            dt = _lib.DT_F32
        a = np.ascontiguousarray(a)
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        rc = self._check(self.lib.umgen_load_tensor(self._h, key.encode(), a.ctypes.data_as(C.c_void_p), dt, shape, a.ndim),
                         f"load_tensor({key})")
        return rc == 0

    def load_state_dict(self, items: Iterable[Tuple[str, np.ndarray]]) -> int:
        n = 0
        for k, v in (items.items() if hasattr(items, "items") else items):
            n += bool(self.load_tensor(k, v))
        return n

    def finalize(self):
        self._check(self.lib.umgen_finalize_weights(self._h), "finalize_weights")

    # -- rollout -------------------------------------------------------------------------------
    def _sampling(self, cfg: RolloutConfig, seeds: Sequence[int]):
        s = np.asarray(list(seeds), dtype=np.uint64)
        smp = _lib.Sampling(
            method=0 if cfg.sample_method == "topk" else 1, top_k=cfg.top_k, top_k_map=cfg.top_k_map,
            topk_image=cfg.topk_image, p=cfg.p, p_map=cfg.p_map, temperature=cfg.sfmx_temp,
            rule_constrain=int(cfg.rule_constrain), merge_ar_tar=int(cfg.merage_ar_tar), only_ar=int(cfg.only_ar),
            seeds=s.ctypes.data_as(C.POINTER(C.c_uint64)))
        return smp, s

    def rollout(self, tokens: Dict[str, np.ndarray], new_frames: int, cond_frames: int = 20,
                input_cond_frames: int = -1, init_tokens: Optional[Dict[str, np.ndarray]] = None,
                control_test: bool = False, seeds: Optional[Sequence[int]] = None,
                sampling: Optional[RolloutConfig] = None) -> Dict[str, np.ndarray]:
        """tokens: mod -> int64 [B, T, S_mod].  Returns mod -> int64 [B, input_cond_frames + new_frames, S_mod]."""
        if input_cond_frames == -1:
            input_cond_frames = cond_frames
        arrs = {m: _i64(tokens[m])[:, :input_cond_frames] for m in MOD_ORDER}
        arrs = {m: np.ascontiguousarray(a) for m, a in arrs.items()}
        B, T_in = arrs["pose"].shape[:2]
        for m in MOD_ORDER:
            if arrs[m].shape != (B, T_in, CONTENT_LEN[m]):
                raise UMGenError(f"tokens[{m}] has shape {arrs[m].shape}, expected {(B, T_in, CONTENT_LEN[m])}")
        outs = {m: np.empty((B, T_in + new_frames, CONTENT_LEN[m]), dtype=np.int64) for m in MOD_ORDER}
        cp = cb = gm = gb = None
        T_ctl = 0
        if init_tokens is not None:
            # init_tokens["image"] is dropped by the reference before the decode loop ("to avoid image token as init tokens",
            # UMGen.py:1512-1520) but still copied into the output (1640-1651): not a generation path -- refused rather than imitated
            extra = [k for k, v in init_tokens.items() if v is not None and k not in ("pose", "bbox3d", "map")]
            if extra:
                raise UMGenError(f"init_tokens for {extra} are not supported (pose / bbox3d control tokens, given map / map + bbox3d tokens)")
        if init_tokens is not None and init_tokens.get("pose") is not None:
            cp = _i64(init_tokens["pose"])
            if cp.ndim != 3 or cp.shape[0] != B or cp.shape[2] != CONTENT_LEN["pose"]:
                raise UMGenError(f"init_tokens['pose'] has shape {cp.shape}, expected ({B}, T_ctl, {CONTENT_LEN['pose']})")
            T_ctl = cp.shape[1]
        if init_tokens is not None and init_tokens.get("map") is not None:      # the map of every new frame is GIVEN (UMGen.py:1184-1201)
            gm = _i64(init_tokens["map"])
            if gm.ndim != 3 or gm.shape[0] != B or gm.shape[2] != CONTENT_LEN["map"] or (T_ctl and gm.shape[1] != T_ctl):
                raise UMGenError(f"init_tokens['map'] has shape {gm.shape}, expected ({B}, {T_ctl or 'T'}, {CONTENT_LEN['map']})")
            T_ctl = gm.shape[1]
        if init_tokens is not None and init_tokens.get("bbox3d") is not None and not control_test and gm is not None:
            gb = _i64(init_tokens["bbox3d"])                                     # boxes given too (behind a given map)
            if gb.shape != (B, T_ctl, CONTENT_LEN["bbox3d"]):
                raise UMGenError(f"init_tokens['bbox3d'] has shape {gb.shape}, expected {(B, T_ctl, CONTENT_LEN['bbox3d'])}")
        elif init_tokens is not None and init_tokens.get("bbox3d") is not None:   # with or without pose tokens (UMGen.py:1458-1473)
            cb = _i64(init_tokens["bbox3d"])
            if cp is None:
                # (umgen_rollout takes ONE T_ctl for every control / given array: a given map and control boxes must cover the same frames)
                if cb.ndim != 3 or cb.shape[0] != B or cb.shape[2] != CONTENT_LEN["bbox3d"] or (T_ctl and cb.shape[1] != T_ctl):
                    raise UMGenError(f"init_tokens['bbox3d'] has shape {cb.shape}, expected ({B}, {T_ctl or 'T_ctl'}, {CONTENT_LEN['bbox3d']})")
                T_ctl = cb.shape[1]
            elif cb.shape != (B, T_ctl, CONTENT_LEN["bbox3d"]):
                raise UMGenError(f"init_tokens['bbox3d'] has shape {cb.shape}, expected {(B, T_ctl, CONTENT_LEN['bbox3d'])}")
        smp, keep = self._sampling(sampling or self.cfg, seeds if seeds is not None else [0] * B)
        self._check(self.lib.umgen_rollout(
            self._h, B, T_in, new_frames, cond_frames, _p64(arrs["pose"]), _p64(arrs["map"]), _p64(arrs["bbox3d"]),
            _p64(arrs["image"]), T_ctl, _p64(cp), _p64(cb), int(control_test), _p64(gm), _p64(gb), C.byref(smp),
            _p64(outs["pose"]), _p64(outs["map"]), _p64(outs["bbox3d"]), _p64(outs["image"])), "rollout")
        del keep
        return outs

    def frame(self, window: Dict[str, np.ndarray], frame_idx: int = 0, ctrl: Optional[Dict[str, np.ndarray]] = None,
              control_test: bool = False, seed: int = 0, sampling: Optional[RolloutConfig] = None,
              forced: Optional[Dict[str, np.ndarray]] = None, trace: bool = False, given: Optional[Dict[str, np.ndarray]] = None):
        """One frame of one scene.  window: mod -> [T, S_mod].  Returns (tokens dict, trace dict or None).
        given: {"map": [1024]} or {"map": ..., "bbox3d": [660]}: the frame's GIVEN tokens (init_tokens of `rollout` for one frame)."""
        cfg = self.cfg
        w = {m: _i64(window[m]) for m in MOD_ORDER}
        T = w["pose"].shape[0]
        outs = {m: np.empty((CONTENT_LEN[m],), dtype=np.int64) for m in MOD_ORDER}
        cp = _i64(ctrl["pose"]) if ctrl is not None and ctrl.get("pose") is not None else None
        cb = _i64(ctrl["bbox3d"]) if ctrl is not None and ctrl.get("bbox3d") is not None else None
        smp, keep = self._sampling(sampling or cfg, [seed])
        tr = None
        tbuf = None
        fz = None
        gz = None
        if trace or forced is not None or given is not None:
            tr = _lib.Trace()
            if given is not None:
                gz = {m: _i64(given[m]).reshape(-1) for m in given}
                tr.given_map = _p64(gz["map"]) if "map" in gz else None
                tr.given_bbox3d = _p64(gz["bbox3d"]) if "bbox3d" in gz else None
            counters = np.zeros(8, np.int32)
            tr.counters = counters.ctypes.data_as(C.POINTER(C.c_int32))
            fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
            if trace:
                tbuf = {"cond": np.zeros((SEQ_LEN, cfg.n_embd), np.float32),
                        "ego_logits": np.zeros((3, cfg.pose_vocab_size), np.float32),
                        "logits_map": np.zeros((1024, cfg.map_vocab_size), np.float32),
                        "logits_bbox3d": np.zeros((660, cfg.bbox3d_vocab_size), np.float32),
                        "logits_image": np.zeros((512, cfg.img_vocab_size), np.float32)}
                tr.cond, tr.ego_logits = fp(tbuf["cond"]), fp(tbuf["ego_logits"])
                tr.logits_map, tr.logits_bbox3d, tr.logits_image = fp(tbuf["logits_map"]), fp(tbuf["logits_bbox3d"]), fp(tbuf["logits_image"])
            if forced is not None:
                fz = {m: _i64(forced[m]).reshape(-1) for m in MOD_ORDER}
                tr.forced_pose, tr.forced_map = _p64(fz["pose"]), _p64(fz["map"])
                tr.forced_bbox3d, tr.forced_image = _p64(fz["bbox3d"]), _p64(fz["image"])
        self._check(self.lib.umgen_frame(
            self._h, T, _p64(w["pose"]), _p64(w["map"]), _p64(w["bbox3d"]), _p64(w["image"]), _p64(cp), _p64(cb),
            int(control_test), C.byref(smp), frame_idx, C.byref(tr) if tr is not None else None,
            _p64(outs["pose"]), _p64(outs["map"]), _p64(outs["bbox3d"]), _p64(outs["image"])), "frame")
        del keep, fz, gz
        if tr is not None:
            tbuf = tbuf if tbuf is not None else {}
            tbuf["counters"] = dict(zip(("pad_avoid", "control_resample", "rule_checked", "rule_collision", "rule_blanked",
                                         "sampled_ne_forced"), counters[:6].tolist()))
        return outs, tbuf

    def dbg_oar_step(self, x: np.ndarray, L: int, use_engine: bool, unmasked: bool = True) -> np.ndarray:
        """Test hook: one decode step through the BlockOAR layers for x [B, n_embd] at KV length L (appends K/V row L)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        self._check(self.lib.umgen_dbg_oar_step(self._h, x.shape[0], L, fp(x), fp(out), int(use_engine), int(unmasked)), "dbg_oar_step")
        return out

    # -- measurement ---------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._check(self.lib.umgen_set_profiling(self._h, int(on)), "set_profiling")

    def timings(self) -> dict:
        t = _lib.Timings()
        self._check(self.lib.umgen_get_timings(self._h, C.byref(t)), "get_timings")
        return {n: getattr(t, n) for n, _ in _lib.Timings._fields_}
