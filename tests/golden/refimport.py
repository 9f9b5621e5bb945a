"""Import harness for the UPSTREAM reference (``/root/reference``) -- build-container only.

This module is test infrastructure.  It makes ``projects.models.UMGen`` importable on a
CPU-only box by injecting inert stand-ins for third-party packages the image lacks
(mmcv, flash_attn, deepspeed, numba, torchmetrics, cv2, torchvision) and by neutralising
the hard-coded ``.cuda()`` / CUDA-autocast calls of the reference (UMGen.py:662,720,1047,1605).
Nothing from the reference is copied: the stand-ins below are written from the call
signatures the reference uses.

The flash-attn stand-in defines the semantics pinned at that third-party boundary
(flash-attn==2.3.2, requirements.txt:8): exact softmax attention, scale passed by the
caller, bottom-right aligned causal mask when Tq != Tk, fp32 math, contiguous output.

Only ``tests/golden/make_golden.py`` and ``tests/test_oracle_vs_reference.py`` use this, and
both skip when ``/root/reference`` is absent (e.g. on the GPU box).
"""
from __future__ import annotations

import contextlib
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("UMGEN_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "projects", "models"))


def _flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False):
    # q [B,Tq,H,D]; k,v [B,Tk,H,D]
    B, Tq, H, D = q.shape
    Tk = k.shape[1]
    scale = float(softmax_scale) if softmax_scale is not None else D ** -0.5
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3)
    vf = v.float().permute(0, 2, 1, 3)
    att = (qf @ kf.transpose(-1, -2)) * scale
    if causal:
        i = torch.arange(Tq).view(-1, 1)
        j = torch.arange(Tk).view(1, -1)
        att = att.masked_fill(j > i + (Tk - Tq), float("-inf"))
    att = torch.softmax(att, dim=-1)
    y = att @ vf
    return y.permute(0, 2, 1, 3).contiguous().to(q.dtype)


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.module_dict[cls.__name__] = cls
            return cls
        return deco


def _identity_jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False
AUTOCAST_DTYPE = None      # None: the reference's autocast region is a no-op (fp32); torch.float16 / torch.bfloat16: CPU autocast in that type


def set_autocast(dtype):
    """None | torch.float16 | torch.bfloat16: what the reference's `torch.cuda.amp.autocast()` region does from now on."""
    global AUTOCAST_DTYPE
    AUTOCAST_DTYPE = dtype



def install_stubs():
    global _installed
    if _installed:
        return
    _installed = True
    mmcv_utils = _mod("mmcv.utils", Registry=_Registry)
    _mod("mmcv", utils=mmcv_utils, imread=None, Config=None)
    _mod("flash_attn", flash_attn_func=_flash_attn_func)
    ckpt = _mod("deepspeed.checkpointing", is_configured=lambda: False)
    _mod("deepspeed", checkpointing=ckpt)
    nb_dec = _mod("numba.cuda.decorators", jit=_identity_jit)
    nb_cuda = _mod("numba.cuda", decorators=nb_dec)
    _mod("numba", jit=_identity_jit, cuda=nb_cuda)

    class Metric:  # torchmetrics.Metric stand-in (never instantiated on the path)
        def __init__(self, *a, **k):
            pass

    _mod("torchmetrics", Metric=Metric)
    _mod("cv2")

    class _T:
        class Compose:      # torchvision.transforms.Compose: apply the transforms in order
            def __init__(self, ts):
                self.ts = ts

            def __call__(self, x):
                for t in self.ts:
                    x = t(x)
                return x

        class ToTensor:
            def __call__(self, x):
                return x

    tv_t = _mod("torchvision.transforms", Compose=_T.Compose, ToTensor=_T.ToTensor)
    _mod("torchvision", transforms=tv_t)

    # the reference hard-codes .cuda() and CUDA autocast (UMGen.py:1604-1605); on this CPU box .cuda() is a no-op and the autocast region is
    # either off (AUTOCAST_DTYPE None: the fp32 goldens) or torch's CPU autocast in the selected 16-bit type (set_autocast: the goldens that pin
    # the 16-bit modes on the reference's own arithmetic -- linear layers in the 16-bit type with 16-bit outputs, fp32 residual stream by type
    # promotion, the reference's LayerNorm in fp32, attention through the flash-attn stand-in: 16-bit q / k / v in, fp32 math, 16-bit out)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.amp.autocast = lambda *a, **k: (contextlib.nullcontext() if AUTOCAST_DTYPE is None else torch.autocast("cpu", dtype=AUTOCAST_DTYPE))
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


@contextlib.contextmanager
def reference_cwd():
    """Codebook / category paths in the reference config are cwd-relative (config.py:103,366)."""
    old = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    try:
        yield
    finally:
        os.chdir(old)


def import_reference():
    """Returns (UMGen class, reference config module)."""
    install_stubs()
    with reference_cwd():
        import projects.configs.UMGen_config_evaluation as cfg  # noqa
        from projects.models.UMGen import UMGen  # noqa
    return UMGen, cfg


def build_reference_model(cfg, state_dict, greedy=True):
    """Instantiate the reference ``UMGen`` at the width/depth of ``cfg`` (umgen_amd.config.RolloutConfig),
    load ``state_dict`` (key -> ndarray) into it and, for ``greedy``, set all three k's to 1 (SURVEY a-15)."""
    import copy

    import numpy as np

    UMGen, refcfg = import_reference()
    mc = copy.copy(refcfg.model_config)
    for k in ("n_embd", "n_head", "n_ego_tar_layer", "n_ego_ca_layer", "n_map_tar_layer", "n_box_tar_layer",
              "n_tar_layer", "n_oar_layer", "max_frame_len", "rule_constrain"):
        setattr(mc, k, getattr(cfg, k))
    # what infer_fun.set_model_config (infer_fun.py:84-159) resolves for evaluate.py
    mc.device_set = torch.device("cpu")
    mc.dropout = 0
    mc.sample_method = cfg.sample_method
    mc.top_k = 1 if greedy else cfg.top_k
    mc.top_k_map = 1 if greedy else cfg.top_k_map
    mc.p = cfg.p
    mc.sample_img = True
    mc.pad_to_length = 60
    mc.num_attritube = 10
    with reference_cwd():
        model = UMGen(mc)
    if greedy:
        model.topk_image = 1
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state_dict.items()}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    model.eval()
    return model
