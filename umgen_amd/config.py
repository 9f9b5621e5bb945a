"""Resolved hyper-parameters of the UMGen next-scene rollout (hot path only).

Values and names follow the reference's resolved configuration:
  projects/configs/UMGen_config_evaluation.py:27-38,65-70,84-91,126-137,271-290,344-430
  projects/tools/infer_fun.py:84-159 (set_model_config), :56-81 (set_inference_setting)
  projects/models/UMGen.py:99-105 (the three separate top-k's; topk_image is hard-coded 16)

``RolloutConfig`` is what the C-ABI ``umgen_config`` struct (include/umgen.h) is filled from and
what the CPU oracle (oracle/umgen_oracle.py) consumes, so the three can never disagree.
"""
from __future__ import annotations

import dataclasses
from argparse import Namespace
from typing import Dict, List, Tuple

# scene-sequence layout of one frame (infer_fun.py:112-118, UMGen.py:976-992)
MOD_ORDER: Tuple[str, ...] = ("pose", "map", "bbox3d", "image")
CONTENT_LEN: Dict[str, int] = {"pose": 3, "map": 1024, "bbox3d": 660, "image": 512}
TOKEN_LEN: Dict[str, int] = {m: n + 2 for m, n in CONTENT_LEN.items()}  # + bos/eos
SEQ_LEN = sum(TOKEN_LEN.values())  # 2207
BOS_EOS: Dict[str, Tuple[int, int]] = {"pose": (0, 1), "map": (2, 3), "bbox3d": (4, 5), "image": (6, 7)}
# first scene position (0-based) of each modality's bos
MOD_START: Dict[str, int] = {}
_p = 0
for _m in MOD_ORDER:
    MOD_START[_m] = _p
    _p += TOKEN_LEN[_m]
del _p, _m

N_SLOTS = 60          # config.py:186 pad_to_length
SLOT_LEN = 11         # 10 attributes + category
BBOX_PAD = 1027       # BBox3DTokenizer.pad_token = 1024 bins + 3 categories
EGO_BOX = (0.0, 0.0, 0.0, 5.176, 2.297, 1.777, 0.0, 0.0, 0.0, 0.0)  # UMGen.py:9-12,1319-1336
# min-max ranges of the 10 box attributes (config.py:126-137), in attribute order
BBOX_RANGE: Tuple[Tuple[float, float], ...] = (
    (-64, 64), (-64, 64), (-5, 5), (0, 15), (0, 4), (0, 5), (-3.14, 3.14), (-20, 20), (-15, 15), (-0.3, 0.3),
)
EGO_STD = (10.0, 4.0, 1.0)  # config.py:223-231


@dataclasses.dataclass
class RolloutConfig:
    # width
    n_embd: int = 768
    n_head: int = 16
    # depth of each stack (config.py:27-35; "larger" => n_tar_layer 36, infer_fun.py:144-146)
    n_ego_tar_layer: int = 12
    n_ego_ca_layer: int = 12
    n_map_tar_layer: int = 24
    n_box_tar_layer: int = 24
    n_tar_layer: int = 36
    n_oar_layer: int = 36
    # vocabularies (config.py:65-70, 277)
    pose_vocab_size: int = 1024
    map_vocab_size: int = 8192
    bbox3d_vocab_size: int = 1028
    img_vocab_size: int = 8192
    aux_vocab_size: int = 8
    n_map_embd: int = 16
    n_img_embd: int = 16
    max_frame_len: int = 100
    task_num: int = 7
    task_id: int = 6            # task_name_id["pose_map_bbox3d_image"]
    # sampling (config.py:86-91,449; UMGen.py:103)
    sample_method: str = "topk"
    top_k: int = 5
    top_k_map: int = 5
    topk_image: int = 16
    p: float = 0.4
    p_map: float = 0.4
    sfmx_temp: float = 1.0
    # rollout switches (config.py:7-21; evaluate.py:59-63)
    rule_constrain: bool = True
    merage_ar_tar: bool = True
    only_ar: bool = False
    no_born: bool = False

    @property
    def head_dim(self) -> int:
        return self.n_embd // self.n_head

    @property
    def seq_len(self) -> int:
        return SEQ_LEN

    def greedy(self) -> "RolloutConfig":
        """Full greedy needs all three k's = 1 (SURVEY a-15)."""
        return dataclasses.replace(self, top_k=1, top_k_map=1, topk_image=1, sample_method="topk")

    # ---- construction from / to the reference's Namespace config (UMGen.__init__, UMGen.py:53-172)
    @classmethod
    def from_namespace(cls, ns) -> "RolloutConfig":
        g = lambda k, d: getattr(ns, k, d)  # noqa: E731
        task_id = g("task_name_id", {"pose_map_bbox3d_image": 6}).get("pose_map_bbox3d_image", 6)
        c = cls(
            n_embd=ns.n_embd, n_head=ns.n_head,
            n_ego_tar_layer=ns.n_ego_tar_layer, n_ego_ca_layer=ns.n_ego_ca_layer,
            n_map_tar_layer=ns.n_map_tar_layer, n_box_tar_layer=ns.n_box_tar_layer,
            n_tar_layer=ns.n_tar_layer, n_oar_layer=ns.n_oar_layer,
            pose_vocab_size=ns.pose_vocab_size, map_vocab_size=ns.map_vocab_size,
            bbox3d_vocab_size=ns.bbox3d_vocab_size, img_vocab_size=ns.img_vocab_size,
            aux_vocab_size=ns.aux_vocab_size, n_map_embd=ns.n_map_embd, n_img_embd=ns.n_img_embd,
            max_frame_len=ns.max_frame_len, task_num=ns.task_num, task_id=task_id,
            sample_method=ns.sample_method, top_k=ns.top_k, top_k_map=g("top_k_map", ns.top_k),
            p=ns.p, p_map=g("p_map", ns.p), sfmx_temp=ns.sfmx_temp,
            rule_constrain=bool(g("rule_constrain", False)), merage_ar_tar=bool(g("merage_ar_tar", True)),
            only_ar=bool(g("only_ar", False)), no_born=bool(g("no_born", False)),
        )
        unsupported = []
        if not g("split_map_tar", True) or not g("split_box_tar", True):
            unsupported.append("split_map_tar/split_box_tar=False")
        if not g("map_transform", True):
            unsupported.append("map_transform=False")
        if g("box_transform", False):
            unsupported.append("box_transform=True")
        if g("n_step", 1) != 1:
            unsupported.append("n_step!=1")
        if not g("sample_img", True):
            unsupported.append("sample_img=False")
        if g("bias", False):
            unsupported.append("bias=True")
        if c.no_born:
            # [probe, round 3] the reference's own no_born branch (UMGen.py:1106-1114) cannot run: without control objects it reads the
            # unassigned `object_id` (UnboundLocalError, line 1109); with them its 0-dim pad token breaks the torch.cat at line 1238
            # ("Tensors must have same number of dimensions: got 4 and 1").  There is no behaviour to reproduce, so it stays refused.
            unsupported.append("no_born=True (the reference itself raises on this branch)")
        # switches whose non-default value changes what the engine hard-codes (UMGen.py:99-172): refuse them instead of
        # silently producing other tokens
        if g("split_image_ar", False):
            unsupported.append("split_image_ar=True")
        if not g("add_posi_embedd", True):
            unsupported.append("add_posi_embedd=False")
        if not g("add_spatial_pos_embedd_on_map", True):
            unsupported.append("add_spatial_pos_embedd_on_map=False")
        if g("pred_task", "pose_map_bbox3d_image") != "pose_map_bbox3d_image":
            unsupported.append(f"pred_task={g('pred_task', None)!r}")
        if g("seq_len", SEQ_LEN) != SEQ_LEN:
            unsupported.append(f"seq_len={g('seq_len', None)} (the scene sequence is {SEQ_LEN} positions)")
        tl = g("token_len", None)
        if tl is not None and dict(tl) != TOKEN_LEN:
            unsupported.append(f"token_len={dict(tl)}")
        be = g("bos_eos", None)
        if be is not None and {k: tuple(v) for k, v in dict(be).items()} != BOS_EOS:
            unsupported.append("bos/eos ids differ from infer_fun.py:99-104")
        if unsupported:
            raise NotImplementedError(
                "umgen_amd implements the UMGen_Large evaluation configuration only; unsupported: "
                + ", ".join(unsupported))
        return c


def tiny_config(**over) -> RolloutConfig:
    """Small-width/depth config used for oracle fixtures and parity tests (S stays 2207).

    head_dim is kept at the production value 48 so the same attention kernels are exercised.
    """
    base = dict(n_embd=96, n_head=2, n_ego_tar_layer=1, n_ego_ca_layer=1, n_map_tar_layer=1,
                n_box_tar_layer=1, n_tar_layer=1, n_oar_layer=2, max_frame_len=8)
    base.update(over)
    return RolloutConfig(**base)


def large_config(**over) -> RolloutConfig:
    return RolloutConfig(**over)


def wide2x_config(**over) -> RolloutConfig:
    """BASELINE.json config #5: synthetic 2x-width UMGen (E=1536, H=32 keeps head_dim 48)."""
    base = dict(n_embd=1536, n_head=32)
    base.update(over)
    return RolloutConfig(**base)
