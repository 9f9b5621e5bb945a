"""Synthetic ``tokenized_origin_scenes``-shaped inputs.

The dataset hands ``UMGen.inference`` a dict of int64 token tensors (UMGen_nuplan_dataset.py:231-417,
config.py:247-257): pose [1,T,3] in [0,1023], map [1,T,1024] in [0,8191], bbox3d [1,T,660] with 60
slots x (10 attribute bins in [0,1023] + 1 category in {1024,1025,1026}), empty slot = 11 x pad(1027),
image [1,T,512] in [0,8191].  Real scenes are not available offline, so scene ``i`` is drawn from
PCG64(1000 + i).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .config import BBOX_PAD, N_SLOTS, SLOT_LEN


def synthetic_scene(scene_id: int, n_frames: int = 20) -> Dict[str, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(1000 + scene_id))
    T = n_frames
    pose = rng.integers(0, 1024, size=(1, T, 3), dtype=np.int64)
    # keep ego motion plausible: tokens near the centre of each bin range (small dx, dy, dtheta)
    pose = 512 + (pose - 512) // 8
    mp = rng.integers(0, 8192, size=(1, T, 1024), dtype=np.int64)
    img = rng.integers(0, 8192, size=(1, T, 512), dtype=np.int64)
    box = np.full((1, T, N_SLOTS, SLOT_LEN), BBOX_PAD, dtype=np.int64)
    k = int(rng.integers(5, 41))
    slots = rng.permutation(N_SLOTS)[:k]
    attrs = rng.integers(100, 901, size=(k, 10), dtype=np.int64)
    cats = rng.integers(1024, 1027, size=(k,), dtype=np.int64)
    for t in range(T):
        box[0, t, slots, :10] = attrs
        box[0, t, slots, 10] = cats
        attrs = np.clip(attrs + rng.integers(-3, 4, size=attrs.shape), 0, 1023)
    return {"pose": pose, "map": mp, "bbox3d": box.reshape(1, T, N_SLOTS * SLOT_LEN), "image": img}


def synthetic_control(scene_id: int, n_frames: int = 30, slot: Optional[int] = 3) -> Dict[str, np.ndarray]:
    """Control pickle analogue (model_pl.py:137-171): control_dict.pose [1,n,3] random walk,
    control_dict.bbox3d [1,n,660] = -1 (free) except one controlled slot."""
    rng = np.random.Generator(np.random.PCG64(5000 + scene_id))
    pose = np.clip(512 + np.cumsum(rng.integers(-2, 3, size=(1, n_frames, 3)), axis=1), 0, 1023).astype(np.int64)
    box = np.full((1, n_frames, N_SLOTS, SLOT_LEN), -1, dtype=np.int64)
    if slot is not None:
        attrs = rng.integers(300, 700, size=(10,), dtype=np.int64)
        for t in range(n_frames):
            box[0, t, slot, :10] = attrs
            box[0, t, slot, 10] = 1024
            attrs = np.clip(attrs + rng.integers(-3, 4, size=attrs.shape), 0, 1023)
    return {"pose": pose, "bbox3d": box.reshape(1, n_frames, -1)}


def synthetic_given_map(scene_id: int, n_frames: int = 2) -> Dict[str, np.ndarray]:
    """init_tokens that GIVE the map of every new frame (infer_oar_net's predefined-token prefix, UMGen.py:1184-1201): map [1,n,1024]."""
    rng = np.random.Generator(np.random.PCG64(7000 + scene_id))
    return {"map": rng.integers(0, 8192, size=(1, n_frames, 1024), dtype=np.int64)}


def golden_init_tokens(scene_id: int, new_frames: int, control: int) -> Optional[Dict[str, np.ndarray]]:
    """init_tokens of a recorded golden case (tests/golden/make_golden.py, meta[5]): 0 video, 1 pose + bbox3d control for every new
    frame, 2 bbox3d-only control (the ego net infers the pose) for the first two new frames, 3 the map of every new frame given
    (the ego net infers the pose, the decode loop starts behind the map), 4 the map and the boxes given."""
    if control == 3:
        return synthetic_given_map(scene_id, n_frames=new_frames)
    if control == 4:      # the map and the boxes of every new frame given (boxes: the scene generator's layout of another scene id)
        return {"map": synthetic_given_map(scene_id, n_frames=new_frames)["map"], "bbox3d": synthetic_scene(900 + scene_id, n_frames=new_frames)["bbox3d"]}
    if control == 2:
        return {"bbox3d": synthetic_control(scene_id, n_frames=2)["bbox3d"]}
    return synthetic_control(scene_id, n_frames=new_frames) if control else None
