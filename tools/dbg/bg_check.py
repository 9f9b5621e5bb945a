"""The overlapped pass on the decode engine's idle XCDs (bg_worker.h; default for one scene per GPU) against UMGEN_BG_ENGINE=0 (every window in the
foreground, engine on 8 XCD groups): tokens must be equal bit for bit.  CFG=full_width|deep|large  FRAMES=n  T=history  PREC=bf16|fp16  TASK=video|control|mapgiven"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from umgen_amd.config import MOD_ORDER, large_config  # noqa: E402
from umgen_amd.engine import Engine  # noqa: E402
from umgen_amd.synth import synthetic_control, synthetic_given_map, synthetic_scene  # noqa: E402
from umgen_amd.weights import expected_keys, synth_tensor  # noqa: E402

name = os.environ.get("CFG", "full_width")
frames = int(os.environ.get("FRAMES", "3"))
T = int(os.environ.get("T", "4"))
prec = os.environ.get("PREC", "bf16")
task = os.environ.get("TASK", "video")
if name == "large":
    cfg = large_config()
else:
    from tests.golden.make_full_width_golden import config as width_config
    cfg = width_config(name)
scene = synthetic_scene(11, n_frames=T)
T_in = T if task != "control" else max(2, T - 2)
extra = {}
if task == "control":
    c = synthetic_control(11, n_frames=frames)
    extra = {"init_tokens": {k: c[k] for k in ("pose", "bbox3d")}, "control_test": True}
elif task == "mapgiven":
    extra = {"init_tokens": {"map": synthetic_given_map(11, n_frames=frames)["map"]}}
outs = {}
only = os.environ.get("ONLY")
for bg in (("0", "1") if not only else (only,)):
    os.environ["UMGEN_BG_ENGINE"] = bg
    e = Engine(cfg, precision=prec, max_batch=1, max_cond_frames=T)
    for key, shape in expected_keys(cfg).items():
        e.load_tensor(key, synth_tensor(key, shape, seed=0))
    e.finalize()
    t0 = time.perf_counter()
    outs[bg] = e.rollout(scene, frames, cond_frames=T, input_cond_frames=T_in, seeds=[5], **extra)
    dt = time.perf_counter() - t0
    tm = e.timings()
    print(f"bg={bg}: {dt / frames * 1e3:.1f} ms/frame  ego {tm['ego_ms'] / frames:.1f} tar {tm['tar_ms'] / frames:.1f} oar {tm['oar_ms'] / frames:.1f} ms  overlapped_frames {tm['overlapped_frames']} "
          f"drain {tm['bg_ms']:.2f} ms  engine {tm['decode_engine']}", flush=True)
    e.close()
if only:
    print("ran", only)
    sys.exit(0)
bad = 0
for m in MOD_ORDER:
    d = int((outs["0"][m] != outs["1"][m]).sum())
    bad += d
    if d:
        fr = sorted(set(np.argwhere(outs["0"][m] != outs["1"][m])[:, 1].tolist()))
        print(f"{m}: {d} tokens differ, frames {fr}")
print("EQUAL" if bad == 0 else f"DIFFERENT ({bad} tokens)")
sys.exit(0 if bad == 0 else 1)
