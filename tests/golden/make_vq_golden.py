"""Golden vectors of the VQ decoders, recorded from the REFERENCE's own classes:

    python tests/golden/make_vq_golden.py [small] [full]      ->  tests/golden/vq_<case>.npz

imports /root/reference/projects/tokenizer/vq_model.py (pure torch + einops: no stand-ins needed), builds NormVQModel with the
configuration of each case, loads the build's deterministic synthetic tensors (umgen_amd/vq.py: synth_vq_tensor -- no checkpoint
exists offline) and runs `decode_code` on seeded random codes in fp32 on the CPU.
  small : reduced widths, both post_quant_conv forms (1x1 map-style / 3x3 image-style), attention inside the up path, a
          non-square token grid: whole outputs are stored;
  full  : the two production configurations of vq_model.py:153-202 (image dim16 res512 f16, map dim16 res256 f8), one frame each:
          every 8th pixel of the output + its mean / rms are stored.
Only inputs (codes, seeds) and outputs are committed; the weights are regenerated from the per-key seeds on both sides."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from umgen_amd.vq import IMAGE_VQ, MAP_VQ, decoder_keys, synth_vq_tensor  # noqa: E402

SMALL = {
    "small_map": dict(n_embed=64, embed_dim=16, z_channels=16, resolution=64, out_ch=5, ch=32, ch_mult=(1, 2, 2), num_res_blocks=1,
                      attn_resolutions=(16,), post_quant_ks=1, post_quant_pad=0, token_hw=(16, 16)),
    "small_image": dict(n_embed=64, embed_dim=16, z_channels=32, resolution=64, out_ch=3, ch=32, ch_mult=(1, 1, 2), num_res_blocks=2,
                        attn_resolutions=(16,), post_quant_ks=3, post_quant_pad=1, token_hw=(8, 16)),
}
FULL = {"full_image": IMAGE_VQ, "full_map": MAP_VQ}
SEED = 17


def reference_model(cfg):
    import torch
    sys.path.insert(0, REF)
    from projects.tokenizer.vq_model import NormVQModel
    dd = dict(double_z=False, z_channels=cfg["z_channels"], resolution=cfg["resolution"], in_channels=cfg["out_ch"], out_ch=cfg["out_ch"],
              ch=cfg["ch"], ch_mult=list(cfg["ch_mult"]), num_res_blocks=cfg["num_res_blocks"], attn_resolutions=list(cfg["attn_resolutions"]), dropout=0.0)
    m = NormVQModel(n_embed=cfg["n_embed"], embed_dim=cfg["embed_dim"], ddconfig=dd, stride=cfg["post_quant_ks"], padding=cfg["post_quant_pad"],
                    ckpt_path=None).eval()
    sd = {k: torch.from_numpy(synth_vq_tensor(k, shape, SEED)) for k, shape in decoder_keys(cfg).items()}
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith(("encoder.", "quant_conv.", "quantize.")) for k in res.missing_keys), res.missing_keys
    return m


def main(names):
    import torch
    for name in names:
        cfg = {**SMALL, **FULL}[name]
        n = 2 if name in SMALL else 1
        rng = np.random.default_rng(SEED + len(name))
        codes = rng.integers(0, cfg["n_embed"], size=(n,) + tuple(cfg["token_hw"]), dtype=np.int64)
        m = reference_model(cfg)
        t0 = time.time()
        with torch.no_grad():
            out = m.decode_code(torch.from_numpy(codes)).numpy()
        print(name, out.shape, f"{time.time() - t0:.1f} s", float(out.mean()), float(np.sqrt((out ** 2).mean())))
        rec = {"codes": codes.astype(np.int16), "seed": np.int32(SEED), "mean": np.float64(out.mean()), "rms": np.float64(np.sqrt((out.astype(np.float64) ** 2).mean()))}
        rec["out"] = out.astype(np.float32) if name in SMALL else out[:, :, ::8, ::8].astype(np.float32)
        path = os.path.join(ROOT, "tests", "golden", f"vq_{name}.npz")
        np.savez_compressed(path, **rec)
        print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    sel = sys.argv[1:] or ["small", "full"]
    names = [k for k in SMALL if "small" in sel] + [k for k in FULL if "full" in sel]
    main(names)
