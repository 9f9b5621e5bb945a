"""-m gpu: the VQ decoders (csrc/vqdec.hip through the C ABI) against outputs recorded from the
reference's own NormVQModel.decode_code (tests/golden/make_vq_golden.py) on the same seeded weights and codes.

fp32 throughout (the reference decodes outside autocast); the two sides differ only in fp32 summation order (im2col GEMM vs
torch's convolution) and in expf / GroupNorm-statistics implementations: 2e-4 absolute on outputs of rms ~0.45."""
import os

import numpy as np
import pytest

from tests.golden.make_vq_golden import FULL, SEED, SMALL
from umgen_amd.vq import IMAGE_VQ, Imagedecoder, Mapdecoder, VQDecoder, VQError, decoder_keys, synth_vq_tensor

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make(cfg):
    d = VQDecoder(cfg)
    sd = {k: synth_vq_tensor(k, shape, SEED) for k, shape in decoder_keys(cfg).items()}
    sd["encoder.conv_in.weight"] = np.zeros((4, 4, 3, 3), np.float32)      # present in real checkpoints, not read by the decode path
    missing, unexpected = d.load_state_dict(sd)
    assert not missing and unexpected == ["encoder.conv_in.weight"]
    return d


@pytest.mark.parametrize("name", list(SMALL))
def test_small_decoders_match_reference_golden(name):
    """Reduced widths, both post_quant_conv forms, attention blocks inside the up path, a non-square token grid, two frames."""
    g = np.load(os.path.join(GOLD, f"vq_{name}.npz"))
    d = make(SMALL[name])
    out = d.decode_code(g["codes"].astype(np.int64))
    d.close()
    assert out.shape == g["out"].shape
    np.testing.assert_allclose(out, g["out"], atol=2e-4, rtol=0)


@pytest.mark.parametrize("name", list(FULL))
def test_production_decoders_match_reference_golden(name):
    """vq_model.py:153-202 at full size (image: 16 x 32 tokens -> 3 x 256 x 512; map: 32 x 32 tokens -> 5 x 256 x 256), one frame:
    every 8th pixel and the global statistics against the reference's CPU output."""
    g = np.load(os.path.join(GOLD, f"vq_{name}.npz"))
    d = make(FULL[name])
    out = d.decode_code(g["codes"].astype(np.int64))
    d.close()
    np.testing.assert_allclose(out[:, :, ::8, ::8], g["out"], atol=5e-4, rtol=0)
    assert abs(float(out.mean()) - float(g["mean"])) < 1e-5
    assert abs(float(np.sqrt((out.astype(np.float64) ** 2).mean())) - float(g["rms"])) < 1e-5


def test_wrappers_and_errors(tmp_path):
    """Mapdecoder / Imagedecoder (decode_map.py:110-183) on the rollout's token layout; frames are independent (a batch of 3 equals
    the three single frames: the reference decodes 20 frames per call); malformed inputs fail loudly."""
    cfg = SMALL["small_image"]
    sd = {k: synth_vq_tensor(k, shape, SEED) for k, shape in decoder_keys(cfg).items()}
    dec = Imagedecoder(sd, cfg=cfg)
    toks = np.random.default_rng(3).integers(0, cfg["n_embed"], size=(1, 3, 8 * 16))
    imgs = dec.decode_images(toks, H=8, W=16)
    assert imgs.shape == (3, 3, 32, 64)
    for i in range(3):
        np.testing.assert_array_equal(dec.decode_images(toks[:, i], H=8, W=16)[0], imgs[i])
    with pytest.raises(VQError, match="outside"):
        dec.dec.decode_code(np.full((1, 8, 16), cfg["n_embed"]))
    with pytest.raises(VQError, match="shape"):
        dec.dec.decode_code(np.zeros((1, 8, 8), np.int64))
    bare = VQDecoder(cfg)
    with pytest.raises(VQError, match="finalize"):
        bare.decode_code(np.zeros((1, 8, 16), np.int64))
    bare.close()
    mcfg = SMALL["small_map"]
    mdec = Mapdecoder({k: synth_vq_tensor(k, s, SEED) for k, s in decoder_keys(mcfg).items()}, cfg=mcfg)
    mtok = np.random.default_rng(4).integers(0, mcfg["n_embed"], size=(1, 2, 256))
    rgb = mdec.decode_maps(mtok, H=16, W=16)
    assert rgb.shape == (2, 3, 64, 64) and rgb.min() >= -1.0 - 1e-6 and rgb.max() <= 1.0 + 1e-6
    # a checkpoint path like the reference's Mapdecoder(ckpt) (VQModel.init_from_ckpt: torch.load(path)["state_dict"], encoder keys ignored)
    import torch
    sdm = {k: torch.from_numpy(synth_vq_tensor(k, s, SEED)) for k, s in decoder_keys(mcfg).items()}
    sdm["encoder.conv_in.weight"] = torch.zeros(4, 3, 3, 3)
    torch.save({"state_dict": sdm}, tmp_path / "map_vq.pt")
    np.testing.assert_array_equal(Mapdecoder(str(tmp_path / "map_vq.pt"), cfg=mcfg).decode_maps(mtok, H=16, W=16), rgb)
    del sdm["decoder.conv_out.bias"]
    with pytest.raises(VQError, match="lacks 1 decoder"):
        Mapdecoder(sdm, cfg=mcfg)
