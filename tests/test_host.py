"""CPU tests of the host-side logic: config resolution, state-dict key set, constant tables, registry surface,
evaluate CLI value resolution, library ABI (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from umgen_amd import _lib
from umgen_amd.config import MOD_START, SEQ_LEN, TOKEN_LEN, RolloutConfig, large_config, tiny_config
from umgen_amd.weights import expected_keys, n_params, synth_tensor, synthetic_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scene_layout_matches_reference_constants():
    assert SEQ_LEN == 2207 and TOKEN_LEN == {"pose": 5, "map": 1026, "bbox3d": 662, "image": 514}
    assert MOD_START == {"pose": 0, "map": 5, "bbox3d": 1031, "image": 1693}


def test_large_config_parameter_count():
    # SURVEY.md section 0 / BASELINE.md: 2.447 B parameters for UMGen_Large (consumed subset excludes the unused heads)
    n = n_params(large_config())
    assert 2.40e9 < n < 2.46e9, n


def test_expected_keys_are_the_reference_state_dict_names():
    keys = expected_keys(tiny_config())
    for k in ("transformer.TAR.0.spatial_attn_1.c_attn.weight", "transformer.OAR.1.mlp.c_proj.weight",
              "transformer.ego_cross_attn.0.cross_attn.k_attn.bias", "transformer.head_tar_bbox3d.weight",
              "map_mlp_pre.c_fc.weight", "img_codebook.weight", "transformer.ln_box_tar.weight"):
        assert k in keys, k
    assert keys["transformer.TAR.0.spatial_attn_1.c_attn.weight"] == (288, 96)


def test_synthetic_weights_are_deterministic():
    a = synth_tensor("transformer.TAR.0.mlp1.c_fc.weight", (384, 96), seed=5)
    b = synth_tensor("transformer.TAR.0.mlp1.c_fc.weight", (384, 96), seed=5)
    c = synth_tensor("transformer.TAR.0.mlp1.c_fc.weight", (384, 96), seed=6)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(a).max() <= 1 / np.sqrt(96) + 1e-6


def test_config_from_reference_namespace_rejects_unsupported_switches():
    from argparse import Namespace
    base = dict(n_embd=768, n_head=16, n_ego_tar_layer=12, n_ego_ca_layer=12, n_map_tar_layer=24, n_box_tar_layer=24,
                n_tar_layer=36, n_oar_layer=36, pose_vocab_size=1024, map_vocab_size=8192, bbox3d_vocab_size=1028,
                img_vocab_size=8192, aux_vocab_size=8, n_map_embd=16, n_img_embd=16, max_frame_len=100, task_num=7,
                task_name_id={"pose_map_bbox3d_image": 6}, sample_method="topk", top_k=5, p=0.4, sfmx_temp=1.0,
                rule_constrain=True)
    c = RolloutConfig.from_namespace(Namespace(**base))
    assert c.head_dim == 48 and c.top_k_map == 5 and c.topk_image == 16
    with pytest.raises(NotImplementedError):
        RolloutConfig.from_namespace(Namespace(**base, box_transform=True))
    # switches that would silently change what the engine hard-codes are refused too (UMGen.py:99-172, infer_fun.py:84-139)
    for bad in (dict(split_image_ar=True), dict(add_posi_embedd=False), dict(add_spatial_pos_embedd_on_map=False),
                dict(seq_len=1693), dict(token_len={"pose": 5, "map": 1026, "bbox3d": 662}), dict(n_step=2), dict(no_born=True),
                dict(bos_eos={"pose": [0, 1], "map": [2, 3], "bbox3d": [4, 5]})):
        with pytest.raises(NotImplementedError):
            RolloutConfig.from_namespace(Namespace(**base, **bad))
    ok = RolloutConfig.from_namespace(Namespace(**base, seq_len=2207, token_len=dict(TOKEN_LEN), split_image_ar=False,
                                                bos_eos={"pose": [0, 1], "map": [2, 3], "bbox3d": [4, 5], "image": [6, 7]}))
    assert ok.n_oar_layer == 36


def test_model_class_tracks_the_device_it_is_moved_to(monkeypatch):
    """nn.Module surface used by the reference's harness (model_pl.py:366-368, 445-447): .to()/.cuda() select the engine's GPU
    (default LOCAL_RANK), .cpu() leaves it alone; no engine is created before weights arrive."""
    from umgen_amd.model import UMGen
    monkeypatch.setenv("LOCAL_RANK", "3")
    m = UMGen(tiny_config())
    assert m._engine_args["device"] == 3 and m._engine is None
    assert m.to("cuda:5") is m and m._engine_args["device"] == 5
    assert m.cuda() is m and m._engine_args["device"] == 3
    assert m.cuda(2)._engine_args["device"] == 2
    assert m.cpu()._engine_args["device"] == 2 and m.to(torch.float16)._engine_args["device"] == 2
    assert m.to(device=torch.device("cuda", 1))._engine_args["device"] == 1 and m._engine is None


def test_registry_build_from_cfg_with_class_object():
    from umgen_amd.registry import MODELS, Registry, build_from_cfg

    r = Registry("t")

    @r.register_module()
    class Foo:
        def __init__(self, config):
            self.config = config

    assert isinstance(build_from_cfg(dict(type=Foo, config=3), r), Foo)
    assert build_from_cfg(dict(type="Foo", config=4), r).config == 4
    import umgen_amd.model  # noqa: F401  (registers the drop-in class under the reference's name)
    assert MODELS.get("UMGen") is umgen_amd.model.UMGen


def test_evaluate_cli_resolution_follows_infer_fun():
    from umgen_amd.evaluate import build_parser, resolve
    cfg, nf, icf = resolve(build_parser().parse_args(["--infer_task", "video", "--set_num_new_frames", "30"]))
    assert (nf, icf, cfg.n_tar_layer, cfg.top_k, cfg.topk_image) == (30, 20, 36, 5, 16)
    cfg, nf, icf = resolve(build_parser().parse_args(["--infer_task", "control", "--set_num_new_frames", "7"]))
    assert (nf, icf) == (30, 13)     # infer_fun.py:68-71: control ignores --set_num_new_frames


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly what include/umgen.h declares (+ the kernel test hooks)."""
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_library()
    lib = _lib.load_library()
    header = open(os.path.join(ROOT, "include", "umgen.h")).read()
    declared = set(re.findall(r"\b(umgen_[a-z_0-9]+)\s*\(", header))
    declared -= {"umgen_step_logits"}
    assert {"umgen_create", "umgen_load_tensor", "umgen_finalize_weights", "umgen_rollout", "umgen_frame",
            "umgen_destroy", "umgen_last_error"} <= declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.umgen_version().startswith(b"umgen_hip")


def test_engine_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device the product path raises instead of computing elsewhere."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from umgen_amd.engine import Engine, UMGenError
    with pytest.raises(UMGenError):
        Engine(tiny_config(), max_cond_frames=4)


def test_struct_layouts_match_header_sizes():
    assert ctypes.sizeof(_lib.Config) == 24 * 4
    assert ctypes.sizeof(_lib.Sampling) == 48
    assert ctypes.sizeof(_lib.Trace) == 12 * 8


def test_vq_decoder_key_inventory_and_checkpoint_unwrapping(tmp_path):
    """umgen_amd/vq.py host logic (no GPU): the decoder's state-dict inventory of the two production configurations (vq_model.py:153-202)
    -- and, when the reference checkout is here, exactly the decode-path entries of its NormVQModel with the same shapes -- and
    the three checkpoint forms Mapdecoder / Imagedecoder accept (path like VQModel.init_from_ckpt, loaded dict, bare state dict)."""
    from umgen_amd.vq import IMAGE_VQ, MAP_VQ, _state_dict_of, decoder_keys
    ki, km = decoder_keys(IMAGE_VQ), decoder_keys(MAP_VQ)
    assert ki["quantize.embedding.weight"] == (8192, 16) and ki["post_quant_conv.weight"] == (256, 16, 3, 3)
    assert km["post_quant_conv.weight"] == (16, 16, 1, 1) and km["decoder.conv_out.weight"] == (5, 128, 3, 3)
    assert ki["decoder.conv_in.weight"] == (512, 256, 3, 3) and "decoder.up.4.attn.0.q.weight" in ki   # attention at resolution 32 = the deepest level
    assert sum(int(np.prod(s)) for s in ki.values()) > 40e6
    sd = {"decoder.conv_in.weight": torch.zeros(2), "encoder.x": torch.zeros(1)}
    assert _state_dict_of(None) is None and _state_dict_of(sd) is sd and _state_dict_of({"state_dict": sd, "epoch": 3}) is sd
    torch.save({"state_dict": sd}, tmp_path / "vq.pt")
    assert set(_state_dict_of(str(tmp_path / "vq.pt"))) == set(sd)
    ref = "/root/reference"
    if os.path.isdir(os.path.join(ref, "projects", "tokenizer")):
        sys.path.insert(0, ref)
        from projects.tokenizer.vq_model import NormVQModel
        for cfg in (MAP_VQ, IMAGE_VQ):
            dd = dict(double_z=False, z_channels=cfg["z_channels"], resolution=cfg["resolution"], in_channels=cfg["out_ch"], out_ch=cfg["out_ch"],
                      ch=cfg["ch"], ch_mult=list(cfg["ch_mult"]), num_res_blocks=cfg["num_res_blocks"],
                      attn_resolutions=list(cfg["attn_resolutions"]), dropout=0.0)
            m = NormVQModel(n_embed=cfg["n_embed"], embed_dim=cfg["embed_dim"], ddconfig=dd, stride=cfg["post_quant_ks"],
                            padding=cfg["post_quant_pad"], ckpt_path=None)
            want = {k: tuple(v.shape) for k, v in m.state_dict().items()
                    if k.startswith(("decoder.", "post_quant_conv.")) or k == "quantize.embedding.weight"}
            assert dict(decoder_keys(cfg)) == want
