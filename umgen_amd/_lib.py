"""ctypes binding of libumgen_hip.so (C ABI: include/umgen.h).  No torch types cross this boundary.

The library is built in-tree by ``build_library()`` (hipcc --offload-arch=gfx950); loading fails loudly
when it is missing -- there is no CPU fallback for the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# UMGEN_LIB_PATH selects an alternative build of the SAME library (kernel experiments with extra -D flags); never a fallback
LIB_PATH = os.environ.get("UMGEN_LIB_PATH") or os.path.join(HERE, "libumgen_hip.so")
SOURCES = ["engine.hip", "gemm.hip", "gemm256.hip", "attn.hip", "gemv.hip", "oar_engine.hip", "oar_engine_wide.hip", "decode_batched.hip", "rowops.hip", "frame.hip", "tokenizers.hip", "vqdec.hip", "debug_api.hip"]
EXPORTS = ["umgen_create", "umgen_load_tensor", "umgen_finalize_weights", "umgen_rollout", "umgen_frame",
           "umgen_set_profiling", "umgen_get_timings", "umgen_last_error", "umgen_version", "umgen_destroy",
           "umgen_tokenize_ego", "umgen_detokenize_ego", "umgen_tokenize_boxes", "umgen_detokenize_boxes",
           "umgen_vq_create", "umgen_vq_load_tensor", "umgen_vq_finalize", "umgen_vq_decode", "umgen_vq_last_error", "umgen_vq_destroy",
           "umgen_dbg_linear", "umgen_dbg_attn_spatial", "umgen_dbg_attn_temporal", "umgen_dbg_attn_decode", "umgen_dbg_gemv", "umgen_dbg_gemm_bench", "umgen_dbg_oar_step", "umgen_dbg_sample_topk", "umgen_dbg_batched_layer_bench"]

HEADERS = ("common.h", "kernels.h", "frame.h", "oar_common.h", "bg_queue.h", "bg_worker.h", "gemm256_body.h", "attn_body.h", "rowops_body.h", "frame_body.h")

PREC_FP32, PREC_BF16, PREC_FP16 = 0, 1, 2
DT_F32, DT_BF16, DT_F16, DT_F64 = 0, 1, 2, 3


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "n_embd", "n_head", "n_ego_tar_layer", "n_ego_ca_layer", "n_map_tar_layer", "n_box_tar_layer",
        "n_tar_layer", "n_oar_layer", "pose_vocab", "map_vocab", "bbox3d_vocab", "img_vocab", "aux_vocab",
        "n_map_embd", "n_img_embd", "max_frame_len", "task_num", "task_id", "precision", "max_batch",
        "max_cond_frames", "device", "use_graphs")]


class VQConfig(C.Structure):
    _fields_ = [("n_embed", C.c_int32), ("embed_dim", C.c_int32), ("z_channels", C.c_int32), ("ch", C.c_int32), ("out_ch", C.c_int32),
                ("n_levels", C.c_int32), ("ch_mult", C.c_int32 * 8), ("num_res_blocks", C.c_int32), ("n_attn_res", C.c_int32),
                ("attn_resolutions", C.c_int32 * 4), ("resolution", C.c_int32), ("post_quant_ks", C.c_int32), ("post_quant_pad", C.c_int32),
                ("token_h", C.c_int32), ("token_w", C.c_int32), ("device", C.c_int32)]


class Sampling(C.Structure):
    _fields_ = [("method", C.c_int32), ("top_k", C.c_int32), ("top_k_map", C.c_int32), ("topk_image", C.c_int32),
                ("p", C.c_float), ("p_map", C.c_float), ("temperature", C.c_float),
                ("rule_constrain", C.c_int32), ("merge_ar_tar", C.c_int32), ("only_ar", C.c_int32),
                ("seeds", C.POINTER(C.c_uint64))]


class Trace(C.Structure):
    _fields_ = [("cond", C.POINTER(C.c_float)), ("ego_logits", C.POINTER(C.c_float)),
                ("logits_map", C.POINTER(C.c_float)), ("logits_bbox3d", C.POINTER(C.c_float)),
                ("logits_image", C.POINTER(C.c_float)),
                ("forced_pose", C.POINTER(C.c_int64)), ("forced_map", C.POINTER(C.c_int64)),
                ("forced_bbox3d", C.POINTER(C.c_int64)), ("forced_image", C.POINTER(C.c_int64)),
                ("counters", C.POINTER(C.c_int32)),
                ("given_map", C.POINTER(C.c_int64)), ("given_bbox3d", C.POINTER(C.c_int64))]


class Timings(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("ego_ms", C.c_double), ("tar_ms", C.c_double), ("oar_ms", C.c_double),
                ("frames", C.c_int64), ("oar_steps", C.c_int64), ("oar_kernels", C.c_int64),
                ("gemm_ms", C.c_double), ("gemm_launches", C.c_int64), ("gemm_flops", C.c_double), ("oar_bytes", C.c_double),
                ("attn_ms", C.c_double), ("attn_launches", C.c_int64), ("attn_flops", C.c_double),
                ("bg_ms", C.c_double), ("overlapped_frames", C.c_int64),
                ("layers_ms", C.c_double), ("layers_launches", C.c_int64), ("decode_engine", C.c_int32), ("engine_fallback", C.c_int32),
                ("decode_batched", C.c_int32), ("decode_lanes", C.c_int32), ("prefix_passes", C.c_int64)]


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: cannot build libumgen_hip.so")
    return p


def source_hash() -> str:
    """sha256 over every source the library is built from (and the compile flags): the staleness check of build_library."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, x) for x in HEADERS] + \
        [os.path.join(os.path.dirname(HERE), "include", "umgen.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has one unified file) instead of AGPRs -- the attention softmax
# reads every S tile and rescales O in place, which otherwise costs ~150 v_accvgpr moves per key tile (423 -> 572 TFLOP/s)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950: compiles every HIP source into umgen_amd/libumgen_hip.so (in-tree).  The library is rebuilt
    whenever the hash of its sources differs from the one recorded next to it (and compiled into umgen_version())."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    want = source_hash()
    stamp = LIB_PATH + ".srchash"
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return LIB_PATH
    # one object per source, cached by the hash of (source, headers, flags) under csrc/.obj/ and compiled in parallel: a one-file
    # change rebuilds one object + the link instead of the whole library
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, ".obj")
    os.makedirs(objdir, exist_ok=True)
    hdr = b"".join(open(os.path.join(CSRC, x), "rb").read() for x in HEADERS) + \
        open(os.path.join(os.path.dirname(HERE), "include", "umgen.h"), "rb").read()
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]

    def obj_for(src):
        extra = [f'-DUMGEN_SRC_HASH="{want}"'] if os.path.basename(src) == "engine.hip" else []     # umgen_version() lives there
        h = hashlib.sha256(open(src, "rb").read() + hdr + " ".join(cflags + extra).encode()).hexdigest()[:16]
        return os.path.join(objdir, f"{os.path.basename(src)}.{h}.o"), extra

    def compile_one(src):
        obj, extra = obj_for(src)
        if force or not os.path.exists(obj):
            t = f"{obj}.{os.getpid()}.tmp"
            cmd = [hipcc_path()] + cflags + extra + ["-c", src, "-o", t]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=CSRC)
            os.replace(t, obj)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, srcs))
    keep = set(objs)
    for f in os.listdir(objdir):                      # objects of older source states
        if os.path.join(objdir, f) not in keep and f.endswith(".o"):
            try:
                os.remove(os.path.join(objdir, f))
            except FileNotFoundError:         # another rank's first-use build removed it first
                pass
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, LIB_PATH)          # atomic: a concurrent rank never maps a half-written library
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return LIB_PATH


HOST_LIB_PATH = os.path.join(HERE, "libumgen_host.so")
HOST_EXPORTS = ["umgen_tokenize_ego", "umgen_detokenize_ego", "umgen_tokenize_boxes", "umgen_detokenize_boxes"]


def build_host_library(force: bool = False) -> str:
    """The scene-format entry points of the C ABI (csrc/tokenizers.hip: plain host C++, no device code) as a small library of their
    own, built with g++: dataset tokenisation / detokenisation (umgen_amd/scene_io.py, the reader either side of the rollout) then
    needs neither hipcc nor a ROCm runtime.  libumgen_hip.so exports the same four functions from the same source."""
    src = os.path.join(CSRC, "tokenizers.hip")
    if not force and os.path.exists(HOST_LIB_PATH) and os.path.getmtime(HOST_LIB_PATH) >= os.path.getmtime(src):
        return HOST_LIB_PATH
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no host C++ compiler (g++) found: cannot build libumgen_host.so")
    # several ranks may hit first use together (torchrun): compile to a private file and rename it into place, so that nobody can
    # CDLL a half-written library
    tmp = f"{HOST_LIB_PATH}.{os.getpid()}.tmp"
    try:
        subprocess.run([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-x", "c++", src,
                        "-o", tmp], check=True)
        os.replace(tmp, HOST_LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return HOST_LIB_PATH


_host = None


def load_host_library() -> C.CDLL:
    """Tokenizer / normaliser entry points (host only).  Built on first use when a compiler is there; fails loudly otherwise."""
    global _host
    if _host is not None:
        return _host
    lib = C.CDLL(build_host_library())
    f64p, f32p, i32p, i64p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    lib.umgen_tokenize_ego.argtypes = [f64p, C.c_int64, i64p]
    lib.umgen_detokenize_ego.argtypes = [i64p, C.c_int64, f32p]
    lib.umgen_tokenize_boxes.argtypes = [f32p, C.c_int64, C.c_int32, i32p, i64p]
    lib.umgen_detokenize_boxes.argtypes = [i64p, C.c_int64, f64p]
    for name in HOST_EXPORTS:
        getattr(lib, name).restype = C.c_int
    _host = lib
    return lib


_lib = None


def load_library() -> C.CDLL:
    """Loads the in-tree library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(umgen_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
    lib.umgen_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.umgen_load_tensor.argtypes = [vp, C.c_char_p, vp, i32, i64p, i32]
    lib.umgen_finalize_weights.argtypes = [vp]
    lib.umgen_rollout.argtypes = [vp, i32, i32, i32, i32, i64p, i64p, i64p, i64p, i32, i64p, i64p, i32, i64p, i64p,
                                  C.POINTER(Sampling), i64p, i64p, i64p, i64p]
    lib.umgen_frame.argtypes = [vp, i32, i64p, i64p, i64p, i64p, i64p, i64p, i32, C.POINTER(Sampling), i32,
                                C.POINTER(Trace), i64p, i64p, i64p, i64p]
    lib.umgen_set_profiling.argtypes = [vp, i32]
    lib.umgen_get_timings.argtypes = [vp, C.POINTER(Timings)]
    lib.umgen_last_error.argtypes = [vp]
    lib.umgen_last_error.restype = C.c_char_p
    lib.umgen_version.restype = C.c_char_p
    lib.umgen_destroy.argtypes = [vp]
    f64p, f32p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.umgen_tokenize_ego.argtypes = [f64p, C.c_int64, i64p]
    lib.umgen_detokenize_ego.argtypes = [i64p, C.c_int64, f32p]
    lib.umgen_tokenize_boxes.argtypes = [f32p, C.c_int64, i32, i32p, i64p]
    lib.umgen_detokenize_boxes.argtypes = [i64p, C.c_int64, f64p]
    fp = C.POINTER(C.c_float)
    lib.umgen_dbg_linear.argtypes = [i32, vp, vp, fp, i32, i32, i32, i32, i32, vp]
    lib.umgen_dbg_attn_spatial.argtypes = [i32, vp, vp, i32, i32, i32, vp]
    lib.umgen_dbg_attn_temporal.argtypes = [i32, vp, i32, i32, i32, i32, i32, vp]
    lib.umgen_dbg_attn_decode.argtypes = [i32, fp, vp, i32, i32, i32, fp]
    lib.umgen_dbg_gemv.argtypes = [i32, fp, fp, vp, fp, i32, i32, i32, i32, fp]
    lib.umgen_dbg_gemm_bench.argtypes = [i32, i32, i32, i32, i32, fp]
    lib.umgen_dbg_oar_step.argtypes = [vp, i32, i32, fp, fp, i32, i32]
    lib.umgen_dbg_sample_topk.argtypes = [fp, i32, i32, i32, C.c_float, fp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.umgen_dbg_batched_layer_bench.argtypes = [i32, i32, i32, i32, fp]
    lib.umgen_vq_create.argtypes = [C.POINTER(VQConfig), C.POINTER(vp)]
    lib.umgen_vq_load_tensor.argtypes = [vp, C.c_char_p, fp, i64p, i32]
    lib.umgen_vq_finalize.argtypes = [vp]
    lib.umgen_vq_decode.argtypes = [vp, i32, i64p, fp]
    lib.umgen_vq_last_error.argtypes = [vp]
    lib.umgen_vq_last_error.restype = C.c_char_p
    lib.umgen_vq_destroy.argtypes = [vp]
    for name in EXPORTS:
        if name not in ("umgen_last_error", "umgen_version", "umgen_vq_last_error"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib
