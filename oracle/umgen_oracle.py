"""CPU ORACLE for the UMGen next-scene rollout -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

A plain PyTorch-CPU fp32 restatement of the reference's algorithm for the one hot path this repo
accelerates, ``UMGen.inference()`` (reference: projects/models/UMGen.py:1542-1671 and everything it
calls in projects/models/module.py).  Each function cites the reference lines it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product path (umgen_amd -> libumgen_hip.so)
never routes through it and fails loudly when the HIP library is missing.

How it is pinned
  * tests/golden/make_golden.py imports the real reference (under the stubs of tests/golden/refimport.py),
    loads the build's deterministic synthetic weights into it and records token sequences (full greedy)
    and stage activations; tests/test_oracle.py checks this oracle against those committed vectors, and
    tests/test_oracle_vs_reference.py re-runs the comparison live when /root/reference is present.
  * NOT pinned by anything in the reference: (i) the flash-attn==2.3.2 kernel
    numerics -- semantics fixed here as exact softmax attention, scale 1/sqrt(head_dim), bottom-right
    aligned causal mask; (ii) numba's lowering of ``ret[i, j] is False`` (misc.py:261-263) -- pure-Python
    semantics are used (the containment branch never runs); (iii) the ``torch.multinomial`` random stream --
    sampling with k>1 uses the build's counter-based RNG below, so only greedy decoding is comparable
    token-for-token with the reference; (iv) fp16-autocast CUDA numerics of the original deployment.

Modes
  * ``weight_dtype="fp32"`` : the reference's CPU fp32 semantics (what the import produces).
  * ``weight_dtype="bf16"`` : identical math with every >=2-D weight rounded to bf16 first.
  * ``weight_dtype="bf16_engine"`` : the rounding-aware restatement of the engine's production (bf16) mode: bf16
    weights AND a round-to-bf16 at every point where libumgen_hip stores or feeds a bf16 value --
      TAR / ego-TAR sub-blocks (gemm.hip / attn.hip): the LayerNorm output that becomes the GEMM A operand, the q | k | v rows,
      the softmax probabilities that feed the P.V MFMA of the SPATIAL attention (64-key tiles, rounded relative to the running
      maximum of the online softmax; the row sum keeps the unrounded fp32 values), the attention output, the GELU output; accumulation and the residual stream stay fp32;
      ego decoder (engine.hip run_ego): ln_3(p) and the cross-attention k | v rows;
      OAR decode (gemv.hip / oar_engine.hip): only the K/V cache rows (incl. the new token's); activations stay fp32.
    Everything else (embeddings, LayerNorm statistics, softmax, heads, samplers) is fp32 like the reference (module.py:34-37
    keeps LayerNorm in fp32 inside the autocast region of UMGen.py:1604-1605).  What this mode cannot reproduce is the ORDER
    of fp32 accumulation inside the MFMA tiles and the online-softmax rescaling, so the comparison is a tolerance, stated in
    the tests, not bit equality.  The fp32 mode is untouched by it (pinned by tests/test_oracle.py on the reference goldens).
  * ``weight_dtype="fp16"`` / ``"fp16_engine"`` : the same two restatements with IEEE half instead of bfloat16 (the engine's
    UMGEN_PREC_FP16 mode; the reference's own deployment arithmetic is torch.cuda.amp.autocast fp16, UMGen.py:1604-1605).

``perm_seed`` (accumulation-order ensemble): with a seed, every F.linear sums its K dimension in a seeded random order and every
attention sums its keys in a seeded random order -- mathematically the same function, another order of the fp32 additions.  The
spread of an ensemble of such runs is the noise floor any implementation with yet another summation order (the MFMA tiles, the
engine's split softmax) sits in: tests/golden/make_ensemble.py records it and the -m gpu tests bound the engine by it.
``mfma_noise``: the members additionally carry the fp32 accumulation noise of the matrix cores at every linear layer -- MEASURED on
MI355X (tools/dbg/mfma_error.py, profiles/r03_mfma_error.txt): against the exact product of the same 16-bit operands the MFMA GEMM is
unbiased with a relative rms error of 1.56e-7 at K = 768 and 3.39e-7 at K = 3072 (PyTorch's CPU sgemm: 1.0e-7 / 1.2e-7), i.e.
1.56e-7 * sqrt(K / 768).  Reordering a CPU sum alone under-states what the 16-bit rounding points downstream get to amplify.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from umgen_amd.weights import is_matrix_weight
from umgen_amd.config import (BBOX_PAD, BBOX_RANGE, BOS_EOS, CONTENT_LEN, EGO_BOX, EGO_STD, MOD_ORDER, MOD_START,
                              N_SLOTS, SEQ_LEN, SLOT_LEN, TOKEN_LEN, RolloutConfig)

MASK64 = (1 << 64) - 1

# --------------------------------------------------------------------------------------------------
# build-owned counter-based RNG (replaces torch.multinomial's stream; mirrored bit-for-bit in
# umgen_amd/csrc/common.h: rng_u24)
# --------------------------------------------------------------------------------------------------
DRAW_MAIN, DRAW_PAD_AVOID, DRAW_CONTROL = 0, 1, 2
EGO_POS_BASE = SEQ_LEN  # ego-net draws use positions SEQ_LEN + {0,1,2}


def rng_u24(seed: int, frame: int, pos: int, draw: int) -> int:
    x = (seed ^ ((frame + 1) * 0x9E3779B97F4A7C15) ^ ((pos + 1) * 0xBF58476D1CE4E5B9)
         ^ ((draw + 1) * 0x94D049BB133111EB)) & MASK64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & MASK64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & MASK64
    x ^= x >> 31
    return x >> 40


def rng_uniform(seed: int, frame: int, pos: int, draw: int) -> np.float32:
    return np.float32(rng_u24(seed, frame, pos, draw)) * np.float32(2.0 ** -24)


_EXP_C = [np.float32(v) for v in (1.0, 0.693147180559945, 0.240226506959101, 0.0555041086648216, 0.00961812910762848,
                                  0.00133335581464284, 1.54035303933816e-4, 1.52527338040598e-5, 1.32154867901443e-6,
                                  1.01780860092397e-7)]


def exp_det(x: np.ndarray) -> np.ndarray:
    """exp(x), x <= 0, bit-identical to the device's exp_det (umgen_amd/csrc/common.h): 2^n * P(f) with separately rounded
    fp32 multiplies and adds.  The samplers' softmax uses it on both sides so that the inverse-CDF walk sees the same bits."""
    x = np.asarray(x, dtype=np.float32)
    ok = x > np.float32(-87.0)
    xs = np.where(ok, x, np.float32(0.0)).astype(np.float32)
    t = xs * np.float32(1.44269504088896341)
    n = np.floor(t)
    f = (t - n).astype(np.float32)
    p = np.full_like(f, _EXP_C[9])
    for k in range(8, -1, -1):
        p = (p * f).astype(np.float32) + _EXP_C[k]
    return np.where(ok, np.ldexp(p, n.astype(np.int32)), np.float32(0.0)).astype(np.float32)


# --------------------------------------------------------------------------------------------------
# constant tables
# --------------------------------------------------------------------------------------------------
def position_encoding_init(n_position: int, emb_dim: int, start_index: int = 0) -> torch.Tensor:
    """module.py:746-768 -- sinusoid table, row 0 zeros, float64 -> bfloat16."""
    j = np.arange(emb_dim)
    denom = np.power(10000, 2 * (j // 2) / emb_dim)
    pos = (np.arange(n_position)[:, None] + start_index) / denom[None, :]
    pos[0, :] = 0.0
    pos[1:, 0::2] = np.sin(pos[1:, 0::2])
    pos[1:, 1::2] = np.cos(pos[1:, 1::2])
    return torch.from_numpy(pos).type(torch.bfloat16)


def grid_center_tokens() -> np.ndarray:
    """UMGen.py:140-150,357-383 -- (x,y) bin token of each of the 32x32 map-grid centres, [1024,2]."""
    g = torch.arange(0, 32)
    gx, gy = torch.meshgrid(g, g, indexing="ij")
    cx = -((gx + 0.5) * 4.0 - 64.0)
    cy = -((gy + 0.5) * 4.0 - 64.0)
    centers = torch.stack([cx, cy], dim=-1)
    norm = ((centers + 64) / 128).numpy()
    tok = np.digitize(norm, np.linspace(0.0, 1.0, 1024))
    return tok.reshape(1024, 2)


# --------------------------------------------------------------------------------------------------
# CPU-side helpers the reference calls inside the decode loop
# --------------------------------------------------------------------------------------------------
_EGO_BINS = np.linspace(-1.0, 1.0, 1024)
_BOX_BINS = np.linspace(0.0, 1.0, 1024)
_EGO_INV_STD = 1.0 / np.array(EGO_STD, dtype=np.float32)  # normalize.py:26


def decode_pose_values(pose_tokens: np.ndarray) -> np.ndarray:
    """UMGen.decode_pose (UMGen.py:1008-1024) -> DigitalBinsTokenizer.decode (tokenizer.py:332-354)
    -> Normalize_Standard.unnormalize_ego (normalize.py:65-76).  int [...,3] -> float32 (dx, dy, dtheta)."""
    t = np.asarray(pose_tokens, dtype=np.int64)
    right = np.clip(t, 0, 1023)
    left = np.clip(t - 1, 0, 1023)
    v = (_EGO_BINS[left] + _EGO_BINS[right]) / 2
    v = v / _EGO_INV_STD + np.zeros(3, dtype=np.float32)
    return v.astype(np.float32)


def decode_box_values(slot_tokens: np.ndarray) -> np.ndarray:
    """BBox3DTokenizer.decode_single_objects (tokenizer.py:679-687) + Normalize.unnormalize_bbox3d
    (normalize.py:136-149,189-229).  11 ints -> 10 float64 attributes."""
    t = np.asarray(slot_tokens[:10], dtype=np.int64)
    right = np.clip(t, 0, 1023)
    left = np.clip(t - 1, 0, 1023)
    v = (_BOX_BINS[left] + _BOX_BINS[right]) / 2
    out = np.empty(10, dtype=np.float64)
    for a, (lo, hi) in enumerate(BBOX_RANGE):
        out[a] = v[a] * (hi - lo) + lo
    return out


def bev_corners(boxes: np.ndarray) -> np.ndarray:
    """misc.py:143-177 bbox3d2bevcorners on (x, y, z, l, w, h, yaw): float64 math, float32 result."""
    centers, dims, angles = boxes[:, :2], boxes[:, 3:5], boxes[:, 6]
    tmpl = np.array([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]], dtype=np.float32)
    c = tmpl[None, ...] * dims[:, None, :]
    s_, c_ = np.sin(angles), np.cos(angles)
    rot = np.transpose(np.array([[c_, -s_], [s_, c_]]), (2, 1, 0))
    c = c @ rot
    c += centers[:, None, :]
    return c.astype(np.float32)


def box_collides_with_last(corners: np.ndarray) -> bool:
    """misc.py:203-311 box_collision_test(boxes=all, qboxes=last) reduced to np.any(ret[:, 0])
    (misc.py:620-626).  float32 arithmetic, pure-Python semantics for ``ret[i, j] is False``
    (never true for a numpy bool => the containment branch is skipped; parity unpinned vs numba)."""
    n = corners.shape[0]
    q = corners[-1]
    f = np.float32
    qx0, qx1 = q[:, 0].min(), q[:, 0].max()
    qy0, qy1 = q[:, 1].min(), q[:, 1].max()
    for i in range(n):
        b = corners[i]
        iw = min(b[:, 0].max(), qx1) - max(b[:, 0].min(), qx0)
        if not iw > 0:
            continue
        ih = min(b[:, 1].max(), qy1) - max(b[:, 1].min(), qy0)
        if not ih > 0:
            continue
        for k in range(4):
            A, B = b[k], b[(k + 1) % 4]
            for l in range(4):
                C, D = q[l], q[(l + 1) % 4]
                acd = f(D[1] - A[1]) * f(C[0] - A[0]) > f(C[1] - A[1]) * f(D[0] - A[0])
                bcd = f(D[1] - B[1]) * f(C[0] - B[0]) > f(C[1] - B[1]) * f(D[0] - B[0])
                if acd != bcd:
                    abc = f(C[1] - A[1]) * f(B[0] - A[0]) > f(B[1] - A[1]) * f(C[0] - A[0])
                    abd = f(D[1] - A[1]) * f(B[0] - A[0]) > f(B[1] - A[1]) * f(D[0] - A[0])
                    if abc != abd:
                        return True
    return False


def check_collision(decoded: List[np.ndarray]) -> bool:
    """BoxOverlap.check_collision(box, fliter=True) (misc.py:591-630) incl. fliter_and_map_object
    (misc.py:475-481: drop boxes with x >= 63).  NB the query box is the LAST box that survives the
    filter, and it is also tested against itself (boxes=bbox2d includes it)."""
    if len(decoded) == 1:
        return False
    b = np.array(decoded)
    b = b[[i for i in range(len(b)) if not b[i][0] >= 63]]
    if b.shape[0] <= 1:
        return False
    seven = np.concatenate([b[:, 0:6], (-b[:, 6]).reshape(-1, 1)], axis=1)
    return box_collides_with_last(bev_corners(seven))


# --------------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------------
def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.bfloat16().float()


def _fp16(x: torch.Tensor) -> torch.Tensor:
    return x.half().float()


def _attention(q, k, v, n_head: int, causal: bool, round_p=None, perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """flash_attn_func as called at module.py:218-225 / 497-504 (third-party; semantics fixed in the header).
    round_p: the engine's spatial attention feeds 16-bit probabilities (rounded by this function) to the P.V MFMA and divides by
    the fp32 row sum.  perm: order in which the keys are summed (accumulation-order ensemble)."""
    B, Tq, C = q.shape
    Tk = k.shape[1]
    D = C // n_head
    qh = q.view(B, Tq, n_head, D).permute(0, 2, 1, 3)
    kh = k.view(B, Tk, n_head, D).permute(0, 2, 1, 3)
    vh = v.view(B, Tk, n_head, D).permute(0, 2, 1, 3)
    scale = float(torch.tensor(1.0 / math.sqrt(C / n_head)))  # module.py:196-198 (fp32 buffer)
    att = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        i = torch.arange(Tq).view(-1, 1)
        j = torch.arange(Tk).view(1, -1)
        att = att.masked_fill(j > i + (Tk - Tq), float("-inf"))
    if perm is not None:
        att, vh = att[..., perm], vh[:, :, perm]
    if round_p is not None:
        # the engine's spatial attention (attn.hip): 64-key tiles, online softmax -- the probabilities are rounded to 16 bits RELATIVE
        # TO THE RUNNING MAXIMUM of their tile (that is what feeds the P.V MFMA), the row sum keeps the unrounded fp32 values and the
        # fp32 accumulator is rescaled when the maximum moves.  Which value gets which rounding error depends on the key order, so
        # this (not a one-shot softmax) is the form whose accumulation-order ensemble shows the engine's real noise floor.
        m = torch.full(att.shape[:-1] + (1,), float("-inf"))
        l = torch.zeros_like(m)
        o = torch.zeros(att.shape[:-1] + (D,))
        for t0 in range(0, Tk, 64):
            st = att[..., t0:t0 + 64]
            m_new = torch.maximum(m, st.amax(dim=-1, keepdim=True))
            alpha = torch.exp(m - m_new)
            pt = torch.exp(st - m_new)
            l = l * alpha + pt.sum(dim=-1, keepdim=True)
            o = o * alpha + round_p(pt) @ vh[:, :, t0:t0 + 64]
            m = m_new
        return (o / l).permute(0, 2, 1, 3).reshape(B, Tq, C)
    att = torch.softmax(att, dim=-1)
    return (att @ vh).permute(0, 2, 1, 3).reshape(B, Tq, C)


class OracleUMGen:
    def __init__(self, cfg: RolloutConfig, state_dict: Dict[str, np.ndarray], weight_dtype: str = "fp32", perm_seed: Optional[int] = None,
                 mfma_noise: bool = False, prefix_contract: str = "decode"):
        self.cfg = cfg
        # How the GIVEN-token prefix of a frame (UMGen.py:1184-1201: the first iteration of infer_oar_net pushes the whole prefix through the BlockOAR layers at once) is
        # rounded in the *_engine modes.  "decode": like every decode step (fp32 activations, only the K/V rows in 16 bits) -- the engine's step-by-step replay.  "stack": the
        # engine's ONE-PASS form (engine.hip run_prefix_prefill): positions 0 .. P - 2 go through the TAR stacks' kernels -- 16-bit LayerNorm outputs, q | k | v, attention
        # probabilities and outputs, MLP hidden values; fp32 residual stream -- and position P - 1 is a decode step on their K/V rows.
        assert prefix_contract in ("decode", "stack"), prefix_contract
        self.prefix_contract = prefix_contract
        self._mfma_noise = mfma_noise and perm_seed is not None
        assert weight_dtype in ("fp32", "bf16", "bf16_engine", "fp16", "fp16_engine"), weight_dtype
        self._round = _fp16 if weight_dtype.startswith("fp16") else _bf16
        self._perm_gen = torch.Generator().manual_seed(perm_seed) if perm_seed is not None else None
        self._perms: Dict[int, torch.Tensor] = {}
        self.w: Dict[str, torch.Tensor] = {}
        for k, v in state_dict.items():
            t = torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v
            if t.dtype != torch.bfloat16:
                t = t.float()
                # 16-bit modes: exactly the tensors the engine stores in 16 bits -- the nn.Linear weights (umgen_amd.weights.is_matrix_weight:
                # attention / MLP / GMLP / head matrices).  Embedding tables, codebooks, LayerNorm weights and biases stay fp32 in the engine
                # (and under the reference's autocast, which only casts the operands of matmuls): rounding them here as well -- rounds 1-2
                # did -- put the oracle FURTHER from the fp32 truth than the engine is (tiny config, bf16: rms 1.8e-3 vs 1.0e-3).
                if weight_dtype != "fp32" and is_matrix_weight(k):
                    t = self._round(t)
            self.w[k] = t
        self.engine_rounding = weight_dtype.endswith("_engine")
        E = cfg.n_embd
        # UMGen.py:137-153 (tables may be overridden by checkpoint entries, UMGen.py:257-261)
        self.fouier_pe = self.w.get("fouier_pe", position_encoding_init(1024, E)).bfloat16()
        self.posi = self.w.get("bbox3d_spatial_posi", position_encoding_init(1030, E, start_index=1024)).bfloat16()
        if "grid_center_posi_embedding" in self.w:
            self.grid_posi = self.w["grid_center_posi_embedding"].bfloat16()
        else:
            gt = torch.from_numpy(grid_center_tokens())
            self.grid_posi = self.posi[gt[:, 0]] + self.posi[gt[:, 1]]  # bf16 + bf16 -> bf16
        self.trace: Optional[dict] = None
        self.counters: Dict[str, int] = {}

    def _count(self, what: str):
        self.counters[what] = self.counters.get(what, 0) + 1

    # ---- primitives (module.py) -------------------------------------------------------------
    def _r(self, x):
        """16-bit round trip at the engine's 16-bit storage points (bf16_engine / fp16_engine modes only)."""
        return self._round(x) if self.engine_rounding else x

    def _perm(self, n: int) -> Optional[torch.Tensor]:
        """Summation order of an n-term reduction (None: natural order).  One fixed permutation per length and oracle instance."""
        if self._perm_gen is None or n < 2:
            return None
        if n not in self._perms:
            self._perms[n] = torch.randperm(n, generator=self._perm_gen)
        return self._perms[n]

    def _ln(self, x, key):  # module.py:26-37: weight only, eps 1e-5
        w = self.w[key + ".weight"]
        return F.layer_norm(x, w.shape, w, None, 1e-5)

    def _lin(self, x, key, bias=True):
        w = self.w[key + ".weight"]
        pm = self._perm(w.shape[1])
        if pm is not None:
            x, w = x[..., pm], w[:, pm]
        y = F.linear(x, w, None)
        if self._mfma_noise:   # measured fp32 accumulation noise of the matrix cores (see the header), relative to the products' rms
            eta = 1.56e-7 * math.sqrt(w.shape[1] / 768.0)
            y = y + (eta * float(y.pow(2).mean().sqrt())) * torch.randn(y.shape, generator=self._perm_gen)
        b = self.w.get(key + ".bias") if bias else None
        return y if b is None else y + b

    def _mlp(self, x, key, tar=False):  # module.py:233-250 (exact erf GELU, no bias)
        h = F.gelu(self._lin(x, key + ".c_fc", bias=False))
        return self._lin(self._r(h) if tar else h, key + ".c_proj", bias=False)

    def _self_attn(self, x, key, causal, kv=None, site="plain"):
        """CausalFlashAttention.forward (module.py:201-230).  site: where the engine rounds to bf16 (bf16_engine mode) --
        "tar_spatial" / "tar_temporal": q | k | v and the output (+ the probabilities of the spatial form); "oar": the K/V rows."""
        E = self.cfg.n_embd
        q, k, v = self._lin(x, key + ".c_attn").split(E, dim=2)
        if site in ("tar_spatial", "tar_temporal"):
            q, k, v = self._r(q), self._r(k), self._r(v)
        elif site == "oar":
            k, v = self._r(k), self._r(v)
        if kv is not None and kv[0] is not None:
            k = torch.cat([kv[0], k], dim=1)
            v = torch.cat([kv[1], v], dim=1)
        y = _attention(q, k, v, self.cfg.n_head, causal, round_p=self._round if (self.engine_rounding and site == "tar_spatial") else None,
                       perm=self._perm(k.shape[1]))
        if site in ("tar_spatial", "tar_temporal"):
            y = self._r(y)
        return self._lin(y, key + ".c_proj"), (k, v)

    def _block_tar(self, x, key):
        """BlockTAR.forward_func (module.py:332-359); kvcache is always None at inference."""
        B, T, S, C = x.shape
        x = x.reshape(B * T, S, C)
        r = self._r
        x = x + self._self_attn(r(self._ln(x, key + ".ln_1")), key + ".spatial_attn_1", False, site="tar_spatial")[0]
        x = x + self._mlp(r(self._ln(x, key + ".ln_2")), key + ".mlp1", tar=True)
        x = x.view(B, T, S, C).permute(0, 2, 1, 3).reshape(B * S, T, C)
        x = x + self._self_attn(r(self._ln(x, key + ".ln_3")), key + ".temporal_attn", True, site="tar_temporal")[0]
        x = x + self._mlp(r(self._ln(x, key + ".ln_4")), key + ".mlp2", tar=True)
        x = x.view(B, S, T, C).permute(0, 2, 1, 3).reshape(B * T, S, C)
        x = x + self._self_attn(r(self._ln(x, key + ".ln_5")), key + ".spatial_attn_2", False, site="tar_spatial")[0]
        x = x + self._mlp(r(self._ln(x, key + ".ln_6")), key + ".mlp3", tar=True)
        return x.view(B, T, S, C)

    def _block_oar(self, x, key, kv):
        """BlockOAR.forward_func (module.py:402-416) on x [B, s, C] with the cat-grown KV cache."""
        a, kv = self._self_attn(self._ln(x, key + ".ln_1"), key + ".temporal_attn", True, kv, site="oar")
        x = x + a
        x = x + self._mlp(self._ln(x, key + ".ln_2"), key + ".mlp")
        return x, kv

    def _prefix_pass(self, x, kv):
        """The engine's one-pass form of the given-token prefix (engine.hip run_prefix_prefill; prefix_contract == "stack"): rows 0 .. P - 2 with the TAR stacks'
        rounding points and the causal S x S attention, row P - 1 as a decode step on their K/V rows.  Fills kv, returns x of all P rows."""
        r = self._r
        xa, xb = x[:, :-1], x[:, -1:]
        for i in range(self.cfg.n_oar_layer):
            key = f"transformer.OAR.{i}"
            a, (ka, va) = self._self_attn(r(self._ln(xa, key + ".ln_1")), key + ".temporal_attn", True, site="tar_spatial")
            xa = xa + a
            xa = xa + self._mlp(r(self._ln(xa, key + ".ln_2")), key + ".mlp", tar=True)
            xb, kv[i] = self._block_oar(xb, key, (ka, va))
        return torch.cat([xa, xb], dim=1)

    def _decoder(self, x, p, key):
        """Decoder.forward_func (module.py:662-683) + FlashCrossAttention.forward (module.py:482-509)."""
        x = x + self._self_attn(self._ln(x, key + ".ln_1"), key + ".self_attn", False)[0]
        qn, pn = self._ln(x, key + ".ln_2"), self._r(self._ln(p, key + ".ln_3"))
        q = self._lin(qn, key + ".cross_attn.q_attn")
        k = self._r(self._lin(pn, key + ".cross_attn.k_attn"))
        v = self._r(self._lin(pn, key + ".cross_attn.v_attn"))
        y = _attention(q, k, v, self.cfg.n_head, False, perm=self._perm(k.shape[1]))
        x = x + self._lin(y, key + ".cross_attn.c_proj")
        return x + self._mlp(self._ln(x, key + ".ln_4"), key + ".mlp1")

    # ---- embeddings (UMGen.py:411-515) ---------------------------------------------------------
    def _gmlp(self, tok, which):  # module.py:710-743 on codebook rows
        cb = self.w[f"{which}_codebook.weight"][tok]
        return self._mlp(cb, f"{which}_mlp_pre")

    def _emb_mod(self, tokens: Dict[str, torch.Tensor], mod: str, map_posi=False) -> torch.Tensor:
        """get_mod_emb_pre (UMGen.py:438-468); the bf16 tables are added in their own dtype first."""
        t = tokens[mod]
        if mod == "bbox3d":
            f = self.w["transformer.be.weight"][t]
            B, T, S = t.shape
            xy = t.reshape(B, T, N_SLOTS, -1)
            pe = self.posi[xy[..., 0]] + self.posi[xy[..., 1]]  # bf16 add (UMGen.py:418-423)
            pe = pe.unsqueeze(-2).expand(-1, -1, -1, S // N_SLOTS, -1).reshape(B, T, S, -1)
            return f + pe
        if mod == "map":
            f = self._gmlp(t, "map")
            return f + self.grid_posi if map_posi else f
        if mod == "pose":
            return self.fouier_pe[t]
        if mod == "image":
            return self._gmlp(t, "img")
        raise ValueError(mod)

    def _bos_eos(self, f: torch.Tensor, mod: str) -> torch.Tensor:  # UMGen.py:470-481
        B, T = f.shape[:2]
        axe = self.w["transformer.axe.weight"]
        b = axe[BOS_EOS[mod][0]].expand(B, T, 1, -1)
        e = axe[BOS_EOS[mod][1]].expand(B, T, 1, -1)
        return torch.cat([b, f, e], dim=2)

    def _pos_emb(self, x):  # UMGen.py:483-515
        B, T, S, C = x.shape
        return x + self.w["transformer.spe.weight"][:S][None, None] + self.w["transformer.tpe.weight"][:T][None, :, None]

    def _affine(self, x: torch.Tensor, pose_diff: torch.Tensor) -> torch.Tensor:
        """affine_transform (UMGen.py:310-354): rigid warp of the 32x32 map-feature grid."""
        B, T, S, C = x.shape
        H = W = int(np.sqrt(S))
        xi = x.reshape(B * T, H, W, C).permute(0, 3, 1, 2)
        pd = pose_diff.reshape(B * T, 3)
        theta = pd[:, 2]
        dx = 2 * (pd[:, 0] / 4.0) / W
        dy = 2 * (pd[:, 1] / 4.0) / H
        m = theta.new_zeros((B * T, 2, 3))
        m[:, 0, 0] = torch.cos(-theta)
        m[:, 0, 1] = -torch.sin(-theta)
        m[:, 0, 2] = -dy
        m[:, 1, 0] = torch.sin(-theta)
        m[:, 1, 1] = torch.cos(-theta)
        m[:, 1, 2] = -dx
        grid = F.affine_grid(m, (B * T, C, H, W), align_corners=False)
        o = F.grid_sample(xi, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        return o.permute(0, 2, 3, 1).reshape(B, T, S, C).to(x.dtype)

    # ---- stacks (UMGen.py:634-872) ---------------------------------------------------------------
    def _run_stack(self, x, name, n, ln):
        for i in range(n):
            x = self._block_tar(x, f"transformer.{name}.{i}")
        return self._ln(x, "transformer." + ln)

    def forward_ego_net(self, tokens) -> torch.Tensor:
        """forward_ego_net (UMGen.py:634-687).  Only frame t=-1 of the decoder output is consumed
        (UMGen.py:1002) and the Decoder is frame-local, so only that frame is evaluated."""
        seq = [self._bos_eos(self._emb_mod(tokens, m), m) for m in MOD_ORDER]
        x = self._pos_emb(torch.cat(seq, dim=2))
        x = self._run_stack(x, "ego_tar", self.cfg.n_ego_tar_layer, "ln_ego_tar")
        B, T = x.shape[:2]
        e = self.w["transformer.egoe.weight"][None, None].expand(B, T, -1, -1)
        e = self._pos_emb(e)[:, -1]            # [B,3,C] (last frame only)
        p = x[:, -1]                           # [B,2207,C]
        for i in range(self.cfg.n_ego_ca_layer):
            e = self._decoder(e, p, f"transformer.ego_cross_attn.{i}")
        return self._ln(e, "transformer.ln_ego")  # [B,3,C]

    def forward_tar(self, tokens, stack: str):
        """forward_tar_for_map / _for_box / forward_tar_net (UMGen.py:781-817, 819-872, 691-778)."""
        mods = {"map_tar": MOD_ORDER[:2], "box_tar": MOD_ORDER[:3], "TAR": MOD_ORDER}[stack]
        pose_diff = torch.from_numpy(decode_pose_values(tokens["pose"].numpy()))
        seq, warped = [], None
        for m in mods:
            f = self._emb_mod(tokens, m, map_posi=(stack == "TAR"))
            if m == "map":
                warped = self._affine(f, pose_diff)
                f = warped + f
            seq.append(self._bos_eos(f, m))
        x = self._pos_emb(torch.cat(seq, dim=2))
        n = {"map_tar": self.cfg.n_map_tar_layer, "box_tar": self.cfg.n_box_tar_layer, "TAR": self.cfg.n_tar_layer}[stack]
        x = self._run_stack(x, stack, n, {"map_tar": "ln_map_tar", "box_tar": "ln_box_tar", "TAR": "ln_tar"}[stack])
        return x, warped

    # ---- samplers (UMGen.py:899-974) -------------------------------------------------------------
    def sample(self, logits: torch.Tensor, k: int, p: float, u: np.float32) -> int:
        """topk (UMGen.py:899-913) + sfmx_temp_sampling (967-974), or sample_top_p (915-965), with the
        multinomial draw replaced by inverse-CDF on the build's uniform ``u`` (kept set in index order
        for top-k; descending-probability order for top-p, as the reference sorts)."""
        l = logits.detach().float().numpy().astype(np.float32)
        temp = np.float32(self.cfg.sfmx_temp)
        if self.cfg.sample_method == "topk":
            kk = min(k, l.shape[0])
            kth = np.partition(l, -kk)[-kk]
            idx = np.nonzero(l >= kth)[0]
            z = l[idx] / temp
            e = exp_det(z - z.max())
        else:
            z = l / temp
            pr = exp_det(z - z.max())
            pr = pr / np.cumsum(pr, dtype=np.float32)[-1]      # sequential fp32 sum in index order (mirrored on the device)
            order = np.argsort(-pr, kind="stable")              # ties: ascending index
            ps = pr[order]
            cum = np.cumsum(ps, dtype=np.float32)                # sequential fp32 prefix sums
            n = 1 + int(np.sum(cum[:-1] <= np.float32(p)))       # entry j is masked when (cumsum - p_j) > p  (UMGen.py:950)
            idx = order[:n]
            e = ps[:n]
        c = np.cumsum(e, dtype=np.float32)                        # == the sequential loop total += x
        target = np.float32(u * c[-1])
        hit = np.nonzero(c > target)[0]
        return int(idx[hit[0]]) if hit.size else int(idx[-1])

    # ---- one frame (UMGen._inference, UMGen.py:1406-1540) ---------------------------------------
    def _frame(self, window: Dict[str, torch.Tensor], init: Optional[Dict[str, torch.Tensor]], control_test: bool,
               seed: int, frame_idx: int, forced: Optional[Dict[str, np.ndarray]] = None):
        cfg = self.cfg
        tr = self.trace
        inputs = dict(window)
        if init is not None and init.get("pose") is not None:
            ego = init["pose"]
        else:
            e = self.forward_ego_net(inputs)                              # [1,3,C]
            lg = self._lin(e, "transformer.head_ego", bias=False)[0]      # [3,1024]
            if tr is not None:
                tr.setdefault("ego_logits", []).append(lg.numpy().copy())
            toks = []
            for j in range(3):
                u = rng_uniform(seed, frame_idx, EGO_POS_BASE + j, DRAW_MAIN)
                toks.append(self.sample(lg[j], cfg.top_k, cfg.p, u))
            if forced is not None:
                toks = [int(x) for x in forced["pose"]]
            ego = torch.tensor(toks, dtype=torch.long).view(1, 1, 3)
        inputs["pose"] = torch.cat([inputs["pose"], ego], dim=1)[:, 1:]     # UMGen.py:1445-1452

        control_slots = None
        if init is not None and init.get("bbox3d") is not None and control_test:   # UMGen.py:1458-1473
            valid = init["bbox3d"][0, -1, :] != -1
            inputs["bbox3d"][0, -1, valid] = init["bbox3d"][0, -1, valid]           # in place on the window
            control_slots = np.where(valid.reshape(N_SLOTS, -1).any(dim=1).numpy())[0]

        x_map, warped_nopos = self.forward_tar(inputs, "map_tar")
        x_box, _ = self.forward_tar(inputs, "box_tar")
        x_tar, _ = self.forward_tar(inputs, "TAR")
        # UMGen.py:1496-1511: per-modality conditioning rows, last history frame only is consumed (1228-1230)
        cond = x_tar[:, -1].clone()                                        # [1,2207,C]
        ms, bs = MOD_START["map"], MOD_START["bbox3d"]
        cond[:, ms:ms + TOKEN_LEN["map"]] = x_map[:, -1, ms:ms + TOKEN_LEN["map"]]
        cond[:, ms + 1:ms + 1 + CONTENT_LEN["map"]] += warped_nopos[:, -1]
        cond[:, bs:bs + TOKEN_LEN["bbox3d"]] = x_box[:, -1, bs:bs + TOKEN_LEN["bbox3d"]]
        if tr is not None:
            tr.setdefault("cond", []).append(cond[0].numpy().copy())
        # modalities GIVEN for this frame besides the pose (infer_oar_net, UMGen.py:1184-1201: "use the predefined tokens and don't infer
        # these tokens any more"): they must continue the pose prefix in scene order -- map, or map + bbox3d; image tokens are dropped by
        # _inference (UMGen.py:1512-1520), bbox3d under control_test was consumed above (UMGen.py:1473)
        given = {}
        if init is not None:
            if init.get("map") is not None:
                given["map"] = init["map"][0, -1]
            if init.get("bbox3d") is not None and not control_test:
                given["bbox3d"] = init["bbox3d"][0, -1]
        res = self._oar(cond, ego, inputs, control_slots, seed, frame_idx, forced, given)
        return res

    # ---- the OAR decode loop (infer_oar_net + sample_next_token, UMGen.py:1029-1273) --------------
    def _oar(self, cond, ego, prev_tokens, control_slots, seed, frame_idx, forced, given=None):
        cfg = self.cfg
        w = self.w
        tr = self.trace
        axe = w["transformer.axe.weight"]
        d_pos = {}
        for m in MOD_ORDER:                                   # d_token_pos (UMGen.py:976-984), 1-based
            d_pos[MOD_START[m] + 1] = BOS_EOS[m][0]
            d_pos[MOD_START[m] + TOKEN_LEN[m]] = BOS_EOS[m][1]
        task = w["transformer.tske.weight"][cfg.task_id][None, None]                    # [1,1,C]
        pose_emb = self.fouier_pe[ego[:, 0]]                                              # [1,3,C] bf16
        prefix = torch.cat([axe[0][None, None], pose_emb.float(), axe[1][None, None]], dim=1)  # [1,5,C]
        res = {"pose": [int(t) for t in ego.view(-1)], "map": [], "bbox3d": [], "image": []}
        given = given or {}
        if "bbox3d" in given and "map" not in given:
            raise ValueError("init_tokens must continue the pose prefix in scene order (map, or map + bbox3d): the reference concatenates "
                             "the given modalities back to back (UMGen.py:1190-1201), so bbox3d without map lands on the map positions")
        if "map" in given:       # get_mod_emb_pre + add_bos_eos of the given tokens (UMGen.py:1194-1196): GMLP(codebook) rows, no position table
            tm = given["map"].view(-1)
            prefix = torch.cat([prefix, axe[BOS_EOS["map"][0]][None, None], self._gmlp(tm, "map")[None], axe[BOS_EOS["map"][1]][None, None]], dim=1)
            res["map"] = [int(t) for t in tm]
        if "bbox3d" in given:
            tb = given["bbox3d"].view(-1)
            prefix = torch.cat([prefix, axe[BOS_EOS["bbox3d"][0]][None, None], w["transformer.be.weight"][tb][None], axe[BOS_EOS["bbox3d"][1]][None, None]], dim=1)
            res["bbox3d"] = [int(t) for t in tb]
        exist = prefix.shape[1]                                                             # exist_seq_len (UMGen.py:1199)
        decoded_boxes: List[np.ndarray] = []
        kv = [None] * cfg.n_oar_layer
        x_in = torch.cat([task, prefix], dim=1) + cond[:, :exist + 1]                       # first call: task + the whole given prefix
        head = {"map": "head_ar_map", "bbox3d": "head_ar_bbox3d", "image": "head_ar_img"}
        logit_trace = {"map": [], "bbox3d": [], "image": []} if tr is not None else None
        prev_box = prev_tokens["bbox3d"][0, -1].numpy()
        for pos in range(exist + 1, SEQ_LEN + 1):             # pos == curr_seq_len (1-based)
            if pos == SEQ_LEN:
                break  # img-eos: the reference still runs a forward whose output is unused (UMGen.py:1209)
            x = x_in
            if self.engine_rounding and self.prefix_contract == "stack" and x.shape[1] > 1 and kv[0] is None:
                x = self._prefix_pass(x, kv)
            else:
                for i in range(cfg.n_oar_layer):
                    x, kv_i = self._block_oar(x, f"transformer.OAR.{i}", (kv[i] if kv[i] is not None else (None, None)))
                    kv[i] = kv_i
            h = self._ln(x[:, -1:], "transformer.ln_oar")                                  # [1,1,C]
            if pos in d_pos:
                nxt = axe[d_pos[pos]][None, None]
            else:
                mod = next(m for m in MOD_ORDER if MOD_START[m] + 1 <= pos <= MOD_START[m] + TOKEN_LEN[m])
                lg = self._lin(h[0, 0], f"transformer.{head[mod]}", bias=False)
                if logit_trace is not None:
                    logit_trace[mod].append(lg.numpy().copy())
                u = rng_uniform(seed, frame_idx, pos, DRAW_MAIN)
                if mod == "map":
                    tok = self.sample(lg, cfg.top_k_map, cfg.p_map, u)
                elif mod == "image":
                    # UMGen.py:1133 passes topk_image as the sampler parameter: in top-p mode that is p = 16.0 (keep everything)
                    tok = self.sample(lg, cfg.topk_image, float(cfg.topk_image), u)
                else:
                    tok = self._sample_bbox(lg, cond[0, pos - 1], pos, prev_box, control_slots, seed, frame_idx, u)
                if forced is None and mod == "bbox3d" and cfg.rule_constrain:
                    tok = self._rule(tok, res["bbox3d"], decoded_boxes, int(prev_box[pos - 1033]), pos)
                if forced is not None:
                    tok = int(forced[mod][len(res[mod])])
                res[mod].append(tok)
                t = torch.tensor([tok])
                if mod == "map":
                    nxt = self._gmlp(t, "map")[None]
                elif mod == "image":
                    nxt = self._gmlp(t, "img")[None]
                else:
                    nxt = w["transformer.be.weight"][t][None]
            x_in = nxt + cond[:, pos:pos + 1]
        if tr is not None:
            tr.setdefault("logits", []).append({m: (np.stack(v) if v else np.zeros((0, 0), np.float32)) for m, v in logit_trace.items()})
        return {m: np.asarray(v, dtype=np.int64) for m, v in res.items()}

    def _sample_bbox(self, lg, cond_row, pos, prev_box, control_slots, seed, frame_idx, u):
        """bbox3d branch of sample_next_token (UMGen.py:1071-1104)."""
        cfg = self.cfg
        w = self.w
        tok = self.sample(lg.clone(), cfg.top_k, cfg.p, u)
        k = pos - 1033                                           # bbox3d_token_id (UMGen.py:1076-1081)
        if control_slots is not None:
            object_id = (pos - 1032) // SLOT_LEN                 # UMGen.py:1084 (category token -> next slot id)
            if object_id in control_slots:
                lt = self._lin(cond_row, "transformer.head_tar_bbox3d", bias=False).clone()
                lt[-1] = float("-inf")
                self._count("control_resample")
                tok = self.sample(lt, cfg.top_k, cfg.p, rng_uniform(seed, frame_idx, pos, DRAW_CONTROL))
        if tok == BBOX_PAD and cfg.merage_ar_tar and int(prev_box[k]) != BBOX_PAD and not cfg.only_ar:
            self._count("pad_avoid")
            lt = self._lin(cond_row, "transformer.head_tar_bbox3d", bias=False)            # UMGen.py:1092-1104
            tok = self.sample(lt, cfg.top_k, cfg.p, rng_uniform(seed, frame_idx, pos, DRAW_PAD_AVOID))
        return tok

    def _rule(self, tok, inferred: List[int], decoded: List[np.ndarray], prev_tok: int, pos: int) -> int:
        """rule_based_constraint (UMGen.py:1275-1383).  Mutates ``inferred`` (blanks the slot) exactly like
        the reference; the KV cache keeps the stale entries (only the re-embedded current token is fed on)."""
        if tok == BBOX_PAD or (pos - 1032) % SLOT_LEN != 0:
            return tok
        slot = inferred[-(SLOT_LEN - 1):] + [tok]
        box = decode_box_values(np.asarray(slot))
        if len(decoded) == 0:
            decoded.append(np.array(EGO_BOX, dtype=np.float64))
        decoded.append(box)
        collision = check_collision(decoded)
        newborn = prev_tok == BBOX_PAD
        self._count("rule_checked")
        self._count("rule_collision" if collision else "rule_free")
        if (newborn and collision) or (len(decoded) > 30 and newborn):
            self._count("rule_blanked")
            for j in range(1, SLOT_LEN):
                inferred[-j] = BBOX_PAD
            decoded.pop()
            return BBOX_PAD
        return tok

    # ---- rollout driver (UMGen.inference, UMGen.py:1542-1671) -----------------------------------
    def inference(self, new_frames: int, cond_frames: int, input_cond_tokens: Dict[str, np.ndarray],
                  input_cond_frames: int = -1, init_tokens: Optional[Dict[str, np.ndarray]] = None,
                  control_test: bool = False, seed: int = 0, trace: bool = False,
                  forced: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, np.ndarray]:
        """Returns dict mod -> int64 [1, input_cond_frames + new_frames, S_mod] (B = 1, like the reference).
        ``forced``: teacher forcing -- dict mod -> [new_frames, S_mod] tokens that replace the sampled ones."""
        if input_cond_frames == -1:
            input_cond_frames = cond_frames
        self.trace = {} if trace else None
        tt = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.long).clone()  # noqa: E731
        out = {m: tt(input_cond_tokens[m])[:, :input_cond_frames] for m in MOD_ORDER}
        cond = {m: out[m].clone() for m in MOD_ORDER}
        init = {k: tt(v) for k, v in init_tokens.items()} if init_tokens is not None else None
        with torch.no_grad():
            for idx in range(new_frames):
                if cond["pose"].shape[1] > cond_frames:
                    cond = {m: cond[m][:, -cond_frames:].clone() for m in MOD_ORDER}
                cur = None
                if init is not None:
                    cur = {k: (v[:, idx:idx + 1].clone() if idx < v.shape[1] else None) for k, v in init.items()}
                    if "pose" in cur and cur["pose"] is None:
                        init, control_test, cur = None, False, None
                fr = {m: forced[m][idx] for m in MOD_ORDER} if forced is not None else None
                res = self._frame(cond, dict(cur) if cur is not None else None, control_test, seed, idx, fr)
                for m in MOD_ORDER:
                    new = torch.from_numpy(res[m]).view(1, 1, -1)
                    if init is not None and m in init and not (control_test and m == "bbox3d"):
                        new = cur[m]
                    cond[m] = torch.cat([cond[m], new], dim=1)
                    out[m] = torch.cat([out[m], new], dim=1)
        return {m: out[m].numpy() for m in MOD_ORDER}
