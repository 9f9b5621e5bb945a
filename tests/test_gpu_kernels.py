"""-m gpu: each hand-written HIP kernel against an fp64/fp32 CPU restatement of the reference op, at the
production width (E=768, H=16, head_dim 48) and with ragged sizes (S=2207 is not a multiple of any tile)."""
import numpy as np
import pytest
import torch

from tests.gpu_util import bits16, check, fp, from_bits16, lib, round16, vp

PREC = [0, 1, 2]          # fp32 | bf16 | fp16 operands (precision code of the debug hooks)
EPS16 = {1: 1.0, 2: 0.125}   # 16-bit tolerances are quoted for bf16 (8 mantissa bits); fp16 has 11

pytestmark = pytest.mark.gpu


def ref_linear(act, W, bias, gelu, resid):
    o = act.astype(np.float64) @ W.astype(np.float64).T
    if bias is not None:
        o = o + bias.astype(np.float64)
    if gelu:
        o = torch.nn.functional.gelu(torch.from_numpy(o)).numpy()
    if resid is not None:
        o = o + resid.astype(np.float64)
    return o


@pytest.mark.parametrize("R,N,K,gelu,resid", [(300, 2304, 768, 0, 0), (2207, 768, 3072, 0, 1), (513, 3072, 768, 1, 0),
                                               (97, 288, 96, 0, 0), (130, 96, 384, 0, 1),
                                               # large token counts, ragged last tile
                                               (8300, 768, 768, 0, 1), (8448, 2304, 768, 0, 0), (9000, 1536, 3072, 1, 0),
                                               # >= 1024 output tiles: the persistent kernel (ragged last token tile, every epilogue)
                                               (22001, 768, 768, 0, 1), (12000, 3072, 768, 1, 0), (11003, 768, 3072, 0, 1)])
@pytest.mark.parametrize("bf16", PREC)
def test_linear(bf16, R, N, K, gelu, resid):
    rng = np.random.default_rng(R + N + K)
    act = rng.standard_normal((R, K), dtype=np.float32)
    W = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal((N,), dtype=np.float32) * 0.1
    x0 = rng.standard_normal((R, N), dtype=np.float32) if resid else None
    if bf16:
        a_in, w_in = bits16(act, bf16), bits16(W, bf16)
        act, W = round16(act, bf16), round16(W, bf16)
    else:
        a_in, w_in = act, W
    out = x0.copy() if resid else np.zeros((R, N), dtype=np.uint16 if bf16 else np.float32)
    check(lib().umgen_dbg_linear(bf16, vp(a_in), vp(w_in), fp(bias), R, N, K, gelu, resid, vp(out)))
    got = out if (resid or not bf16) else from_bits16(out, bf16)
    ref = ref_linear(act, W, bias, gelu, x0)
    # fp32 path: summation-order noise only; bf16 path: exact products, fp32 accumulate, (bf16 output rounding when stored)
    tol = 2e-5 if not bf16 else (2e-5 if resid else 1.2e-2 * EPS16[bf16])
    np.testing.assert_allclose(got, ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=0)


@pytest.mark.parametrize("R,N,K,gelu,resid", [(2207, 768, 768, 0, 1), (8300, 768, 768, 0, 1), (8448, 2304, 768, 0, 0), (9000, 1536, 3072, 1, 0),
                                               (600, 256, 128, 0, 0), (5000, 3072, 768, 1, 0), (257, 768, 3072, 0, 1), (70000, 768, 768, 0, 1),
                                               (3000, 512, 256, 0, 0)])
@pytest.mark.parametrize("prec", [1, 2])
def test_linear_256_tile_kernel(prec, R, N, K, gelu, resid):
    """gemm256.hip (256 x 256 x 64 tiles, 8-slot LDS ring, one counted wait per k-tile) forced on shapes the launcher would give to
    the 128-tile kernels too: ragged token counts, the shortest legal K (two k-tiles), many tiles per workgroup (the ring runs on
    across output tiles), every epilogue.  Bit-identical to the 128-tile kernels (same k order inside and across the MFMAs)."""
    rng = np.random.default_rng(R + N + K)
    act = rng.standard_normal((R, K), dtype=np.float32)
    W = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal((N,), dtype=np.float32) * 0.1
    x0 = rng.standard_normal((R, N), dtype=np.float32) if resid else None
    a_in, w_in = bits16(act, prec), bits16(W, prec)
    act, W = round16(act, prec), round16(W, prec)
    outs = []
    for flag in (16, 32):      # 16: force the 256-tile kernel, 32: 128-tile kernels only
        out = x0.copy() if resid else np.zeros((R, N), dtype=np.uint16)
        check(lib().umgen_dbg_linear(prec | flag, vp(a_in), vp(w_in), fp(bias), R, N, K, gelu, resid, vp(out)))
        outs.append(out)
    got = outs[0] if resid else from_bits16(outs[0], prec)
    ref = ref_linear(act, W, bias, gelu, x0)
    tol = 2e-5 if resid else 1.2e-2 * EPS16[prec]
    np.testing.assert_allclose(got, ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=0)
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("F,S,N,K,chunk", [(3, 2207, 768, 768, None), (2, 300, 768, 128, None), (5, 256, 768, 256, None), (1, 2207, 768, 768, None),
                                            (4, 1031, 768, 768, "2100"), (2, 7, 768, 128, None), (3, 513, 1536, 256, None)])
@pytest.mark.parametrize("prec", [1, 2])
def test_linear_256_tile_kernel_v_transposed(monkeypatch, prec, F, S, N, K, chunk):
    """The spatial attention's V^T ([frame][feature][padded tokens], GEMM_VT) on the 256-tile kernel: token tiles per frame, the frame's
    ragged last tile, pad columns written as zeros, frames as launch chunks (hook) -- bit-identical to the 128-tile kernel's rows, and
    equal to the fp64 product within the 16-bit rounding.  (N is a whole number of 48-wide heads and of 256-row tiles: 768, 1536.)"""
    rng = np.random.default_rng(F + S + N + K)
    act = rng.standard_normal((F * S, K), dtype=np.float32)
    W = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal((N,), dtype=np.float32) * 0.1
    a_in, w_in = bits16(act, prec), bits16(W, prec)
    S_pad = (S + 63) // 64 * 64
    outs = []
    for flag in (16, 32):
        if chunk and flag == 16:
            monkeypatch.setenv("UMGEN_DEBUG_GEMM256_MAX_ROWS", chunk)
        else:
            monkeypatch.delenv("UMGEN_DEBUG_GEMM256_MAX_ROWS", raising=False)
        out = np.full((F, N, S_pad), 0x7fc0 if prec == 1 else 0x7e00, dtype=np.uint16)       # (the hook clears it: pad columns must come back zero)
        check(lib().umgen_dbg_linear_vt(prec | flag, vp(a_in), vp(w_in), fp(bias), F, S, N, K, vp(out)))
        outs.append(out)
    np.testing.assert_array_equal(outs[0], outs[1])
    assert not outs[0][:, :, S:].any()
    ref = (round16(act, prec).astype(np.float64) @ round16(W, prec).astype(np.float64).T + bias).reshape(F, S, N).transpose(0, 2, 1)
    got = from_bits16(outs[0], prec)[:, :, :S]
    np.testing.assert_allclose(got, ref, atol=1.2e-2 * EPS16[prec] * max(1.0, np.abs(ref).max()) + 2e-5, rtol=0)


@pytest.mark.parametrize("R,N,K,gelu,resid", [(9000, 768, 3072, 0, 1), (5000, 3072, 768, 1, 0), (2049, 768, 768, 0, 0)])
def test_linear_256_tile_kernel_row_chunks(monkeypatch, R, N, K, gelu, resid):
    """The 256-tile kernel addresses its operands with 32-bit element offsets; more token rows than 2^31 / ld elements (16 scenes' K = 3072
    activations) go out as several launches on whole-tile row chunks (launch_gemm256).  With the chunk size forced down to 2048 rows the
    outputs are bit-identical to the one-launch form and to the 128-tile kernels."""
    rng = np.random.default_rng(R + N + K + 1)
    act = rng.standard_normal((R, K), dtype=np.float32)
    W = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal((N,), dtype=np.float32) * 0.1
    x0 = rng.standard_normal((R, N), dtype=np.float32) if resid else None
    a_in, w_in = bits16(act, 1), bits16(W, 1)
    outs = []
    for flag, chunk in ((16, "2048"), (16, None), (32, None)):
        if chunk:
            monkeypatch.setenv("UMGEN_DEBUG_GEMM256_MAX_ROWS", chunk)
        else:
            monkeypatch.delenv("UMGEN_DEBUG_GEMM256_MAX_ROWS", raising=False)
        out = x0.copy() if resid else np.zeros((R, N), dtype=np.uint16)
        check(lib().umgen_dbg_linear(1 | flag, vp(a_in), vp(w_in), fp(bias), R, N, K, gelu, resid, vp(out)))
        outs.append(out)
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])


@pytest.mark.parametrize("R,N,K,gelu,resid", [(2207, 768, 768, 0, 1), (4500, 2304, 768, 0, 0), (3000, 3072, 768, 1, 0), (1031, 768, 3072, 0, 1),
                                               (8192, 3072, 16, 1, 0), (130, 1028, 768, 0, 0), (257, 768, 770, 0, 0), (300, 132, 37, 0, 0)])
def test_fp32_mfma_gemm_is_the_fma_chain(R, N, K, gelu, resid):
    """fp32 parity mode's GEMM on the matrix cores (gemm_f32_mfma_kernel: v_mfma_f32_32x32x2_f32, 128 x 128 x 32 tiles) against the VALU
    FMA-chain kernel it replaces (flag 32): the instruction is a k-ascending fmaf chain with one rounding per product, so the two are
    BIT-IDENTICAL -- ragged token counts, K = 16 (the GMLP tables), K not a multiple of the slab or of 4, an odd K, every epilogue."""
    rng = np.random.default_rng(R + N + K)
    act = rng.standard_normal((R, K), dtype=np.float32)
    W = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal((N,), dtype=np.float32) * 0.1
    x0 = rng.standard_normal((R, N), dtype=np.float32) if resid else None
    outs = []
    for flag in (0, 32):       # 0: the launcher's choice (matrix cores), 32: the VALU kernel
        out = x0.copy() if resid else np.zeros((R, N), dtype=np.float32)
        check(lib().umgen_dbg_linear(flag, vp(act), vp(W), fp(bias), R, N, K, gelu, resid, vp(out)))
        outs.append(out)
    ref = ref_linear(act, W, bias, gelu, x0)
    np.testing.assert_allclose(outs[0], ref, atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=0)
    np.testing.assert_array_equal(outs[0], outs[1])


def ref_attention(q, k, v, H, causal):
    B, Tq, E = q.shape
    D = E // H
    qh = torch.from_numpy(q).double().view(B, Tq, H, D).permute(0, 2, 1, 3)
    kh = torch.from_numpy(k).double().view(B, -1, H, D).permute(0, 2, 1, 3)
    vh = torch.from_numpy(v).double().view(B, -1, H, D).permute(0, 2, 1, 3)
    att = (qh @ kh.transpose(-1, -2)) * float(np.float32(1.0 / np.sqrt(D)))
    if causal:
        Tk = k.shape[1]
        i = torch.arange(Tq).view(-1, 1)
        j = torch.arange(Tk).view(1, -1)
        att = att.masked_fill(j > i + (Tk - Tq), float("-inf"))
    return (torch.softmax(att, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, E).numpy()


@pytest.mark.parametrize("F,S,H", [(2, 2207, 16), (3, 1031, 2), (1, 70, 2), (2, 1693, 4)])
@pytest.mark.parametrize("bf16", PREC)
def test_attn_spatial(bf16, F, S, H):
    E = H * 48
    rng = np.random.default_rng(S + H)
    q = rng.standard_normal((F, S, E), dtype=np.float32) * 1.5
    k = rng.standard_normal((F, S, E), dtype=np.float32) * 1.5
    v = rng.standard_normal((F, S, E), dtype=np.float32)
    if bf16:
        q, k, v = round16(q, bf16), round16(k, bf16), round16(v, bf16)
    qk = np.ascontiguousarray(np.concatenate([q, k], axis=-1))
    y = np.zeros((F, S, E), dtype=np.uint16 if bf16 else np.float32)
    check(lib().umgen_dbg_attn_spatial(bf16, vp(bits16(qk, bf16) if bf16 else qk), vp(bits16(v, bf16) if bf16 else v), F, S, H, vp(y)))
    got = from_bits16(y, bf16) if bf16 else y
    ref = ref_attention(q, k, v, H, False)
    np.testing.assert_allclose(got, ref, atol=(2e-2 * EPS16[bf16] if bf16 else 2e-5), rtol=0)


@pytest.mark.parametrize("F,S,H", [(2, 2207, 16), (3, 1031, 2), (1, 70, 2), (2, 1693, 4), (1, 33, 2), (1, 129, 2)])
def test_attn_spatial_fp32_matrix_core_kernel(F, S, H):
    """fp32 parity mode's spatial attention on v_mfma_f32_32x32x2_f32 (attn_spatial_f32_mfma_kernel: 32 queries per wave, 32-key tiles,
    the probabilities feed the second product straight from the score registers) and the one-thread-per-query VALU kernel it replaces
    (flag 32): both within 2e-5 of the fp64 reference at production shapes, ragged key / query tails (S % 32 = 1, 2, 6, 29, 31)."""
    E = H * 48
    rng = np.random.default_rng(S + H)
    q = rng.standard_normal((F, S, E), dtype=np.float32) * 1.5
    k = rng.standard_normal((F, S, E), dtype=np.float32) * 1.5
    v = rng.standard_normal((F, S, E), dtype=np.float32)
    qk = np.ascontiguousarray(np.concatenate([q, k], axis=-1))
    ref = ref_attention(q, k, v, H, False)
    for flag in (0, 32):
        y = np.zeros((F, S, E), dtype=np.float32)
        check(lib().umgen_dbg_attn_spatial(flag, vp(qk), vp(v), F, S, H, vp(y)))
        np.testing.assert_allclose(y, ref, atol=2e-5, rtol=0, err_msg=f"flag {flag}")


@pytest.mark.parametrize("B,T,S,H", [(1, 20, 333, 16), (2, 3, 100, 2), (1, 40, 77, 4), (1, 64, 31, 2)])
@pytest.mark.parametrize("bf16", PREC)
def test_attn_temporal(bf16, B, T, S, H):
    E = H * 48
    rng = np.random.default_rng(T + S)
    qkv = rng.standard_normal((B, T, S, 3 * E), dtype=np.float32)
    if bf16:
        qkv = round16(qkv, bf16)
    y = np.zeros((B, T, S, E), dtype=np.uint16 if bf16 else np.float32)
    bits = bits16(qkv, bf16) if bf16 else qkv
    check(lib().umgen_dbg_attn_temporal(bf16, vp(bits), B, T, S, H, 0, vp(y)))
    # slots 0..T-2 ahead of time, the last slot against the slot cache (the rollout's overlapped TAR pass): same bits
    y2 = np.zeros_like(y)
    check(lib().umgen_dbg_attn_temporal(bf16, vp(bits), B, T, S, H, T - 1, vp(y2)))
    np.testing.assert_array_equal(y2, y)
    got = from_bits16(y, bf16) if bf16 else y
    x = qkv.transpose(0, 2, 1, 3).reshape(B * S, T, 3 * E)        # (b s) t c
    ref = ref_attention(np.ascontiguousarray(x[..., :E]), np.ascontiguousarray(x[..., E:2 * E]),
                        np.ascontiguousarray(x[..., 2 * E:]), H, True)
    ref = ref.reshape(B, S, T, E).transpose(0, 2, 1, 3)
    np.testing.assert_allclose(got, ref, atol=(1.6e-2 * EPS16[bf16] if bf16 else 2e-5), rtol=0)


@pytest.mark.parametrize("NQ,L,H", [(1, 1, 16), (1, 7, 16), (3, 2207, 16), (2, 1100, 2)])
@pytest.mark.parametrize("bf16", PREC)
def test_attn_decode(bf16, NQ, L, H):
    E = H * 48
    rng = np.random.default_rng(L)
    q = rng.standard_normal((NQ, E), dtype=np.float32)
    kv = rng.standard_normal((L, 2 * E), dtype=np.float32)
    if bf16:
        kv = round16(kv, bf16)
    y = np.zeros((NQ, E), dtype=np.float32)
    check(lib().umgen_dbg_attn_decode(bf16, fp(q), vp(bits16(kv, bf16) if bf16 else kv), NQ, L, H, fp(y)))
    ref = ref_attention(q[None], np.ascontiguousarray(kv[None, :, :E]), np.ascontiguousarray(kv[None, :, E:]), H, False)[0]
    np.testing.assert_allclose(y, ref, atol=2e-5, rtol=0)


@pytest.mark.parametrize("M,N,K,gelu,ln", [(1, 2304, 768, 0, 1), (3, 3072, 768, 1, 1), (8, 1028, 768, 0, 0), (11, 96, 96, 0, 1)])
@pytest.mark.parametrize("bf16", PREC)
def test_gemv(bf16, M, N, K, gelu, ln):
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K), dtype=np.float32) * 2 + 0.3
    W = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    lw = (1 + 0.1 * rng.standard_normal((K,), dtype=np.float32)).astype(np.float32)
    bias = rng.standard_normal((N,), dtype=np.float32) * 0.1
    if bf16:
        W = round16(W, bf16)
    out = np.zeros((M, N), dtype=np.float32)
    check(lib().umgen_dbg_gemv(bf16, fp(x), fp(lw) if ln else None, vp(bits16(W, bf16) if bf16 else W), fp(bias), M, N, K, gelu, fp(out)))
    xin = torch.from_numpy(x).double()
    if ln:
        xin = torch.nn.functional.layer_norm(xin, (K,), torch.from_numpy(lw).double(), None, 1e-5)
    ref = ref_linear(xin.numpy(), W, bias, gelu, None)
    np.testing.assert_allclose(out, ref, atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=0)


def ref_sample_topk(l, k, temp, u):
    """oracle/umgen_oracle.py OracleUMGen.sample (UMGen.py:899-913 + 967-974): keep the logits >= the k-th largest (ties kept), softmax
    with temperature over the kept set in token order, inverse CDF on the uniform u with sequential fp32 sums."""
    from oracle.umgen_oracle import exp_det
    kk = min(k, l.shape[0])
    kth = np.partition(l, -kk)[-kk]
    idx = np.nonzero(l >= kth)[0]
    z = l[idx] / np.float32(temp)
    e = exp_det(z - z.max())
    c = np.cumsum(e, dtype=np.float32)
    hit = np.nonzero(c > np.float32(u * c[-1]))[0]
    return int(idx[hit[0]]) if hit.size else int(idx[-1])


@pytest.mark.parametrize("V", [8192, 1028, 1024])
@pytest.mark.parametrize("k", [1, 5, 16])
def test_sampler_topk_rows(V, k):
    """The sampler's k-th-largest selection (threshold = k-th largest per-thread maximum, then the exact k-th among the logits above
    it; exhaustive arg-max rounds when more than 64 logits pass the threshold) gives the oracle's token on random rows, on rows with
    ties at the k-th value, and on rows built to take the fallback (one thread's 32 registers hold every large logit)."""
    rng = np.random.default_rng(V * 31 + k)
    n = 96
    L = rng.standard_normal((n, V)).astype(np.float32)
    L[8:16] = np.round(L[8:16] * 4) / 4                      # heavy quantisation: many exact ties, also at the k-th value
    for r in range(16, 40):                                   # the top values sit in FEW threads (indices tid + 256 i share a thread)
        nthr = 1 + (r - 16) // 8                              # 1, 2, 3 threads hold 32 large logits each: 32 / 64 / 96 logits >= T
        for t in range(nthr):
            ii = np.arange(t * 7 + r, V, 256)
            L[r, ii] = 6.0 + rng.random(ii.size).astype(np.float32)
    L[40, :70] = 9.0                                          # 70 equal maxima: more ties than the kept-set buffer's 64 slots -> the exhaustive walk
    L[41, :] = 0.0                                            # a zero-initialised head: every logit ties, the draw is uniform over V (UMGen.py:899-913 keeps every tie)
    L[42, 5::3] = 7.5                                         # ~V / 3 ties at the top, strided over every thread
    L[43, :70] = 9.0
    u = rng.random(n).astype(np.float32)
    u[43] = np.float32(1.0 - 2.0 ** -24)                      # the largest uniform the generator emits: the walk ends on the last tie
    tok = np.zeros(n, np.int32)
    ovf = np.zeros(1, np.int32)
    import ctypes as C
    check(lib().umgen_dbg_sample_topk(fp(L), n, V, k, C.c_float(1.0), fp(u), tok.ctypes.data_as(C.POINTER(C.c_int32)),
                                      ovf.ctypes.data_as(C.POINTER(C.c_int32))))
    assert ovf[0] == 0, "no row is refused any more: ties beyond the 64-slot kept set are SAMPLED"
    for r in range(n):
        assert tok[r] == ref_sample_topk(L[r], k, 1.0, u[r]), (r, int(tok[r]))
    assert tok[40] < 70 and tok[43] < 70
