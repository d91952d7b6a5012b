"""GPU: the kernel VARIANTS the benchmark's pipeline actually launches, at the benchmark's row count
(M = 32 x 500 = 16 000), each called through the C ABI and compared with an independent numpy / oracle
statement on the same seeded inputs:

  * gemm_f16_pp3<kind, MI> for kind 1 (f16 row-major), 2 (fp32 + residual + FSMN addend), 3 (f16 blocked
    layout) x MI 1 / 2 (128- / 256-row tiles), incl. the blocked A operand and the q-column scaling;
  * the FFN-up (blocked out) -> FFN-down (blocked in) hand-off as enc_layer() runs it;
  * fsmn_enc_kernel<11> (f16 V slice with row stride 3D) and fsmn_dec_kernel<11 / 21>;
  * the vocabulary tail: log-softmax + last-index arg-max over the LOG-PROBS (what
    AliParaformerAsr/OfflineRecognizer.cs:139-152 scans), both kernel variants (row in registers / re-read).

Tolerances: f16 x f16 products are exact in fp32, so GEMM results differ from the float64 product of the
f16-rounded operands only by fp32 accumulation order (1e-4 rel) plus, for f16 results, one f16 rounding
(2^-11 rel); index work is bit-exact.
"""
import numpy as np
import pytest
import torch

from aliparaformerasr_amd import weights as W
from oracle import model as om

pytestmark = pytest.mark.gpu

M_BENCH = 16000


@pytest.fixture(scope="module")
def eng():
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
    w = W.synth_weights(cfg, seed=5)
    e = Engine(weights=W.pack_pfw(cfg, w), cmvn=W.synth_cmvn(), device=0)
    yield e
    e.close()


def h16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def _ref(A, Wm, bias):
    # float32 BLAS product of the f16-rounded operands (products exact, accumulation order differs)
    return h16(A) @ h16(Wm).T + bias


@pytest.mark.parametrize("tile_rows", [128, 256])
def test_gemm_kind1_f16_rowmajor_with_q_scale(eng, tile_rows):
    """QKV projection shape: [16000 x 512] x [512 x 1536], q columns (first 512) scaled by 1/sqrt(128)."""
    rng = np.random.default_rng(100 + tile_rows)
    M, N, K = M_BENCH, 1536, 512
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    Wm += (np.arange(N)[:, None] * 1e-4).astype(np.float32)            # asymmetric: detects transposes
    bias = rng.standard_normal(N).astype(np.float32)
    sc = np.float32(128 ** -0.5)
    ref = _ref(A, Wm, bias)
    ref[:, :512] *= sc
    got = eng.op_gemm_ex(A, Wm, bias, out_kind=1, tile_rows=tile_rows, scale_cols=512, scale=float(sc))
    np.testing.assert_allclose(got, ref, rtol=1.5e-3, atol=1.5e-3)
    assert np.array_equal(got, eng.op_gemm_ex(A, Wm, bias, out_kind=1, tile_rows=tile_rows, scale_cols=512, scale=float(sc)))


@pytest.mark.parametrize("tile_rows", [128, 256])
def test_gemm_kind2_fp32_residual_and_fsmn_addend(eng, tile_rows):
    """Out-projection shape: [16000 x 512] x [512 x 512] + bias + fp32 residual + fp32 FSMN memory."""
    rng = np.random.default_rng(200 + tile_rows)
    M, N, K = M_BENCH, 512, 512
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32) * 3
    add2 = rng.standard_normal((M, N)).astype(np.float32)
    ref = _ref(A, Wm, bias) + add2 + resid
    got = eng.op_gemm_ex(A, Wm, bias, resid=resid, add2=add2, out_kind=0, tile_rows=tile_rows)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=3e-4)
    got2 = eng.op_gemm_ex(A, Wm, bias, resid=resid, out_kind=0, tile_rows=tile_rows)
    np.testing.assert_allclose(got2, _ref(A, Wm, bias) + resid, rtol=1e-4, atol=3e-4)


@pytest.mark.parametrize("tile_rows", [128, 256])
def test_gemm_kind3_blocked_output(eng, tile_rows):
    """FFN-up shape: [16000 x 512] x [512 x 2048] + bias + ReLU into the blocked activation layout (de-blocked by
    the host side of the op with index arithmetic written independently of the kernel's)."""
    rng = np.random.default_rng(300 + tile_rows)
    M, N, K = M_BENCH, 2048, 512
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    Wm += (np.arange(N)[:, None] * 1e-4).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = np.maximum(_ref(A, Wm, bias), 0)
    got = eng.op_gemm_ex(A, Wm, bias, relu=True, out_kind=2, tile_rows=tile_rows)
    np.testing.assert_allclose(got, ref, rtol=1.5e-3, atol=1.5e-3)


@pytest.mark.parametrize("tile_rows", [128, 256])
def test_gemm_blocked_a_operand(eng, tile_rows):
    """FFN-down shape: blocked A [16000 x 2048] x [2048 x 512] + bias + fp32 residual (K = 2048: 32 k-steps)."""
    rng = np.random.default_rng(400 + tile_rows)
    M, N, K = M_BENCH, 512, 2048
    A = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    A += (np.arange(K)[None, :] * 1e-4).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    ref = _ref(A, Wm, bias) + resid
    got = eng.op_gemm_ex(A, Wm, bias, resid=resid, out_kind=0, a_blocked=True, tile_rows=tile_rows)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=6e-4)




def test_gemm_ragged_edges_all_kinds(eng):
    """M and N that are not multiples of the tile, every kind x tile height."""
    rng = np.random.default_rng(77)
    for (M, N, K) in ((5344, 512, 512), (333, 576, 2048), (5000, 2048, 512), (83, 1536, 576)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        ref = _ref(A, Wm, bias)
        for tr in (128, 256):
            np.testing.assert_allclose(eng.op_gemm_ex(A, Wm, bias, out_kind=0, tile_rows=tr), ref, rtol=1e-4, atol=6e-4)
            np.testing.assert_allclose(eng.op_gemm_ex(A, Wm, bias, out_kind=1, tile_rows=tr), ref, rtol=1.5e-3, atol=1.5e-3)
            np.testing.assert_allclose(eng.op_gemm_ex(A, Wm, bias, out_kind=2, relu=True, tile_rows=tr), np.maximum(ref, 0),
                                       rtol=1.5e-3, atol=1.5e-3)


def test_ffn_blocked_handoff_at_bench_rows(eng):
    """FFN-up (kind 3, blocked out) -> FFN-down (blocked A, kind 2) exactly as enc_layer() chains them."""
    rng = np.random.default_rng(9)
    M, D, F = M_BENCH, 512, 2048
    x = rng.standard_normal((M, D)).astype(np.float32)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(F)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(D)).astype(np.float32)
    resid = rng.standard_normal((M, D)).astype(np.float32) * 2
    h = h16(np.maximum(h16(x) @ h16(w1).T + b1, 0))                   # the hidden is stored as f16
    ref = h @ h16(w2).T + b2 + resid
    got = eng.op_ffn(x, w1, b1, w2, b2, resid)
    # hidden values that sit on an f16 rounding boundary may round the other way (fp32 accumulation order):
    # each flips one operand by 2^-11 relative
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-3)
    assert np.abs(got - ref).mean() < 1e-4


@pytest.mark.parametrize("M", [16000, 10880, 64000, 2048 + 37, 64, 1])
def test_ffn_fused_kernel(eng, M):
    """The whole encoder FFN block + the next LayerNorm in ONE launch (k_ffn.hip, round 5) at the row counts of the three
    BASELINE workloads (32 x 500, 64 x 170, 128 x 500) and ragged / tiny ones.  Reference: fp64 products of the f16-rounded
    operands with the hidden rounded to f16 where the kernel rounds it (after bias + ReLU), then the exact LayerNorm."""
    rng = np.random.default_rng(90 + M % 7)
    D, F = 512, 2048
    x = rng.standard_normal((M, D)).astype(np.float32)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(F)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(D)).astype(np.float32)
    resid = (2 * rng.standard_normal((M, D))).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    be = (0.1 * rng.standard_normal(D)).astype(np.float32)
    rows = np.unique(np.concatenate([np.arange(min(M, 96)), rng.integers(0, M, 256), np.arange(max(0, M - 70), M)]))
    xs = h16(x[rows]).astype(np.float64)
    h = h16(np.maximum(xs @ h16(w1).T.astype(np.float64) + b1, 0).astype(np.float32)).astype(np.float64)
    ref = h @ h16(w2).T.astype(np.float64) + b2 + resid[rows]
    mu = ref.mean(-1, keepdims=True)
    ln = (ref - mu) / np.sqrt(((ref - mu) ** 2).mean(-1, keepdims=True) + 1e-12) * g + be
    got_x, got_n = eng.op_ffn_fused(x, w1, b1, w2, b2, resid, ln=(g, be))
    # hidden values on an f16 rounding boundary may round the other way (fp32 accumulation order): each flips one
    # operand of the second product by 2^-11 relative
    np.testing.assert_allclose(got_x[rows], ref, rtol=2e-4, atol=2e-3)
    assert np.abs(got_x[rows] - ref).mean() < 1e-4
    np.testing.assert_allclose(got_n[rows], ln, rtol=2e-3, atol=2e-3)          # f16 result
    # no residual, no LayerNorm: the bare block
    got2, none = eng.op_ffn_fused(x, w1, b1, w2, b2)
    assert none is None
    np.testing.assert_allclose(got2[rows], ref - resid[rows], rtol=2e-4, atol=2e-3)
    # and against the two-launch form the pipeline used before (same rounding points)
    if M <= 16000:
        np.testing.assert_allclose(got_x, eng.op_ffn(x, w1, b1, w2, b2, resid), rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("M,splits", [(5344, 0), (5344, 1), (5344, 2), (5344, 4), (5344, 8), (21376, 0), (1300, 0), (777, 3), (64, 8), (5, 0)])
def test_dec_ffn_fused_kernel(eng, M, splits):
    """The decoder's FFN block — w_1 + ReLU, LayerNorm over the 2048 hidden columns, w_2 without bias — and the LayerNorm behind
    it, as the split form of the fused FFN kernel + its finishing pass (k_ffn.hip): the hidden LayerNorm is
    applied AFTER the second product from row statistics collected on the way.  Row counts: the benchmark's decoder (32 x 167),
    batch 128, short and ragged ones; every split count.  Reference: fp64 products of the f16-rounded operands, the hidden
    rounded to f16 where the kernel rounds it (after bias + ReLU), the exact LayerNorm on it, gamma (.) W2 rounded to f16."""
    rng = np.random.default_rng(300 + M % 11 + splits)
    D, F = 512, 2048
    x = rng.standard_normal((M, D)).astype(np.float32)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(F)).astype(np.float32)
    gf = (1 + 0.2 * rng.standard_normal(F)).astype(np.float32)
    bf = (0.1 * rng.standard_normal(F)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    be = (0.1 * rng.standard_normal(D)).astype(np.float32)
    rows = np.unique(np.concatenate([np.arange(min(M, 96)), rng.integers(0, M, 256), np.arange(max(0, M - 70), M)]))
    xs = h16(x[rows]).astype(np.float64)
    h = h16(np.maximum(xs @ h16(w1).T.astype(np.float64) + b1, 0).astype(np.float32)).astype(np.float64)
    mu = h.mean(-1, keepdims=True)
    hn = (h - mu) / np.sqrt(((h - mu) ** 2).mean(-1, keepdims=True) + 1e-12)
    # the kernel multiplies the un-normalised hidden with f16(gamma (.) W2): the same sum, rounded at a different place
    ref = (hn * gf) @ w2.T.astype(np.float64) + bf.astype(np.float64) @ w2.T.astype(np.float64)
    ref16 = hn @ h16(w2 * gf[None, :]).T.astype(np.float64) + bf.astype(np.float64) @ w2.T.astype(np.float64)
    m2 = ref16.mean(-1, keepdims=True)
    ln = (ref16 - m2) / np.sqrt(((ref16 - m2) ** 2).mean(-1, keepdims=True) + 1e-12) * g + be
    t, n = eng.op_dec_ffn_fused(x, w1, b1, (gf, bf), w2, ln=(g, be), splits=splits)
    np.testing.assert_allclose(t[rows], ref16, rtol=2e-4, atol=2e-3)
    assert np.abs(t[rows] - ref16).mean() < 1e-4
    np.testing.assert_allclose(t[rows], ref, rtol=2e-3, atol=2e-2)              # against unrounded gamma (.) W2: f16 weight rounding only
    np.testing.assert_allclose(n[rows], ln, rtol=1e-3, atol=1e-3)
    t2, none = eng.op_dec_ffn_fused(x, w1, b1, (gf, bf), w2, splits=splits)
    assert none is None
    np.testing.assert_array_equal(t2, t)                                        # deterministic: fixed summation order of the shares


@pytest.mark.parametrize("M,splits", [(5344, 0), (5344, 1), (5344, 2), (5344, 8), (1300, 0), (777, 3), (200, 4), (5, 0)])
def test_dec_out_ffn_fused_kernel(eng, M, splits):
    """The same launch with the previous decoder layer's cross-attention out-projection in front (k_ffn.hip, OP = 2):
    x = resid + ctx Wo^T + bo (written by the tile's first share), LayerNorm norm1 of it stays in LDS as the block's operand.
    Reference: fp64 products of the f16-rounded operands, the norm1 result and the hidden rounded to f16 where the kernel
    rounds them."""
    rng = np.random.default_rng(400 + M % 13 + splits)
    D, F = 512, 2048
    ctx = rng.standard_normal((M, D)).astype(np.float32)
    wo = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    bo = (0.1 * rng.standard_normal(D)).astype(np.float32)
    resid = (2 * rng.standard_normal((M, D))).astype(np.float32)
    g1 = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    be1 = (0.1 * rng.standard_normal(D)).astype(np.float32)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(F)).astype(np.float32)
    gf = (1 + 0.2 * rng.standard_normal(F)).astype(np.float32)
    bf = (0.1 * rng.standard_normal(F)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    be = (0.1 * rng.standard_normal(D)).astype(np.float32)
    rows = np.unique(np.concatenate([np.arange(min(M, 96)), rng.integers(0, M, 256), np.arange(max(0, M - 70), M)]))
    xr = h16(ctx[rows]).astype(np.float64) @ h16(wo).T.astype(np.float64) + bo + resid[rows]
    m1 = xr.mean(-1, keepdims=True)
    xn = h16(((xr - m1) / np.sqrt(((xr - m1) ** 2).mean(-1, keepdims=True) + 1e-12) * g1 + be1).astype(np.float32)).astype(np.float64)
    h = h16(np.maximum(xn @ h16(w1).T.astype(np.float64) + b1, 0).astype(np.float32)).astype(np.float64)
    mu = h.mean(-1, keepdims=True)
    hn = (h - mu) / np.sqrt(((h - mu) ** 2).mean(-1, keepdims=True) + 1e-12)
    ref = hn @ h16(w2 * gf[None, :]).T.astype(np.float64) + bf.astype(np.float64) @ w2.T.astype(np.float64)
    m2 = ref.mean(-1, keepdims=True)
    ln = (ref - m2) / np.sqrt(((ref - m2) ** 2).mean(-1, keepdims=True) + 1e-12) * g + be
    t, n, xo = eng.op_dec_ffn_fused(None, w1, b1, (gf, bf), w2, ln=(g, be), splits=splits, out_proj=(ctx, wo, bo, resid, (g1, be1)))
    np.testing.assert_allclose(xo[rows], xr, rtol=1e-5, atol=2e-5)
    # norm1 values on an f16 rounding boundary may round the other way: each flips one operand of the first product by 2^-11
    np.testing.assert_allclose(t[rows], ref, rtol=2e-3, atol=1e-2)
    assert np.abs(t[rows] - ref).mean() < 3e-4
    np.testing.assert_allclose(n[rows], ln, rtol=5e-3, atol=5e-3)
    # the operand tile equals what the two-step path feeds the block: same result as the plain form on LayerNorm(x_out)
    m1a = xo.astype(np.float64).mean(-1, keepdims=True)
    xa = ((xo - m1a) / np.sqrt(((xo - m1a) ** 2).mean(-1, keepdims=True) + 1e-12) * g1 + be1).astype(np.float32)
    t2, _ = eng.op_dec_ffn_fused(xa, w1, b1, (gf, bf), w2, splits=splits)
    np.testing.assert_allclose(t, t2, rtol=2e-3, atol=1e-2)


@pytest.mark.parametrize("B,T", [(32, 500), (64, 170), (5, 83 + 40), (3, 9)])
def test_attn_out_ffn_fused_kernel(eng, B, T):
    """Two thirds of an encoder layer in ONE launch (k_ffn.hip, OP = 1): out-projection + bias + residual + FSMN memory of
    the V slice + LayerNorm norm2 (its result stays in LDS) + the FFN block + the next LayerNorm.  Reference: fp64 products of
    the f16-rounded operands, the norm2 result and the hidden rounded to f16 where the kernel rounds them."""
    rng = np.random.default_rng(190 + T)
    M, D, F = B * T, 512, 2048
    ctx = rng.standard_normal((M, D)).astype(np.float32)
    v = rng.standard_normal((M, D)).astype(np.float32)
    wo = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    bo = (0.1 * rng.standard_normal(D)).astype(np.float32)
    fw = (0.1 * rng.standard_normal((D, 11))).astype(np.float32)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(F)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(D)).astype(np.float32)
    resid = (2 * rng.standard_normal((M, D))).astype(np.float32)
    g2, be2, g, be = [(s + 0.1 * rng.standard_normal(D)).astype(np.float32) for s in (1, 0, 1, 0)]

    def ln_(x, gg, bb):
        mu = x.mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(((x - mu) ** 2).mean(-1, keepdims=True) + 1e-12) * gg + bb
    # rows: whole utterances at the start / end + random ones (the FSMN window needs neighbours, so the reference runs on
    # complete utterances)
    utts = sorted(set([0, B - 1] + list(rng.integers(0, B, 3))))
    fs = om.fsmn(torch.from_numpy(h16(v).reshape(B, T, D)[utts]), torch.from_numpy(fw), 11).numpy().astype(np.float64)
    rows = np.concatenate([np.arange(u * T, (u + 1) * T) for u in utts])
    xmid = h16(ctx[rows]).astype(np.float64) @ h16(wo).T.astype(np.float64) + bo + resid[rows] + fs.reshape(-1, D)
    a = h16(ln_(xmid, g2, be2).astype(np.float32)).astype(np.float64)
    hid = h16(np.maximum(a @ h16(w1).T.astype(np.float64) + b1, 0).astype(np.float32)).astype(np.float64)
    ref = xmid + hid @ h16(w2).T.astype(np.float64) + b2
    got_x, got_n = eng.op_attn_ffn_fused(ctx, wo, bo, v, fw, T, (g2, be2), w1, b1, w2, b2, resid=resid, ln=(g, be))
    # the norm2 result and the hidden are f16 operands of the next product: a value on a rounding boundary may round the
    # other way (fp32 vs fp64 accumulation), each flips one operand by 2^-11 relative
    np.testing.assert_allclose(got_x[rows], ref, rtol=3e-4, atol=4e-3)
    assert np.abs(got_x[rows] - ref).mean() < 2e-4
    np.testing.assert_allclose(got_n[rows], ln_(ref, g, be), rtol=3e-3, atol=3e-3)
    # ... and with the NEXT layer's Q | K | V projection behind it in the same launch (Q scaled, Q | K through the blocked
    # layout, V row-major): the products of the f16 LayerNorm result the launch returns
    wqkv = (rng.standard_normal((3 * D, D)) / np.sqrt(D)).astype(np.float32)
    bqkv = (0.1 * rng.standard_normal(3 * D)).astype(np.float32)
    gx, gn, gq, gk, gv = eng.op_attn_ffn_fused(ctx, wo, bo, v, fw, T, (g2, be2), w1, b1, w2, b2, resid=resid, ln=(g, be), qkv=(wqkv, bqkv))
    np.testing.assert_array_equal(gx, got_x)
    qkv_ref = got_n[rows].astype(np.float64) @ h16(wqkv).T.astype(np.float64) + bqkv
    qkv_ref[:, :D] *= np.float32(1.0 / np.sqrt(128.0))
    np.testing.assert_allclose(np.concatenate([gq[rows], gk[rows], gv[rows]], 1), qkv_ref, rtol=2e-3, atol=2e-3)
    # no residual (the first encoder layer)
    got0, _ = eng.op_attn_ffn_fused(ctx, wo, bo, v, fw, T, (g2, be2), w1, b1, w2, b2)
    xmid0 = xmid - resid[rows]
    a0 = h16(ln_(xmid0, g2, be2).astype(np.float32)).astype(np.float64)
    hid0 = h16(np.maximum(a0 @ h16(w1).T.astype(np.float64) + b1, 0).astype(np.float32)).astype(np.float64)
    np.testing.assert_allclose(got0[rows], xmid0 + hid0 @ h16(w2).T.astype(np.float64) + b2, rtol=3e-4, atol=4e-3)


def test_fsmn_enc_kernel_f16_strided(eng):
    rng = np.random.default_rng(10)
    for (B, T) in ((32, 500), (2, 83), (3, 7), (1, 1), (2, 166)):
        v = rng.standard_normal((B, T, 512)).astype(np.float32)
        w = (0.1 * rng.standard_normal((512, 11))).astype(np.float32)
        ref = om.fsmn(torch.from_numpy(h16(v)), torch.from_numpy(w), 11).numpy()
        got = eng.op_fsmn_enc(v, w)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


def test_fsmn_dec_kernel(eng):
    rng = np.random.default_rng(11)
    for (B, L, k) in ((32, 167, 11), (2, 25, 11), (64, 150, 21), (1, 3, 21), (3, 40, 11)):
        tn = rng.standard_normal((B, L, 512)).astype(np.float32)
        x = rng.standard_normal((B, L, 512)).astype(np.float32)
        w = (0.1 * rng.standard_normal((512, k))).astype(np.float32)
        n = rng.integers(0, L + 1, B).astype(np.int32)
        n[0] = L
        mask = (np.arange(L)[None, :] < n[:, None]).astype(np.float32)[..., None]
        ref = x + om.fsmn(torch.from_numpy(tn), torch.from_numpy(w), k, torch.from_numpy(mask)).numpy()
        got = eng.op_fsmn_dec(tn, w, n, x)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------- vocabulary tail
def test_logsoftmax_argmax_scans_the_log_probs(eng):
    rng = np.random.default_rng(12)
    for V in (8404, 25055, 512, 37, 1):
        x = (rng.standard_normal((300, V)) * 2).astype(np.float32)
        x[5, :] = 0.25                               # all equal -> V-1
        if V > 100:
            x[7, 50] = x[7].max() + 1                # clear winner
            x[9, 10] = x[9, 90] = x[9].max() + 2     # exact tie in the logits -> later index
        y, ids = eng.op_logsoftmax_argmax(x)
        ref = om.log_softmax(torch.from_numpy(x)).numpy()
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-5)
        # index work is bit-exact ON THE TENSOR THE DEVICE RETURNS (= what the reference loop would scan)
        np.testing.assert_array_equal(ids, om.argmax_last(y))
        # ids-only variant of the kernel (log-probs formed on the fly, never stored): identical ids
        np.testing.assert_array_equal(eng.op_logsoftmax_argmax(x, store=False), ids)
        assert ids[5] == V - 1
        if V > 100:
            assert ids[7] == 50 and ids[9] == 90


def test_logsoftmax_argmax_collision_kat(eng):
    """Two logits one ulp apart whose log-probs are the SAME float32: the raw arg-max says the earlier index,
    the reference (scanning log-probs, ties -> larger index) says the later one."""
    V = 8404
    x = np.full((4, V), -1.0, np.float32)
    hi = np.nextafter(np.float32(0.5), np.float32(1.0))
    x[:, 3] = hi          # the larger logit, earlier
    x[:, 7] = 0.5         # one ulp smaller, later
    x[1, 7] = hi; x[1, 3] = 0.5                      # mirrored: larger one later anyway
    x[2, 7] = np.float32(0.6)                        # clear margin: no collision
    assert (np.argmax(x[0]) == 3) and x[0, 3] > x[0, 7]
    y, ids = eng.op_logsoftmax_argmax(x)
    ref = om.log_softmax(torch.from_numpy(x)).numpy()
    # lse ~ 8.1: ulp(y) = 9.5e-7 >> ulp(x) = 6e-8, so the two log-probs collide in the oracle as well
    assert ref[0, 3] == ref[0, 7] and y[0, 3] == y[0, 7]
    np.testing.assert_array_equal(ids, om.argmax_last(y))
    np.testing.assert_array_equal(ids, om.argmax_last(ref))
    assert list(ids) == [7, 7, 7, 7]
    np.testing.assert_array_equal(eng.op_logsoftmax_argmax(x, store=False), ids)


def test_logsoftmax_argmax_adversarial_rows(eng):
    """Many near-tie rows: flat distributions (lse >> |x|) with clusters of logits a few ulps apart."""
    rng = np.random.default_rng(13)
    R, V = 20000, 512
    base = rng.uniform(-0.5, 0.5, (R, 1)).astype(np.float32)
    x = np.repeat(base, V, axis=1)
    steps = rng.integers(0, 3, (R, V)).astype(np.int32)
    xi = x.view(np.int32) + steps * np.sign(x).astype(np.int32)        # 0..2 ulps away from the base value
    x = xi.view(np.float32).copy()
    y, ids = eng.op_logsoftmax_argmax(x)
    np.testing.assert_array_equal(ids, om.argmax_last(y))
    np.testing.assert_array_equal(eng.op_logsoftmax_argmax(x, store=False), ids)
    raw = om.argmax_last(x)
    assert (raw != ids).mean() > 0.2          # the raw-logit arg-max really is a different function here


def test_logsoftmax_argmax_nan_and_inf(eng):
    V = 300
    x = np.random.default_rng(14).standard_normal((6, V)).astype(np.float32)
    x[0, 17] = np.nan          # NaN poisons the sum -> every log-prob NaN -> the loop ends on V-1
    x[1, :] = -np.inf          # -inf - (-inf) = NaN everywhere
    x[2, 5] = np.inf           # inf - inf = NaN
    x[3, 100] = -np.inf        # harmless
    y, ids = eng.op_logsoftmax_argmax(x)
    np.testing.assert_array_equal(ids, om.argmax_last(y))
    assert ids[0] == V - 1 and ids[1] == V - 1 and ids[2] == V - 1
    assert np.isnan(y[0]).all() and np.isneginf(y[3, 100])
    np.testing.assert_array_equal(eng.op_logsoftmax_argmax(x, store=False), ids)


# ---------------------------------------------------------------- short-input GEMM (k_gemm_small.hip)
@pytest.mark.parametrize("M,N,K", [(83, 1536, 560), (83, 512, 2048), (23, 2048, 512), (23, 515, 512), (300, 512, 1536),
                                   (129, 8404, 512), (1, 512, 512), (500, 2048, 512), (500, 512, 2048)])
def test_gemm_small(eng, M, N, K):
    """The GEMM the 1 x 5 s path runs (M = 83 encoder rows / 23 decoder rows): every epilogue the pipeline asks of it,
    K not a multiple of 64 on the caller's side (560 -> 576 padded), N not a multiple of 32 / 4, more than one row
    tile, the split form (K > 576: partials + row-wise reduction) and bit-identical results over repeated launches."""
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    Wm += (np.arange(N)[:, None] * 1e-4).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    add2 = rng.standard_normal((M, N)).astype(np.float32)
    ref = _ref(A, Wm, bias)
    got = eng.op_gemm_ex(A, Wm, bias, resid=resid, add2=add2, out_kind=0, tile_rows=32)
    np.testing.assert_allclose(got, ref + add2 + resid, rtol=2e-4, atol=2e-4)
    for _ in range(3):
        assert np.array_equal(got, eng.op_gemm_ex(A, Wm, bias, resid=resid, add2=add2, out_kind=0, tile_rows=32))
    got = eng.op_gemm_ex(A, Wm, bias, relu=True, out_kind=0, tile_rows=32)
    np.testing.assert_allclose(got, np.maximum(ref, 0), rtol=2e-4, atol=2e-4)
    sc_cols = 64 * min(N // 64, 8) if K <= 576 else 0             # the q scaling exists on K = 512 projections only
    sc = np.float32(128 ** -0.5)
    r2 = ref.copy()
    r2[:, :sc_cols] *= sc
    got = eng.op_gemm_ex(A, Wm, bias, out_kind=1, tile_rows=32, scale_cols=sc_cols, scale=float(sc))
    np.testing.assert_allclose(got, r2, rtol=1.5e-3, atol=1.5e-3)
    # the dispatcher picks it by itself for short inputs, and the persistent kernel agrees with it
    auto = eng.op_gemm_ex(A, Wm, bias, out_kind=1, scale_cols=sc_cols, scale=float(sc))
    assert np.array_equal(auto, got)
    big = eng.op_gemm_ex(A, Wm, bias, out_kind=1, tile_rows=128, scale_cols=sc_cols, scale=float(sc))
    np.testing.assert_allclose(big, got, rtol=1.5e-3, atol=1.5e-3)


# ---------------------------------------------------------------- row-complete GEMM (k_gemm_rc.hip)
def _ln_ref(x, g, b):
    return om.layer_norm(torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(b)).numpy()


def _fsmn_ref(v, w, T):
    M = v.shape[0]
    out = np.zeros_like(v)
    for lo in range(0, M, T):                       # utterances are runs of T rows (the last one may be short)
        seg = h16(v[lo:lo + T])[None]
        out[lo:lo + T] = om.fsmn(torch.from_numpy(seg), torch.from_numpy(w), w.shape[1]).numpy()[0]
    return out


def test_gemm_rc_out_projection_with_fsmn_and_layernorm(eng):
    """Attention output projection as enc_layer() launches it: [16000 x 512] x [512 x 512] + bias + fp32 residual +
    FSMN(11 taps over the f16 V slice, zero padding at the 32 utterance edges) and the LayerNorm that follows."""
    rng = np.random.default_rng(500)
    M, K, T = M_BENCH, 512, 500
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((512, K)) / np.sqrt(K)).astype(np.float32)
    Wm += (np.arange(512)[:, None] * 1e-4).astype(np.float32)
    bias = rng.standard_normal(512).astype(np.float32)
    resid = (rng.standard_normal((M, 512)) * 3).astype(np.float32)
    v = rng.standard_normal((M, 512)).astype(np.float32)
    fw = (0.1 * rng.standard_normal((512, 11))).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(512)).astype(np.float32)
    b = (0.1 * rng.standard_normal(512)).astype(np.float32)
    x_ref = _ref(A, Wm, bias) + resid + _fsmn_ref(v, fw, T)
    x, n16, n32 = eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, fsmn_v=v, fsmn_w=fw, T=T, ln=(g, b))
    np.testing.assert_allclose(x, x_ref, rtol=1e-4, atol=5e-4)
    n_ref = _ln_ref(x, g, b)                        # LayerNorm of the x the device produced: isolates the LN stage
    np.testing.assert_allclose(n32, n_ref, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(n16, n_ref, rtol=1.5e-3, atol=1.5e-3)
    np.testing.assert_array_equal(n16, h16(n32))    # the f16 output is the rounded fp32 output
    # first layer: no residual
    x0, _, _ = eng.op_gemm_rc(A, Wm, bias=bias, fsmn_v=v, fsmn_w=fw, T=T)
    np.testing.assert_allclose(x0, x_ref - resid, rtol=1e-4, atol=5e-4)


def test_gemm_rc_ffn_down_blocked_a(eng):
    """FFN down-projection: blocked A [16000 x 2048] x [2048 x 512] + bias + residual, then the NEXT LayerNorm."""
    rng = np.random.default_rng(501)
    M, K = M_BENCH, 2048
    A = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    A += (np.arange(K)[None, :] * 1e-4).astype(np.float32)
    Wm = (rng.standard_normal((512, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(512).astype(np.float32)
    resid = rng.standard_normal((M, 512)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(512)).astype(np.float32)
    b = (0.1 * rng.standard_normal(512)).astype(np.float32)
    x, n16, n32 = eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, ln=(g, b), a_blocked=True)
    np.testing.assert_allclose(x, _ref(A, Wm, bias) + resid, rtol=1e-4, atol=6e-4)
    np.testing.assert_allclose(n32, _ln_ref(x, g, b), rtol=2e-5, atol=2e-5)
    x2, _, _ = eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, a_blocked=False)
    np.testing.assert_array_equal(x2, x)            # the operand layout does not change the arithmetic


# ---------------------------------------------------------------- fused Q | K | V projection (k_gemm_qkv.hip) + attention on its layout
def test_qkv_split_kernel_and_attention_on_the_blocked_layout(eng):
    """The persistent 256 x 192 kernel writes Q (scaled) and K in the blocked layout and V row-major; the attention kernel
    reads that layout.  Both must reproduce the row-major kernels bit for bit (same products, same summation order), at the
    benchmark's shape (32 x 500: utterances start at rows that are not multiples of 32 or 64), at layer 0's depth (K = 560,
    padded to 576), and at ragged shapes (T not a multiple of the 64-key tile, M not a multiple of 256)."""
    rng = np.random.default_rng(520)
    for B, T, K in ((32, 500, 512), (3, 77, 560), (5, 333, 512), (1, 2000, 512), (64, 171, 512)):
        M = B * T
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((1536, K)) / np.sqrt(K)).astype(np.float32)
        bias = rng.standard_normal(1536).astype(np.float32)
        q, k, v, ctx = eng.op_qkv_attention(x, w, bias, B, T)
        ref = eng.op_gemm_ex(x, w, bias, out_kind=1, tile_rows=256, scale_cols=512, scale=float(1.0 / np.sqrt(128.0)))
        np.testing.assert_array_equal(q, ref[:, :512], err_msg=f"Q B={B} T={T} K={K}")
        np.testing.assert_array_equal(k, ref[:, 512:1024], err_msg=f"K B={B} T={T} K={K}")
        np.testing.assert_array_equal(v, ref[:, 1024:], err_msg=f"V B={B} T={T} K={K}")
        np.testing.assert_allclose(ref, h16(_ref(x, w, bias) * np.r_[np.full(512, 1 / np.sqrt(128.0)), np.ones(1024)].astype(np.float32)),
                                   rtol=2e-3, atol=2e-3)
        ctx_ref = eng.op_attention(q.reshape(B, T, 512), k.reshape(B, T, 512), v.reshape(B, T, 512), 4).reshape(M, 512)
        np.testing.assert_array_equal(ctx, ctx_ref, err_msg=f"attention B={B} T={T}")




def test_gemm_rc_ragged_shapes_and_utterance_edges(eng):
    """M not a multiple of 64 / of T, utterances shorter than a tile, T = 8 (every wave straddles an utterance edge),
    one k-step only, and rows whose LayerNorm is ill-conditioned in fp32 (large common offset)."""
    rng = np.random.default_rng(502)
    for (M, K, T) in ((83, 512, 83), (166, 64, 83), (1000, 576, 40), (333, 2048, 8), (64, 512, 9), (5000, 512, 5000)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        Wm = (rng.standard_normal((512, K)) / np.sqrt(K)).astype(np.float32)
        bias = rng.standard_normal(512).astype(np.float32)
        resid = rng.standard_normal((M, 512)).astype(np.float32)
        resid[M // 2] += 300.0                          # mean >> spread on one row
        v = rng.standard_normal((M, 512)).astype(np.float32)
        fw = (0.1 * rng.standard_normal((512, 11))).astype(np.float32)
        g = (1 + 0.1 * rng.standard_normal(512)).astype(np.float32)
        b = (0.1 * rng.standard_normal(512)).astype(np.float32)
        x_ref = _ref(A, Wm, bias) + resid + _fsmn_ref(v, fw, T)
        x, n16, n32 = eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, fsmn_v=v, fsmn_w=fw, T=T, ln=(g, b))
        np.testing.assert_allclose(x, x_ref, rtol=1e-4, atol=6e-4)
        np.testing.assert_allclose(n32, _ln_ref(x, g, b), rtol=3e-5, atol=3e-5)


# ---------------------------------------------------------------- persistent 256 x 256 tile GEMM (k_gemm_big.hip)
@pytest.mark.parametrize("M,N,K", [(16000, 2048, 512), (5000, 512, 576), (300, 256, 128), (40000, 256, 192)])
def test_gemm_big_persistent_blocked(eng, M, N, K):
    """The persistent 256 x 256-tile kernel (blocked f16 result, FFN-up): the benchmark's shape (504 tiles on 256 CUs:
    two tiles per workgroup, the second one's operands and bias prefetched under the first), fewer tiles than CUs, the
    minimum K, and a run of tiles per workgroup with a k loop of six steps (tile ends inside the ring's depth)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    Wm += (np.arange(N)[:, None] * 1e-4).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = np.maximum(_ref(A, Wm, bias), 0)
    got = eng.op_gemm_ex(A, Wm, bias, relu=True, out_kind=2, tile_rows=1024)
    np.testing.assert_allclose(got, ref, rtol=1.5e-3, atol=1.5e-3)
    assert np.array_equal(got, eng.op_gemm_ex(A, Wm, bias, relu=True, out_kind=2, tile_rows=1024))
    nob = eng.op_gemm_ex(A, Wm, None, relu=False, out_kind=2, tile_rows=1024)
    np.testing.assert_allclose(nob, _ref(A, Wm, 0.0), rtol=1.5e-3, atol=1.5e-3)
    pp3 = eng.op_gemm_ex(A, Wm, bias, relu=True, out_kind=2, tile_rows=256)
    np.testing.assert_allclose(got, pp3, rtol=1.5e-3, atol=1.5e-3)


def test_gemm_big_refuses_what_it_cannot_do(eng):
    from aliparaformerasr_amd._native import PfError
    with pytest.raises(PfError):                                     # row-major result: not this kernel's layout
        eng.op_gemm_ex(np.zeros((300, 512), np.float32), np.zeros((512, 512), np.float32), None, out_kind=1, tile_rows=1024)


@pytest.mark.parametrize("M,T,K", [(83, 83, 512), (166, 83, 512), (23, 23, 2048), (300, 100, 2048), (7, 7, 512)])
def test_gemm_small_fsmn_epilogue_and_layernorm_in_the_reduction(eng, M, T, K):
    """The two fused forms of the short-input path: K = 512 — attention out-projection with the 11-tap FSMN memory of
    the V slice and the residual as epilogue terms (then the LayerNorm kernel); K = 2048 — FFN-down as split partials
    whose reduction adds bias + residual and applies the LayerNorm behind the block.  Against the same float64 / fp32
    statement the row-complete kernel is tested with, and against that kernel itself."""
    rng = np.random.default_rng(M + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((512, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(512).astype(np.float32)
    resid = rng.standard_normal((M, 512)).astype(np.float32)
    g = (1.0 + 0.1 * rng.standard_normal(512)).astype(np.float32)
    b = (0.1 * rng.standard_normal(512)).astype(np.float32)
    x_ref = _ref(A, Wm, bias) + resid
    kw = {}
    if K <= 576:
        v = rng.standard_normal((M, 512)).astype(np.float32)
        fw = (0.2 * rng.standard_normal((512, 11))).astype(np.float32)
        vq = h16(v).reshape(M // T, T, 512)
        mem = vq.copy()
        for j in range(11):
            sh = j - 5
            lo, hi = max(0, -sh), min(T, T - sh)
            if lo < hi:
                mem[:, lo:hi] += fw[:, j][None, None, :] * vq[:, lo + sh:hi + sh]
        x_ref = _ref(A, Wm, bias) + mem.reshape(M, 512) + resid
        kw = dict(fsmn_v=v, fsmn_w=fw, T=T)
    x, n16, n32 = eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, ln=(g, b), short_input=True, **kw)
    np.testing.assert_allclose(x, x_ref, rtol=3e-4, atol=3e-4)
    xd = x_ref.astype(np.float64)
    mu = xd.mean(-1, keepdims=True)
    ln_ref = ((xd - mu) / np.sqrt(((xd - mu) ** 2).mean(-1, keepdims=True) + 1e-12) * g + b).astype(np.float32)
    np.testing.assert_allclose(n32, ln_ref, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(n16, ln_ref, rtol=3e-3, atol=3e-3)
    if T >= 8:
        x2, _, m32 = eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, ln=(g, b), **kw)       # the row-complete kernel
        np.testing.assert_allclose(x, x2, rtol=3e-4, atol=3e-4)
        np.testing.assert_allclose(n32, m32, rtol=1e-3, atol=1e-3)
    again = eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, ln=(g, b), short_input=True, **kw)
    assert np.array_equal(again[0], x) and np.array_equal(again[2], n32)


def test_decoder_middle_in_one_launch(monkeypatch):
    """Round 6: finishing pass of the split FFN + norm2 + FSMN memory + residual + norm3 + q-projection as ONE launch
    (k_decmid.hip, `PF_DEC_MID`) against the three launches it replaces, on the same engine inputs: ragged utterances (token_num
    below L, L not a multiple of the kernel's 32-row blocks, an utterance shorter than the FSMN half-window), log-probs equal up to
    the f16 rounding of q (measured: bit-identical — both forms accumulate the q product over K in the same order and round at the same
    points), token_num / L / ids identical; and against the oracle with the engine's rounding points."""
    from aliparaformerasr_amd.engine import Engine
    from oracle import frontend as fe, model as om
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=3, vocab=300)
    w = W.synth_weights(cfg, seed=19)
    w["predictor.out.bias"] = np.asarray([-0.2], np.float32)          # ~0.45 per frame: L well above 32 for the long utterances
    cmvn = W.synth_cmvn()
    lens = [16000 * 24, 16000 * 9, 16000 * 17, 4000, 16000 * 30, 16000 * 13] * 3
    audio = [W.synth_audio(n, 300 + u) for u, n in enumerate(lens)]
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("PF_DEC_MID", flag)
        eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
        out[flag] = eng.recognize(audio, want_logits=True)
        ids_only = eng.recognize(audio)
        np.testing.assert_array_equal(ids_only.token_ids, out[flag].token_ids)
        eng.close()
    a, b = out["0"], out["1"]
    assert a.L == b.L and a.L > 64 and a.L % 32 != 0 and int(a.token_num.min()) < a.L - 32   # (every row is padded to Tmax, quirk Q2: no short rows)
    np.testing.assert_array_equal(a.token_num, b.token_num)
    d = float(np.abs(a.logits - b.logits).max())
    print("decoder middle fused vs three launches: L = %d, token_num %d .. %d, max |d log-prob| %.3e" % (a.L, a.token_num.min(), a.token_num.max(), d))
    assert d < 5e-3, d
    srt = np.sort(a.logits, axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 2e-2
    np.testing.assert_array_equal(a.token_ids[safe], b.token_ids[safe])
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(x, conf, cmvn[0], cmvn[1]) for x in audio]
    speech = fe.pad_sequence(feats).reshape(len(audio), -1, 560)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").paraformer(speech)
    np.testing.assert_array_equal(b.token_num, ref["token_num"])
    err = np.abs(b.logits - ref["logits"])             # 30 s rows: a few CIF fires move by a frame under 16-bit operands
    k = err.size - max(1, err.size // 1000)
    assert float(err.max()) < 2e-1 and float(np.partition(err.reshape(-1), k)[k]) < 3e-2, (float(err.max()),)
