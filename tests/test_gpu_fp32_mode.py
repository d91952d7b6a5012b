"""GPU: math_mode = 1, the fp32 parity mode (pf_engine_config.math_mode; DESIGN.md "fp32 parity mode").

The reference runs the exported graphs through onnxruntime in fp32 (OfflineModel.cs:41-57).  The default engine path
feeds the matrix cores f16 operands, which is good for ~1e-2 on log-probs; this mode keeps every activation and every
weight in fp32 (v_mfma_f32_32x32x2_f32, fp32 softmax / LayerNorm / CIF) so that the only differences from the fp32
oracle are summation order.  Covers the paraformer and SenseVoice graphs AND the two heads of configs[4] (BiCIF
timestamps, SeACo bias decoder).  Tolerance: 2e-4 abs on log-probs of magnitude ~10 (fp32 accumulation order over K <= 2048,
through 50 + 16 layers measured 9.5e-6), token_num / L identical, ids identical wherever the oracle's top-1/top-2 margin
exceeds 1e-3.
"""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import glue
from oracle import model as om

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(params=[1, 3], ids=["fp32_mfma", "exact_x3"])
def mode(request):
    """1 = fp32 operands on v_mfma_f32_32x32x2_f32; 3 = the same graph with every large Linear as three f16 MFMA products of
    (hi, 2^11 lo) operand pairs — 22 mantissa bits — and the fp32-MFMA flash attention (round 5).  Same tolerances."""
    return request.param


def _speech(audio, cmvn):
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    T = max(f.shape[0] for f in feats)
    return fe.pad_sequence(feats).reshape(len(audio), T, 560)


def _ids_match(res, ref_logits, margin=1e-3):
    ids_ref = om.argmax_last(ref_logits)
    srt = np.sort(ref_logits, axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > margin
    assert safe.mean() > 0.9
    np.testing.assert_array_equal(res.token_ids[safe], ids_ref[safe])
    return float((res.token_ids == ids_ref).mean())


@pytest.mark.parametrize("variant", ["loop", "cumsum"])
def test_fp32_mode_small_paraformer(variant, mode):
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=515)     # V not a multiple of 4
    cfg["cif_variant"] = variant
    w = W.synth_weights(cfg, seed=33)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=mode)
    audio = [W.synth_audio(n, 5 + u) for u, n in enumerate((48000, 30000, 41000))]
    speech = _speech(audio, cmvn)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32").paraformer(speech)
    for res in (eng.forward_feats(speech, want_logits=True), eng.recognize(audio, want_logits=True)):
        np.testing.assert_array_equal(res.token_num, ref["token_num"])
        assert res.logits.shape == ref["logits"].shape
        err = np.abs(res.logits - ref["logits"]).max()
        assert err < TOL, err
        _ids_match(res, ref["logits"])
    ids_only = eng.recognize(audio)
    np.testing.assert_array_equal(ids_only.token_ids, res.token_ids)
    eng.close()


def test_fp32_mode_full_depth_is_closer_than_f16_mode(mode):
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config()
    w = W.synth_weights(cfg, seed=42)
    cmvn = W.synth_cmvn()
    blob = W.pack_pfw(cfg, w)
    audio = [W.synth_audio(80000, u) for u in range(2)]
    speech = _speech(audio, cmvn)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32", fast=True).paraformer(speech)
    e32 = Engine(weights=blob, cmvn=cmvn, device=0, math_mode=mode)
    r32 = e32.forward_feats(speech, want_logits=True)
    e32.close()
    e16 = Engine(weights=blob, cmvn=cmvn, device=0)
    r16 = e16.forward_feats(speech, want_logits=True)
    e16.close()
    np.testing.assert_array_equal(r32.token_num, ref["token_num"])
    err32 = np.abs(r32.logits - ref["logits"]).max()
    err16 = np.abs(r16.logits - ref["logits"]).max()
    print(f"full depth: |fp32 mode - oracle| = {err32:.2e}, |f16 mode - oracle| = {err16:.2e}")
    assert err32 < 5e-4, err32                       # Oracle(fast=True) uses torch kernels: a second summation order
    assert err32 * 10 < err16
    agree = _ids_match(r32, ref["logits"])
    assert agree > 0.99


def test_fp32_mode_above_the_short_input_threshold(mode, sv_embed):
    """Row counts above the short-input threshold (M > 512 for encoder AND decoder): where math_mode 3 takes its one-launch
    products (K-loop wrap), the fused Q | K | V product, LayerNorm / attention writing operand pairs and the pair epilogue of the
    FFN hidden — the forms the benchmark runs, on models small enough for the oracle to follow here.  Same bars as the small
    models."""
    from aliparaformerasr_amd.engine import Engine
    cmvn = W.synth_cmvn()
    cfg = W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=515)
    w = W.synth_weights(cfg, seed=35)
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=mode)
    audio = [W.synth_audio(n, 15 + u) for u, n in enumerate((480000, 470000, 400000, 480000, 333000, 480000))]
    speech = _speech(audio, cmvn)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32", fast=True).paraformer(speech)
    res = eng.forward_feats(speech, want_logits=True)
    assert speech.shape[0] * speech.shape[1] > 2048 and res.token_ids.shape[0] * res.token_ids.shape[1] > 512
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    err = np.abs(res.logits - ref["logits"]).max()
    assert err < 5e-4, err                           # (fast=True: torch kernels, a second summation order)
    _ids_match(res, ref["logits"])
    eng.close()
    # SenseVoice: prompt rows in front, two encoder stacks, CTC head over every frame
    cfg = W.sensevoice_small_config(enc_layers=3, tp_layers=2, vocab=403)
    w = W.synth_weights(cfg, seed=9)
    w["embed.weight"] = sv_embed.astype(np.float32)
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=mode)
    conf = fe.FrontendConf(dither=0.0)
    audio = [W.synth_audio(n, 70 + u) for u, n in enumerate((160000, 150000, 160000, 120000, 160000, 160000, 99000, 160000))]
    feats = [glue.sensevoice_prepend(fe.wav_frontend(a, conf, cmvn[0], cmvn[1]), sv_embed, use_itn=True) for a in audio]
    T = max(f.shape[0] for f in feats)
    speech = fe.pad_sequence(feats).reshape(len(audio), T, 560)
    assert speech.shape[0] * T > 1200
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32", fast=True).sensevoice(speech)
    res = eng.forward_feats(speech, want_logits=True)
    err = np.abs(res.logits - ref["logits"]).max()
    assert err < 5e-4, err
    _ids_match(res, ref["logits"])
    eng.close()


def test_fp32_mode_sensevoice(sv_embed, mode):
    from aliparaformerasr_amd.engine import Engine
    cfg = W.sensevoice_small_config(enc_layers=3, tp_layers=2, vocab=403)
    w = W.synth_weights(cfg, seed=9)
    w["embed.weight"] = sv_embed.astype(np.float32)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=mode)
    conf = fe.FrontendConf(dither=0.0)
    audio = [W.synth_audio(n, 70 + u) for u, n in enumerate((32000, 24000))]
    feats = [glue.sensevoice_prepend(fe.wav_frontend(a, conf, cmvn[0], cmvn[1]), sv_embed, use_itn=True) for a in audio]
    T = max(f.shape[0] for f in feats)
    speech = fe.pad_sequence(feats).reshape(2, T, 560)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32").sensevoice(speech)
    res = eng.forward_feats(speech, want_logits=True)
    assert res.logits.shape == ref["logits"].shape and res.L == T
    err = np.abs(res.logits - ref["logits"]).max()
    assert err < TOL, err
    _ids_match(res, ref["logits"])
    eng.close()


def test_fp32_mode_bicif_timestamp_head(mode):
    """configs[4]'s timestamp head in fp32 (ConvTranspose1d, BiLSTM, Linear(1024, 1), renormalisation, cif_wo_hidden):
    us_cif_peak within 2e-4 of the fp32 oracle modulo the integrator reset, the SAME fire frames wherever the oracle's
    crossing clears the threshold by 1e-3, recognizer-side timestamps (integer milliseconds) identical on those."""
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, vocab=256, timestamp_head=True)
    w = W.synth_weights(cfg, seed=1)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=mode)
    audio = [W.synth_audio(n, 11 + u) for u, n in enumerate((48000, 36000))]
    speech = _speech(audio, cmvn)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32").paraformer(speech)
    res = eng.recognize(audio, want_logits=True)
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    assert np.abs(res.logits - ref["logits"]).max() < TOL
    T3 = 3 * speech.shape[1]
    assert res.cif_peak.shape == (2, T3)
    thr = np.float32(np.float32(1.0) - np.float32(1e-4))
    d = np.abs(res.cif_peak - ref["us_cif_peak"])
    d = np.minimum(d, np.abs(d - thr))                    # a fire decided the other way shifts the integrator by thr
    assert d.max() < 2e-4, d.max()
    n_clear = 0
    for b in range(2):
        pk = ref["us_cif_peak"][b]
        f_ref = np.nonzero(pk > thr)[0]
        f_dev = np.nonzero(res.cif_peak[b] > thr)[0]
        clear = np.asarray([min(pk[f] - thr, thr - (pk[f - 1] if (f > 0 and pk[f - 1] <= thr) else 0.0)) > 1e-3 for f in f_ref])
        assert len(f_dev) == len(f_ref)
        np.testing.assert_array_equal(f_dev[clear], f_ref[clear])
        n_clear += int(clear.sum())
    assert n_clear >= 10
    eng.close()


def test_exact_mode_timestamp_recurrence_with_two_utterance_tiles():
    """math_mode 3 runs the BiCIF head's BiLSTM as ONE persistent launch with (hi, lo') pair operands (lstm_ring_kernel<true>).  40
    utterances = two tiles of 32 per direction (256 workgroups: the whole device, the last tile ragged): token counts identical and
    peaks within the fp32 mode's bar of the fp32 oracle's."""
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128, timestamp_head=True)
    w = W.synth_weights(cfg, seed=4)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=3)
    audio = [W.synth_audio(9000 + 700 * (u % 7), 40 + u) for u in range(40)]
    speech = _speech(audio, cmvn)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32").paraformer(speech)
    res = eng.recognize(audio)
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    thr = np.float32(np.float32(1.0) - np.float32(1e-4))
    d = np.abs(res.cif_peak - ref["us_cif_peak"])
    d = np.minimum(d, np.abs(d - thr))
    assert d.max() < 2e-4, d.max()
    eng.close()


def test_fp32_mode_seaco_bias_decoder(mode):
    """configs[4]'s SeACo branch in fp32: hotword embedder (Embedding + 2 x LSTM), the bias decoder on [CIF embeds ;
    decoder hidden], hotword_output_layer, NO-BIAS merge — merged log-probs within 2e-4 of the fp32 oracle on every
    row whose NO-BIAS decision is not a near-tie, ids identical off the near-ties; + the timestamp head of the same model."""
    from aliparaformerasr_amd.engine import Engine
    cfg = W.seaco_paraformer_config(enc_layers=2, dec_layers=2, seaco_layers=2, vocab=300, seaco_nobias=290)
    w = W.synth_weights(cfg, seed=6)
    w["seaco.output.bias"][290] += 2.6          # random heads never pick NO-BIAS: lift it so that both sides of the merge occur
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=mode)
    audio = [W.synth_audio(n, 40 + u) for u, n in enumerate((32000, 48000, 40000))]
    speech = _speech(audio, cmvn)
    hw = np.asarray(glue.pad_list([[11, 12], [100, 200, 30], [7, 8, 9, 10], [1]]), np.int32)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32").seaco(speech, hw)
    res = eng.recognize(audio, want_logits=True, hotwords=hw)
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    assert res.logits.shape == ref["logits"].shape
    dha = ref["dha_logits"]
    nb = cfg["seaco_nobias"]
    other = np.where(np.arange(dha.shape[-1])[None, None, :] == nb, -np.inf, dha).max(-1)
    clear = np.abs(dha[..., nb] - other) > 1e-3
    assert clear.mean() > 0.9
    err = np.abs(res.logits - ref["logits"]).max(-1)
    assert err[clear].max() < TOL, err[clear].max()
    margin = np.sort(ref["logits"], axis=-1)
    safe = clear & ((margin[..., -1] - margin[..., -2]) > 1e-3)
    np.testing.assert_array_equal(res.token_ids[safe], om.argmax_last(ref["logits"])[safe])
    took_hotword_rows = (np.argmax(dha, -1) != nb)
    assert 0.1 < took_hotword_rows.mean() < 0.9, took_hotword_rows.mean()      # both sides of the merge are exercised
    d = np.abs(res.cif_peak - ref["us_cif_peak"])
    d = np.minimum(d, np.abs(d - 0.9999))
    assert d.max() < 2e-4, d.max()
    # no hotwords: the bias branch is skipped, the ASR rows come back
    r0 = eng.recognize(audio, want_logits=True, hotwords=np.zeros((0, 10), np.int32))
    assert np.abs(r0.logits - ref["asr_logits"]).max() < TOL
    eng.close()


# ---- operator level: one Linear / one FFN block of the fp32 graph exactly as Engine::gemm32 launches them ----------------
# Reference: float64 products of the fp32 inputs.  What the bars separate: an f16-operand product is off by ~2^-11 per
# term (5e-4 at these magnitudes, measured 1.5e-3 max), a product of (hi, lo') pairs — 22 mantissa bits — by ~2^-22 per term
# plus the fp32 accumulation (measured < 3e-6); fp32 operands (mode 1) the accumulation alone.
def _tiny_engine(mode):
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=64)
    return Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, seed=1)), cmvn=W.synth_cmvn(), device=0, math_mode=mode)


@pytest.mark.parametrize("M,N,K", [(1000, 512, 512), (700, 512, 560), (1100, 2048, 512), (600, 8404, 512), (5344, 512, 2048), (300, 512, 512)])
def test_linear32_operator_keeps_22_bits(mode, M, N, K):
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    Wt = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    eng = _tiny_engine(mode)
    tol = 1e-5 * max(1.0, K / 512.0) ** 0.5            # fp32 accumulation grows with sqrt(K): 1.02e-5 measured at K = 2048 on the fp32 matrix path
    for relu, resid in ((False, None), (True, None), (False, r)):
        y = eng.op_linear32(x, Wt, b, resid=resid, relu=relu)
        ref = x.astype(np.float64) @ Wt.astype(np.float64).T + b
        if resid is not None:
            ref = ref + resid
        if relu:
            ref = np.maximum(ref, 0.0)
        err = np.abs(y - ref).max()
        assert err < tol, (relu, resid is not None, err)
    # a second call with other weights at the same scratch address must not meet a cached operand image
    y2 = eng.op_linear32(x, -Wt, b)
    assert np.abs(y2 - (x.astype(np.float64) @ (-Wt).astype(np.float64).T + b)).max() < tol
    eng.close()


@pytest.mark.parametrize("M", [1100, 4000, 200])
def test_ffn32_block_hidden_travels_as_an_operand_pair(mode, M):
    """x + relu(x W1^T + b1) W2^T + b2 with the encoder's shapes: above the short-input threshold math_mode 3 never stores the
    hidden in fp32 — the first product's epilogue writes the (hi, lo') pair the second consumes.  A hidden that lost its lo'
    half (f16-rounded) shows up as ~5e-4 here."""
    D, F = 512, 2048
    rng = np.random.default_rng(M)
    x = rng.standard_normal((M, D)).astype(np.float32)
    W1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    W2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    b1 = rng.standard_normal(F).astype(np.float32)
    b2 = rng.standard_normal(D).astype(np.float32)
    eng = _tiny_engine(mode)
    y = eng.op_ffn32(x, W1, b1, W2, b2)
    x64 = x.astype(np.float64)
    # the hidden is an fp32 tensor in the graph: round it where the graph does
    h = np.maximum(x64 @ W1.astype(np.float64).T + b1, 0.0)
    ref = x64 + h @ W2.astype(np.float64).T + b2
    err = np.abs(y - ref).max()
    assert err < 2e-5, err
    eng.close()
