"""GPU: an operator's result must not depend on what ELSE runs on the device.

Round 3 found a latent race in the persistent GEMM kernel (csrc/k_gemm.hip): the second wave group refilled an LDS
stage slot whose fragment reads were only *timed*, not waited, to have executed — true as long as the workgroup had
its CU to itself, false the moment another kernel's workgroup shared the CU and kept the LDS pipe busy (a second engine
on the same GPU, or this engine's own timestamp head on its side stream): wrong products, up to |d| ~ 2, in EVERY launch
of the 128-row-tile variant beside a running BiLSTM recurrence.  No single-stream test could see it.

Here engine A recognises with a timestamp model (encoder + persistent BiLSTM + decoder) in a loop on its own stream while
engine B repeats single operators at the decoder's shapes; every result must be bit-identical to the quiet one."""
import threading
import time

import numpy as np
import pytest

from aliparaformerasr_amd import weights as W

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_operators_are_bit_stable_beside_a_busy_second_engine():
    from aliparaformerasr_amd.engine import Engine
    cfgA = W.paraformer_large_config(enc_layers=2, dec_layers=1, timestamp_head=True)
    A = Engine(weights=W.pack_pfw(cfgA, W.synth_weights(cfgA, 1)), cmvn=W.synth_cmvn(), device=0)
    cfgB = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=64)
    B = Engine(weights=W.pack_pfw(cfgB, W.synth_weights(cfgB, 2)), cmvn=W.synth_cmvn(), device=0)
    audio = [W.synth_audio(30 * 16000, u) for u in range(32)]
    rng = np.random.default_rng(0)
    Md, D, F, T, L, Bt = 5344, 512, 2048, 500, 167, 32
    x = rng.standard_normal((Md, D)).astype(np.float32)
    h = np.abs(rng.standard_normal((Md, F))).astype(np.float32)
    w1 = (rng.standard_normal((F, D)) / 22).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / 45).astype(np.float32)
    wq = (rng.standard_normal((D, D)) / 22).astype(np.float32)
    b2 = rng.standard_normal(D).astype(np.float32)
    xe = rng.standard_normal((16000, D)).astype(np.float32)
    g = rng.standard_normal(D).astype(np.float32)
    be = rng.standard_normal(D).astype(np.float32)
    q = rng.standard_normal((Bt, L, D)).astype(np.float32)
    k = rng.standard_normal((Bt, T, D)).astype(np.float32)
    v = rng.standard_normal((Bt, T, D)).astype(np.float32)
    wqkv = (rng.standard_normal((3 * D, D)) / 22).astype(np.float32)
    bqkv = rng.standard_normal(3 * D).astype(np.float32)
    he = np.abs(rng.standard_normal((16000, F))).astype(np.float32)
    b1f = rng.standard_normal(F).astype(np.float32)
    fw = (0.1 * rng.standard_normal((D, 11))).astype(np.float32)
    ops = {
        "fp32-result GEMM 5344x512x2048, 128-row tiles": lambda: B.op_gemm_ex(h, w2, None, tile_rows=128),
        "fp32-result GEMM 5344x512x2048, 256-row tiles": lambda: B.op_gemm_ex(h, w2, None, tile_rows=256),
        "fp32 + residual GEMM 5344x512x512, 128-row tiles": lambda: B.op_gemm_ex(x, wq, b2, resid=x, tile_rows=128),
        "f16-result GEMM 5344x512x512, 128-row tiles": lambda: B.op_gemm_ex(x, wq, b2, out_kind=1, tile_rows=128),
        "f16-result GEMM 5344x2048x512": lambda: B.op_gemm_ex(x, w1, None, relu=True, out_kind=1),
        "cross-attention 32x167x500": lambda: B.op_attention(q, k, v),
        "blocked-result GEMM 16000x2048x512 (persistent 256x256 tiles)": lambda: B.op_gemm_ex(xe, w1, None, relu=True, out_kind=2),
        "row-complete GEMM + residual + LayerNorm 16000x512x512": lambda: B.op_gemm_rc(xe, wq, b2, resid=xe, ln=(g, be))[0],
        "self-attention 8x500x500": lambda: B.op_attention(k[:8], k[:8], v[:8]),
        # round 4: the fused Q|K|V projection on 256 x 192 tiles + attention on its blocked layout, the row-complete FFN-down
        # (blocked A, K = 2048) with its LayerNorm, and the split-K pair form (an in-launch exchange between workgroups)
        "Q|K|V 256x192 kernel + attention on the blocked layout 32x500": lambda: np.concatenate(B.op_qkv_attention(xe, wqkv, bqkv, 32, 500), axis=1),
        "row-complete FFN-down + LayerNorm 16000x512x2048 (blocked A)": lambda: np.concatenate(
            B.op_gemm_rc(he, w2, b2, resid=xe, ln=(g, be), a_blocked=True)[::2], axis=1),
        # round 5: the fused FFN block (weights streamed straight into registers, the hidden in LDS) and the same launch with
        # the attention out-projection + FSMN + norm2 in front of it
        "fused FFN block + LayerNorm 16000 rows (k_ffn.hip)": lambda: np.concatenate(
            B.op_ffn_fused(xe, w1, b1f, w2, b2, resid=xe, ln=(g, be)), axis=1),
        "out-projection + FSMN + norm2 + FFN + LayerNorm 32x500 (k_ffn.hip, OP = 1)": lambda: np.concatenate(
            B.op_attn_ffn_fused(xe, wq, b2, xe, fw, 500, (g, be), w1, b1f, w2, b2, resid=xe, ln=(g, be)), axis=1),
    }
    quiet = {name: f() for name, f in ops.items()}
    stop = []
    def disturb():
        while not stop:
            A.recognize(audio)
    th = threading.Thread(target=disturb)
    th.start()
    try:
        time.sleep(1.0)
        for name, f in ops.items():
            t0, n = time.time(), 0
            while time.time() - t0 < 2.0 or n < 6:
                y = f()
                n += 1
                assert np.array_equal(y, quiet[name]), "%s: run %d beside the busy engine differs from the quiet run by %.3g" % (
                    name, n, float(np.abs(y - quiet[name]).max()))
    finally:
        stop.append(1)
        th.join()
    A.close()
    B.close()


@pytest.mark.timeout(600)
def test_timestamp_head_beside_the_decoder_changes_nothing():
    """The BiCIF head runs on its own stream beside the decoder (Engine::forward): ids and peaks must equal the
    single-stream order's, call after call (PF_TS_STREAM is read once per process: compare through the ring / counter forms
    of the recurrence instead, which share nothing but the result)."""
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, timestamp_head=True)
    eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
    audio = [W.synth_audio(30 * 16000, 40 + u) for u in range(32)]
    first = eng.recognize(audio)
    for _ in range(6):
        r = eng.recognize(audio)
        np.testing.assert_array_equal(r.token_ids, first.token_ids)
        np.testing.assert_array_equal(r.cif_peak, first.cif_peak)
    eng.close()
