"""Times the fused encoder FFN block (k_ffn.hip, one launch) through pf_op_ffn_fused against the two launches it replaces
(pf_op_gemm_ex FFN-up on the persistent 256 x 256 kernel + pf_op_gemm_rc FFN-down with the LayerNorm epilogue), HIP events
around the launches only, cold (first launch after the upload) and warm (PF_OP_REPEAT back-to-back launches).
    M=16000 PF_FFN_PF=12 PF_FFN_ROT=7 python tools/bench_ffn.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PF_OP_REPEAT", "8")
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)
M = int(os.environ.get("M", 16000))
D, F = 512, 2048
x = rng.standard_normal((M, D)).astype(np.float32)
w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
b1 = (0.1 * rng.standard_normal(F)).astype(np.float32)
w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
b2 = (0.1 * rng.standard_normal(D)).astype(np.float32)
resid = rng.standard_normal((M, D)).astype(np.float32)
ln = (np.ones(D, np.float32), np.zeros(D, np.float32))


def timed(name, fn, flops):
    for cls in ("gemm_op", "gemm_op_warm"):
        eng.profile_reset(); eng.profile_select(cls); eng.profile(True)
        fn()
        eng.profile(False)
        ms, n, _ = eng.profile_get(cls)
        if n:
            us = ms / n * 1e3
            print("%-52s %-5s %7.1f us  %6.0f TF" % (name, "cold" if cls == "gemm_op" else "warm", us, flops / (us * 1e-6) / 1e12), flush=True)


timed("fused FFN + bias + residual + LayerNorm (k_ffn.hip)", lambda: eng.op_ffn_fused(x, w1, b1, w2, b2, resid, ln=ln), 4.0 * M * D * F)
h = np.maximum(x @ w1.T + b1, 0).astype(np.float32)
timed("FFN-up, persistent 256x256 blocked result", lambda: eng.op_gemm_ex(x, w1, b1, out_kind=2, relu=True, tile_rows=1024), 2.0 * M * D * F)
timed("FFN-down row-complete + residual + LayerNorm", lambda: eng.op_gemm_rc(h, w2, bias=b2, resid=resid, ln=ln, a_blocked=True, want_n32=False), 2.0 * M * D * F)
eng.close()
