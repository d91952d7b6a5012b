"""Kernel micro-benchmarks through the C ABI (device time from HIP events on the engine stream).
Usage: python tools/bench_ops.py [gemm|attn|all]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
rng = np.random.default_rng(0)
if what in ("gemm", "all"):
    for (M, N, K) in ((16000, 1536, 512), (16000, 512, 512), (16000, 2048, 512), (16000, 512, 2048),
                      (7232, 512, 512), (7232, 2048, 512), (4800, 8404, 512), (16000, 16384, 512)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        Wm = rng.standard_normal((N, K)).astype(np.float32)
        eng.profile_reset(); eng.profile_select("gemm_op"); eng.profile(True)
        for _ in range(4):
            eng.op_gemm(A, Wm, None, f16_out=True)
        eng.profile(False)
        ms, n, fpl = eng.profile_get("gemm_op")
        print("gemm M=%d N=%d K=%d: %.1f us  %.0f TF" % (M, N, K, ms / n * 1e3, fpl / (ms / n * 1e-3) / 1e12), flush=True)
if what in ("attn", "all"):
    for (B, Lq, Lk) in ((32, 500, 500), (32, 226, 500), (32, 150, 500)):
        q = rng.standard_normal((B, Lq, 512)).astype(np.float32) * 0.3
        k = rng.standard_normal((B, Lk, 512)).astype(np.float32) * 0.3
        v = rng.standard_normal((B, Lk, 512)).astype(np.float32)
        eng.profile_reset(); eng.profile_select("attn_op"); eng.profile(True)
        for _ in range(6):
            eng.op_attention(q, k, v, 4)
        eng.profile(False)
        ms, n, fpl = eng.profile_get("attn_op")
        print("attn B=%d Lq=%d Lk=%d: %.1f us  %.0f TF" % (B, Lq, Lk, ms / n * 1e3, fpl / (ms / n * 1e-3) / 1e12), flush=True)
eng.close()
