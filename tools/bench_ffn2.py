"""FFN-down ([16000 x 2048] x [2048 x 512], fp32 result) with and without the fp32 residual operand, and the
out-projection shape: how much of the launch is the residual round trip?  (pf_op_gemm_ex, HIP events around the launch.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)


def run(name, M, N, K, reps=8, **kw):
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    eng.op_gemm_ex(A, Wm, b, **kw)
    eng.profile_reset(); eng.profile_select("gemm_op"); eng.profile(True)
    for _ in range(reps):
        eng.op_gemm_ex(A, Wm, b, **kw)
    eng.profile(False)
    ms, n, fpl = eng.profile_get("gemm_op")
    us = ms / n * 1e3
    print("%-46s %6d x %5d x %4d  %7.1f us  %6.0f TF" % (name, M, N, K, us, fpl / (us * 1e-6) / 1e12), flush=True)


M = 16000
res = rng.standard_normal((M, 512)).astype(np.float32)
for ab in (False, True):
    run("FFN-down fp32 out, no residual, a_blocked=%d" % ab, M, 512, 2048, out_kind=0, a_blocked=ab)
    run("FFN-down fp32 out + residual,   a_blocked=%d" % ab, M, 512, 2048, out_kind=0, a_blocked=ab, resid=res)
run("FFN-down f16 out (no fp32 traffic)", M, 512, 2048, out_kind=1)
run("out-proj fp32 out + residual (pp3)", M, 512, 512, out_kind=0, resid=res)
run("out-proj fp32 out, no residual (pp3)", M, 512, 512, out_kind=0)
eng.close()
