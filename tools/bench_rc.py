"""Times the row-complete GEMM (k_gemm_rc.hip) at the benchmark's shapes through pf_op_gemm_rc (HIP events around
the launch only), one line per epilogue configuration — the ablation used to tune it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)
M = int(os.environ.get("M", 16000))
T = 500


def run(name, K, reps=6, **kw):
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((512, K)) / np.sqrt(K)).astype(np.float32)
    eng.profile_reset(); eng.profile_select("gemm_op"); eng.profile(True)
    for _ in range(reps):
        eng.op_gemm_rc(A, Wm, **kw)
    eng.profile(False)
    ms, n, fpl = eng.profile_get("gemm_op")
    us = ms / n * 1e3
    print("%-34s K=%4d  %7.1f us  %6.0f TF" % (name, K, us, fpl / (us * 1e-6) / 1e12), flush=True)


bias = rng.standard_normal(512).astype(np.float32)
resid = rng.standard_normal((M, 512)).astype(np.float32)
v = rng.standard_normal((M, 512)).astype(np.float32)
fw = (0.1 * rng.standard_normal((512, 11))).astype(np.float32)
ln = (np.ones(512, np.float32), np.zeros(512, np.float32))
for K in (512, 2048):
    run("bare (x only)", K)
    run("+bias+resid", K, bias=bias, resid=resid)
    run("+bias+resid+ln", K, bias=bias, resid=resid, ln=ln, want_n32=False)
    run("+bias+resid+fsmn", K, bias=bias, resid=resid, fsmn_v=v, fsmn_w=fw, T=T)
    run("+bias+resid+fsmn+ln (out-proj)", K, bias=bias, resid=resid, fsmn_v=v, fsmn_w=fw, T=T, ln=ln, want_n32=False)
    run("blocked A +bias+resid+ln (ffn2)", K, bias=bias, resid=resid, ln=ln, a_blocked=True, want_n32=False)
eng.close()
