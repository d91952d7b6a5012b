"""FFN-down + bias + residual + LayerNorm at the benchmark's shape, three forms (HIP events around the launch only):
the split-K pair kernel (k_gemm_sk.hip), the 64-row row-complete kernel (k_gemm_rc.hip), and the persistent 256 x 128
kernel (k_gemm.hip; its LayerNorm launch is NOT in that figure: add ~9 us)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)
ln = (np.ones(512, np.float32), np.zeros(512, np.float32))
bias = rng.standard_normal(512).astype(np.float32)


def timed(fn, reps=6):
    fn()
    eng.profile_reset(); eng.profile_select("gemm_op"); eng.profile(True)
    for _ in range(reps):
        fn()
    eng.profile(False)
    ms, n, fpl = eng.profile_get("gemm_op")
    us = ms / n * 1e3
    return us, fpl / (us * 1e-6) / 1e12


for M, K in ((16000, 2048), (16000, 512), (64000, 2048), (10944, 2048), (5344, 2048), (5344, 512)):
    A = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    Wm = (rng.standard_normal((512, K)) / np.sqrt(K)).astype(np.float32)
    resid = rng.standard_normal((M, 512)).astype(np.float32)
    for name, fn in (
        ("split-K pairs 128x512 (+LN)", lambda: eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, ln=ln, a_blocked=True, split_k=True, want_n32=False)),
        ("split-K pairs, x only", lambda: eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, a_blocked=True, split_k=True)),
        ("row-complete 64x512 (+LN)", lambda: eng.op_gemm_rc(A, Wm, bias=bias, resid=resid, ln=ln, a_blocked=True, want_n32=False)),
        ("persistent 256x128 (no LN)", lambda: eng.op_gemm_ex(A, Wm, bias, out_kind=0, a_blocked=True, resid=resid)),
    ):
        us, tf = timed(fn)
        print("M=%6d K=%4d  %-30s %7.1f us  %6.0f TF" % (M, K, name, us, tf), flush=True)
eng.close()
