import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)
M, K = 16000, 512
A = rng.standard_normal((M, K)).astype(np.float32)
for N in [int(x) for x in sys.argv[1:]]:
    Wm = rng.standard_normal((N, K)).astype(np.float32)
    for f16 in (True,):
        eng.op_gemm(A, Wm, None, f16_out=f16)
        eng.profile_reset(); eng.profile_select("gemm_op"); eng.profile(True)
        for _ in range(3):
            eng.op_gemm(A, Wm, None, f16_out=f16)
        eng.profile(False)
        ms, n, fpl = eng.profile_get("gemm_op")
        print("N=%d f16_out=%s: %.1f us  %.0f TF" % (N, f16, ms / n * 1e3, fpl / (ms / n * 1e-3) / 1e12), flush=True)
eng.close()
