import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
M, N, K = [int(x) for x in sys.argv[1:4]]
rng = np.random.default_rng(0)
A = rng.standard_normal((M, K)).astype(np.float32); Wm = rng.standard_normal((N, K)).astype(np.float32)
for _ in range(3):
    eng.op_gemm(A, Wm, None, f16_out=True)
eng.close()
