"""Times one GEMM shape through pf_op_gemm (kernel-only, HIP events).  Used by tools/gemm_abl.sh,
which rebuilds k_gemm.hip with -DPF_ABL=<bits> (1 no steady-state DMA, 2 no fragment reads, 4 no MFMA,
8 no f16 output stores, 32 no transpose passes) — ablated results are wrong by construction.
usage: gemm_abl.py TAG N K f16out(0/1)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)
tag, N, K, f16 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1"
M = 16000
A = rng.standard_normal((M, K)).astype(np.float32)
Wm = rng.standard_normal((N, K)).astype(np.float32)
eng.op_gemm(A, Wm, None, f16_out=f16)
eng.profile_reset(); eng.profile_select("gemm_op"); eng.profile(True)
for _ in range(5):
    eng.op_gemm(A, Wm, None, f16_out=f16)
eng.profile(False)
ms, n, fpl = eng.profile_get("gemm_op")
print("abl=%s N=%d K=%d f16=%s: %.1f us  %.0f TF-equivalent" % (tag, N, K, f16, ms / n * 1e3, fpl / (ms / n * 1e-3) / 1e12), flush=True)
eng.close()
