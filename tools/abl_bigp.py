"""Timing of the persistent 256 x 256 kernel (FFN-up shape) at ABL_M rows (64000 = 8 tiles per workgroup); the
ablation bits are compile-time (tools/abl_big.sh rebuilds with -DPF_BIG_ABL=n)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)
M, N, K = int(os.environ.get('ABL_M', 64000)), 2048, int(os.environ.get('ABL_K', 512))
A = rng.standard_normal((M, K)).astype(np.float32)
Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
b = rng.standard_normal(N).astype(np.float32)
eng.op_gemm_ex(A, Wm, b, relu=True, out_kind=2, tile_rows=1024)
eng.profile_reset(); eng.profile_select("gemm_op"); eng.profile(True)
for _ in range(4):
    eng.op_gemm_ex(A, Wm, b, relu=True, out_kind=2, tile_rows=1024)
eng.profile(False)
ms, n, fpl = eng.profile_get("gemm_op")
print("M=%d K=%d persistent 256x256: %7.1f us  %.0f TF" % (M, K, ms / n * 1e3, fpl / (ms / n * 1e-3) / 1e12), flush=True)
eng.close()
