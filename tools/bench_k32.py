"""k-step-32 six-stage kernel (k_gemm_k32.hip) against gemm_f16_pp3<2,2> with the SAME fp32 epilogue, shape by shape:
isolates what the K-loop pipeline changes.  L2-resident operands (QKV / out-projection shapes) vs the HBM-streamed
blocked FFN hidden."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)


def run(name, M, N, K, reps=6, **kw):
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
    print("%-44s %6d x %5d x %4d  %7.1f us  %6.0f TF" % (name, M, N, K, us, fpl / (us * 1e-6) / 1e12), flush=True)


for (M, N, K, blk) in ((16000, 1536, 512, False), (16000, 512, 512, False), (16000, 2048, 512, False), (16000, 512, 2048, True),
                       (16000, 512, 2048, False), (16000, 512, 8192, False), (4096, 512, 8192, False)):
    for tr, nm in ((256, "pp3<2,2>"), (2048, "k32")):
        run("fp32 out %s%s" % (nm, " blockedA" if blk else ""), M, N, K, out_kind=0, a_blocked=blk, tile_rows=tr)
eng.close()
