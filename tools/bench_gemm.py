"""Times the f16-result GEMM variants at the benchmark's shapes through pf_op_gemm_ex (HIP events around the launch)."""
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
    print("%-40s %6d x %5d x %4d  %7.1f us  %6.0f TF" % (name, M, N, K, us, fpl / (us * 1e-6) / 1e12), flush=True)


for M in (16000, 64000):
    for tr, nm in ((256, "pp3 256x128"), (512, "big")):
        run("QKV f16 " + nm, M, 1536, 512, out_kind=1, tile_rows=tr, scale_cols=512, scale=0.088)
        run("FFN-up blocked relu " + nm, M, 2048, 512, out_kind=2, relu=True, tile_rows=tr)
    run("FFN-up blocked relu persistent 256x256", M, 2048, 512, out_kind=2, relu=True, tile_rows=1024)
run("dec K/V pp3", 16000, 16384, 512, out_kind=1, tile_rows=256, reps=3)
run("dec K/V big", 16000, 16384, 512, out_kind=1, tile_rows=512, reps=3)
eng.close()
