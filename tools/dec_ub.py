import sys, os
sys.path.insert(0, os.getcwd())
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
    print("%-50s %6d x %5d x %4d  %7.1f us  %6.0f TF" % (name, M, N, K, us, fpl / (us * 1e-6) / 1e12), flush=True)
M = 5344
run("dec FFN1 f16 row-major relu (auto)", M, 2048, 512, out_kind=1, relu=True)
run("dec FFN1 f16 row-major relu 256-row", M, 2048, 512, out_kind=1, relu=True, tile_rows=256)
run("dec FFN1 blocked relu (auto)", M, 2048, 512, out_kind=2, relu=True)
run("dec FFN1 blocked relu bigp forced", M, 2048, 512, out_kind=2, relu=True, tile_rows=1024)
run("dec FFN2 fp32 out row-major A (auto)", M, 512, 2048, out_kind=0)
run("dec FFN2 fp32 out blocked A (auto)", M, 512, 2048, out_kind=0, a_blocked=True)
run("dec FFN2 fp32 out blocked A 256-row", M, 512, 2048, out_kind=0, a_blocked=True, tile_rows=256)
eng.close()
