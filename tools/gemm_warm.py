"""Cold vs warm-L2 timing of one GEMM launch (PF_OP_REPEAT=4: launches 2..4 find A and W in L2 when they fit):
is the k-step of the persistent kernel bound by operand LATENCY (HBM vs L2) or by something else?"""
import sys, os
os.environ["PF_OP_REPEAT"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)
def run(name, M, N, K, **kw):
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    eng.op_gemm_ex(A, Wm, b, **kw)
    res = []
    for cls in ("gemm_op", "gemm_op_warm"):
        eng.profile_reset(); eng.profile_select(cls); eng.profile(True)
        for _ in range(4):
            eng.op_gemm_ex(A, Wm, b, **kw)
        eng.profile(False)
        ms, n, fpl = eng.profile_get(cls)
        res.append(ms / n * 1e3)
    print("%-34s %6d x %5d x %4d   cold %6.1f us   warm %6.1f us" % (name, M, N, K, res[0], res[1]), flush=True)
run("FFN-down f16 256-row tiles (32 tiles)", 2048, 512, 2048, out_kind=1, tile_rows=256)
run("FFN-down f16 128-row tiles (64 tiles)", 2048, 512, 2048, out_kind=1, tile_rows=128)
run("FFN-down f16 128-row tiles (32 tiles)", 1024, 512, 2048, out_kind=1, tile_rows=128)
run("K=8192 256-row tiles (32 tiles)", 2048, 512, 8192, out_kind=1, tile_rows=256)
run("K=8192 128-row tiles (32 tiles)", 1024, 512, 8192, out_kind=1, tile_rows=128)
for M in ():
    run("FFN-down shape fp32 out 256-row", M, 512, 2048, out_kind=0, tile_rows=256)
    run("FFN-down shape f16 out 256-row", M, 512, 2048, out_kind=1, tile_rows=256)
    run("FFN-up shape f16 256-row", M, 2048, 512, out_kind=1, tile_rows=256, relu=True)
    run("QKV shape f16 256-row", M, 1536, 512, out_kind=1, tile_rows=256)
eng.close()
