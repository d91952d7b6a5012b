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
        for _ in range(6):
            eng.op_gemm_ex(A, Wm, b, **kw)
        eng.profile(False)
        ms, n, fpl = eng.profile_get(cls)
        res.append(ms / n * 1e3)
    print("%-30s %5d x %5d x %4d   cold %6.1f us   warm %6.1f us" % (name, M, N, K, res[0], res[1]), flush=True)
run("QKV small", 83, 1536, 512, out_kind=1, tile_rows=32)
run("FFN-up small", 83, 2048, 512, out_kind=1, tile_rows=32, relu=True)
run("out small fp32", 83, 512, 512, out_kind=0, tile_rows=32)
run("FFN-down split", 83, 512, 2048, out_kind=0, tile_rows=32)
run("dec q small", 23, 512, 512, out_kind=1, tile_rows=32)
run("QKV pp3 128", 83, 1536, 512, out_kind=1, tile_rows=128)
eng.close()
