"""Times the decoder's FFN block in the split form of the fused kernel (k_ffn.hip) for every split count at a given row
count (PF_OP_REPEAT=8 for warm calls; under `rocprofv3 --kernel-trace --stats` the per-kernel averages).
usage: python tools/bench_dec_ffn.py [M] [splits ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aliparaformerasr_amd import weights as W  # noqa: E402
from aliparaformerasr_amd.engine import Engine  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 5344
    splits = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 8]
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=64)
    eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, seed=3)), cmvn=W.synth_cmvn(), device=0)
    rng = np.random.default_rng(0)
    D, F = 512, 2048
    x = rng.standard_normal((M, D)).astype(np.float32)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(F)).astype(np.float32)
    gf = (1 + 0.2 * rng.standard_normal(F)).astype(np.float32)
    bf = (0.1 * rng.standard_normal(F)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    g = np.ones(D, np.float32)
    be = np.zeros(D, np.float32)
    for s in splits:
        eng.profile_reset(); eng.profile_select("gemm_op_warm"); eng.profile(True)
        eng.op_dec_ffn_fused(x, w1, b1, (gf, bf), w2, ln=(g, be), splits=s)
        eng.profile(False)
        ms, n, _ = eng.profile_get("gemm_op_warm")
        print("M %d splits %d: %.2f us per call (split kernel + finishing pass, %d warm calls)" % (M, s, ms / max(n, 1) * 1e3, n))
    eng.close()


if __name__ == "__main__":
    main()
