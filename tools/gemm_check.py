import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
rng = np.random.default_rng(0)
h16 = lambda x: x.astype(np.float16).astype(np.float32)
for (M, N, K) in ((150, 1536, 576), (150, 2048, 512), (1000, 1536, 512), (16000, 2048, 512), (333, 512, 2048), (4800, 1024, 512), (257, 128, 64), (600, 16384, 512)):
    A = rng.standard_normal((M, K)).astype(np.float32); Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = h16(A).astype(np.float64) @ h16(Wm).astype(np.float64).T + b
    bad = 0
    for rep in range(3):
        got = eng.op_gemm(A, Wm, b, f16_out=True)
        err = np.abs(got - ref)
        tol = 2e-3 * np.maximum(1, np.abs(ref))
        nbad = int((err > tol).sum())
        if nbad:
            idx = np.argwhere(err > tol)
            print("  M=%d N=%d K=%d rep %d: %d bad, rows %s cols %s" % (M, N, K, rep, nbad, sorted(set(idx[:, 0] // 8 * 8))[:12], sorted(set(idx[:, 1] // 64 * 64))[:12]))
        bad += nbad
    print("M=%d N=%d K=%d: %s" % (M, N, K, "OK" if bad == 0 else "FAIL %d" % bad), flush=True)
eng.close()
