"""-DDM_TIMING build of k_decmid.hip: where the waves of the decoder's fused middle launch are at its phase boundaries."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W, _native as N
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config(enc_layers=4, dec_layers=4, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
audio = [W.synth_audio(480000, 3 + u) for u in range(32)]
for _ in range(3):
    res = eng.recognize(audio)
lib = N.load()
lib.pf_debug_decmid_timing.restype = C.c_int
buf = np.zeros(256 * 8 * 8, np.uint32)
assert lib.pf_debug_decmid_timing(buf.ctypes.data_as(C.POINTER(C.c_uint)), buf.size) == 0
t = buf.reshape(256, 8, 8).astype(np.float64)
t = t[t[:, 0, 5] > 0]
names = ["A: shares summed, hidden LN, norm2 -> LDS rows", "barrier", "B: FSMN + residual + norm3 -> LDS tile", "barrier", "C: q loop issued", "x and q stores acknowledged"]
prev = 0
print("dec_mid_kernel<11>: %d workgroups, L = %d; total %.0f ticks" % (len(t), res.L, t[:, :, 5].mean()))
for i, n in enumerate(names):
    d = t[:, :, i] - prev
    print("  %-50s at %7.0f (+%6.0f; min +%.0f max +%.0f)" % (n, t[:, :, i].mean(), d.mean(), d.min(), d.max()))
    prev = t[:, :, i]
eng.close()
