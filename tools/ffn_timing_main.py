"""-DFF_TIMING -DFF_TIMING_MAIN build of k_ffn.hip (tools/ffn_timing.sh with FF_EXTRA=-DFF_TIMING_MAIN FF_PY=tools/ffn_timing_main.py): where the
waves are at the end of phase U and phase D of each of the 8 chunks of the FFN main loop (ticks since the wave's start; in the order the
workgroup WALKS the chunks, i.e. after its rotation)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W, _native as N
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=6, dec_layers=2, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
audio = [W.synth_audio(480000, 3 + u) for u in range(32)]
for _ in range(3):
    eng.recognize(audio)
lib = N.load()
lib.pf_debug_ffn_timing.restype = C.c_int
buf = np.zeros(256 * 8 * 16, np.uint32)
assert lib.pf_debug_ffn_timing(buf.ctypes.data_as(C.POINTER(C.c_uint)), buf.size) == 0
t = buf.reshape(256, 8, 16)[:250].astype(np.float64)
prev = None
for c in range(8):
    u, d = t[:, :, 2 * c], t[:, :, 2 * c + 1]
    start = prev if prev is not None else u - 0
    print("chunk %d (walk order): U ends at %8.0f (+%6.0f), D ends at %8.0f (+%6.0f)   [MFMA minimum per phase: 4096 per wave, 8192 per SIMD]"
          % (c, u.mean(), (u - (prev if prev is not None else u)).mean(), d.mean(), (d - u).mean()))
    prev = d
eng.close()
