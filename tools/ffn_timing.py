"""Reads the per-wave phase stamps an -DFF_TIMING build of k_ffn.hip leaves behind (tools/ffn_timing.sh): where the waves of the
encoder's fused launch (ffn_fused_kernel<8, 0, 2, 1, 1, 0>: out-projection + FSMN + norm2 + FFN + next norm1 + next Q | K | V) are at each
phase boundary, in shader-clock ticks since the wave's own start; mean / min / max over the 250 x 8 waves of the LAST such launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W, _native as N
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=6, dec_layers=2, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
audio = [W.synth_audio(480000, 3 + u) for u in range(32)]
for _ in range(3):
    res = eng.recognize(audio)
lib = N.load()
lib.pf_debug_ffn_timing.restype = C.c_int
buf = np.zeros(256 * 8 * 16, np.uint32)
rc = lib.pf_debug_ffn_timing(buf.ctypes.data_as(C.POINTER(C.c_uint)), buf.size)
assert rc == 0, rc
t = buf.reshape(256, 8, 16)[:250].astype(np.float64)
names = ["context tile + first Wo fragments landed (vmcnt(0) + barrier)", "P1: ctx Wo^T MFMA loop issued", "P2: residual + V window loaded, bias + FSMN done",
         "x_mid back in accumulators, norm2 -> LDS operand tile, b1 + first W fragments landed", "FFN main loop (8 chunks) issued",
         "epilogue: dump, bias + residual, x stores issued", "next norm1 -> LDS tile, bq DMA + first Wq fragments issued",
         "vmcnt(0) + barrier in front of the Q | K | V tail", "pass Q loop issued", "Q stores issued", "pass K loop issued", "K stores issued",
         "pass V loop issued", "V stores issued", "all stores acknowledged (vmcnt(0))"]
prev = np.zeros_like(t[:, :, 0])
tot = t[:, :, 14].mean()
print("ffn_fused_kernel<8,0,2,1,1,0>, 250 workgroups x 8 waves; total %.0f ticks per wave (100 MHz-class constant clock if s_memtime counts REFCLK; see ratio only)" % tot)
for i, n in enumerate(names):
    d = t[:, :, i] - prev
    print("  %2d %-92s at %8.0f  (+%7.0f = %5.1f %%; min +%.0f, max +%.0f)" % (i, n, t[:, :, i].mean(), d.mean(), 100 * d.mean() / tot, d.min(), d.max()))
    prev = t[:, :, i]
eng.close()
