"""Reads the per-wave phase sums an -DATT_TIMING build of k_attn.hip leaves behind (tools/attn_timing.sh).  One encoder step of the headline
shape; the LAST attention launch of the step (self-attention of the last encoder layer when the decoder is switched off below) is what
the table shows: mean over workgroups and waves, in shader-clock ticks and as a share of the kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W, _native as N
from aliparaformerasr_amd.engine import Engine

cfg = W.paraformer_large_config(enc_layers=4, dec_layers=2, vocab=128)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 5)), cmvn=W.synth_cmvn(), device=0)
audio = [W.synth_audio(480000, 3 + u) for u in range(32)]
for _ in range(3):
    res = eng.recognize(audio)
lib = N.load()
lib.pf_debug_attn_timing.restype = C.c_int
buf = np.zeros(1024 * 8 * 8, np.uint64)
rc = lib.pf_debug_attn_timing(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size)
assert rc == 0, rc
names = ["DMA wait", "barrier", "DMA issue", "QK issue + K reads", "mask + softmax (+ wait for S)", "PV (V reads + issue)", "prologue", "whole kernel"]
for title, lo, nw in (("self-attention, attn_kernel<8, .> (256 queries per workgroup)", 0, 8), ("cross-attention, attn_kernel<4, .>", 512, 4)):
    t = buf.reshape(1024, 8, 8)[lo:lo + 512, :nw].astype(np.float64)
    used = t[:, :, 7].sum(axis=1) > 0
    t = t[used]
    if not len(t):
        continue
    busy = t[:, :, 4] > 0                                     # waves whose queries lie beyond Lq only stage
    print(title, "- workgroups:", int(used.sum()), " working waves: %d of %d" % (int(busy.sum()), busy.size))
    tot = t[:, :, 7][busy].mean()
    for i, n in enumerate(names):
        v = t[:, :, i][busy]
        print("  %-32s %9.0f ticks  %5.1f %% of the kernel   (min %.0f, max %.0f over waves)" % (n, v.mean(), 100.0 * v.mean() / tot, v.min(), v.max()))
    loop = t[:, :, :6].sum(axis=2)[busy].mean()
    print("  tile loop %.0f ticks = %.1f %%; epilogue + rest %.1f %%" % (loop, 100 * loop / tot, 100 * (tot - loop - t[:, :, 6][busy].mean()) / tot))
eng.close()
