"""How much of the 1 x 5 s wall time is GPU kernel time?  Run under rocprofv3 --kernel-trace; prints wall per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config()
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 42)), cmvn=W.synth_cmvn(), device=0)
audio = [W.synth_audio(5 * 16000, 0)]
eng.stage_audio(audio)
for _ in range(3):
    eng.run_staged(); eng.sync()
t0 = time.perf_counter()
for _ in range(20):
    eng.run_staged(); eng.sync()
print("WALL_MS_PER_CALL %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
eng.close()
