"""Latency config under rocprofv3: 1 x 5 s, 20 staged runs (kernel-trace gives the GPU-busy share of the wall time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
print("staged 1 x 5 s: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
eng.close()
