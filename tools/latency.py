"""Small-batch latency of the whole path (config 1: 1 x 5 s) — host wall time per recognise call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config()
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 42)), cmvn=W.synth_cmvn(), device=0)
_warm = [W.synth_audio(5 * 16000, 0)]
for _ in range(20):                    # the first calls of a process run ~2 ms slower (clocks, first-touch of the arenas)
    eng.recognize(_warm)
for (B, secs) in ((1, 5), (1, 30), (4, 5), (32, 5)):
    audio = [W.synth_audio(secs * 16000, u) for u in range(B)]
    eng.stage_audio(audio)
    for _ in range(3):
        eng.run_staged(); eng.sync()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.run_staged(); eng.sync()
    dt = (time.perf_counter() - t0) / n
    r = eng.fetch()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.recognize(audio)
    dt2 = (time.perf_counter() - t0) / n
    print("B=%d x %ds: staged %.2f ms, recognize(host audio) %.2f ms, L=%d, rtf %.2e" % (B, secs, dt * 1e3, dt2 * 1e3, r.L, dt2 / (B * secs)), flush=True)
eng.close()
