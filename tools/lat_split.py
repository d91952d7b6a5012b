"""Where the host-audio call spends its time at 1 x 5 s: stage (H2D), run, sync, fetch — and recognize() as a whole."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config()
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 42)), cmvn=W.synth_cmvn(), device=0)
for secs in (5, 30, 5):
    audio = [W.synth_audio(secs * 16000, 0)]
    for _ in range(3):
        eng.recognize(audio)
    acc = [0.0] * 5
    n = 20
    for _ in range(n):
        t0 = time.perf_counter(); eng.stage_audio(audio)
        t1 = time.perf_counter(); eng.run_staged()
        t2 = time.perf_counter(); eng.sync()
        t3 = time.perf_counter(); r = eng.fetch()
        t4 = time.perf_counter(); eng.recognize(audio)
        t5 = time.perf_counter()
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[i] += d
    print("1 x %2d s: stage %.3f  run(host side) %.3f  sync %.3f  fetch %.3f | recognize %.3f ms" % ((secs,) + tuple(a / n * 1e3 for a in acc)), flush=True)
eng.close()
