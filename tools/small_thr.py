"""Row threshold of the short-input GEMM path: staged time of a few batch shapes (run with PF_SMALL_M=...)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config()
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 42)), cmvn=W.synth_cmvn(), device=0)
out = []
for (B, secs) in ((1, 5), (8, 5), (2, 30), (4, 30), (1, 5), (8, 5), (2, 30), (4, 30)):
    audio = [W.synth_audio(secs * 16000, u) for u in range(B)]
    eng.stage_audio(audio)
    for _ in range(3):
        eng.run_staged(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.run_staged(); eng.sync()
    out.append("%dx%ds %.2f" % (B, secs, (time.perf_counter() - t0) * 100))
print("PF_SMALL_M=%s:" % os.environ.get("PF_SMALL_M", "default"), "  ".join(out[4:]), flush=True)
eng.close()
