"""Streaming path timing (the reference publishes rtf 0.1064 for its online recogniser on an i7-10750H,
README.EN.md:183-185): paraformer-large geometry, N concurrent streams fed 0.6 s pieces (= one 60-frame chunk per call),
GetResults after every piece.  Prints rtf = wall / audio per stream count."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.online_recognizer import OnlineRecognizer
from oracle import frontend as fe

d = tempfile.mkdtemp()
cfg = W.paraformer_large_config()
w = W.synth_weights(cfg, 42)
w["predictor.out.bias"] = np.asarray([0.8], np.float32)
W.save_pfw(os.path.join(d, "model.pfw"), cfg, w)
sh, sc = W.synth_cmvn()
open(os.path.join(d, "am.mvn"), "w").write(fe.format_mvn_text(sh, sc))
open(os.path.join(d, "asr.yaml"), "w").write("model: paraformer\nfrontend_conf:\n  dither: 0\n")
open(os.path.join(d, "tokens.txt"), "w").write("\n".join("t%d" % i for i in range(cfg["vocab"])) + "\n")
rec = OnlineRecognizer(os.path.join(d, "model.pfw"), "", os.path.join(d, "asr.yaml"), os.path.join(d, "am.mvn"), os.path.join(d, "tokens.txt"))
SEC = 12
for n in (1, 8, 32):
    audio = [W.synth_audio(16000 * SEC, u) for u in range(n)]
    streams = [rec.CreateOnlineStream() for _ in range(n)]
    for warm in range(2):
        for s, a in zip(streams, audio):
            s.AddSamples(a[:9600])
        rec.GetResults(streams)
    t0 = time.perf_counter()
    calls = 0
    for off in range(9600, 16000 * SEC, 9600):
        for s, a in zip(streams, audio):
            s.AddSamples(a[off: off + 9600])
        rec.GetResults(streams)
        calls += 1
    dt = time.perf_counter() - t0
    aud = n * (16000 * SEC - 9600) / 16000.0
    print("streams %2d: %.1f ms per GetResults (one 0.6 s chunk per stream), rtf %.5f (RTFx %.0f), tokens/stream %d"
          % (n, dt / calls * 1e3, dt / aud, aud / dt, len(streams[0].Tokens)), flush=True)
rec.Dispose()
