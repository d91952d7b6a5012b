"""A second PROCESS that keeps the GPU's CUs shared: recognises with a timestamp model (encoder + persistent BiLSTM beside
the decoder) in a loop for N seconds.  Run the test suite beside it to look for kernels whose results depend on timing:
    python tools/disturber.py 600 & python -m pytest tests -m gpu -x -q"""
import sys, time
sys.path.insert(0, ".")
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, timestamp_head=True)
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 1)), cmvn=W.synth_cmvn(), device=0)
audio = [W.synth_audio(30 * 16000, u) for u in range(32)]
t0, n = time.time(), 0
while time.time() - t0 < secs:
    eng.recognize(audio); n += 1
print("disturber: %d recognitions in %.0f s" % (n, time.time() - t0))
