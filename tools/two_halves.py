"""Experiment: ONE batch of 32 x 30 s as TWO resident half-batches on two engines of the same GPU, each engine's persistent
kernels sized for half the CUs (PF_CU_CAP=128), against one engine x 32.  Audio resident in HBM in both cases (bench.py's
--group form stages from host memory, which confounds the comparison).  Usage: [PF_CU_CAP=128] python tools/two_halves.py N_ENGINES B_EACH"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine

n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 2
b_each = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(os.environ.get("STEPS", 10))
samples = int(os.environ.get("SAMPLES", 480000))      # per utterance (80000 = the 5 s latency case)
cfg = W.paraformer_large_config()
blob = W.pack_pfw(cfg, W.synth_weights(cfg, 42))
cm = W.synth_cmvn()
engs = [Engine(weights=blob, cmvn=cm, device=0) for _ in range(n_eng)]
for e, eng in enumerate(engs):
    eng.stage_audio([W.synth_audio(samples, e * b_each + u) for u in range(b_each)])
    for _ in range(3):
        eng.run_staged()
    eng.sync()
bar = threading.Barrier(n_eng + 1)


def work(eng):
    bar.wait()
    for _ in range(steps):
        eng.run_staged()
    eng.sync()
    bar.wait()


th = [threading.Thread(target=work, args=(e,)) for e in engs]
for t in th:
    t.start()
bar.wait()
t0 = time.perf_counter()
bar.wait()
dt = time.perf_counter() - t0
for t in th:
    t.join()
per32 = dt / steps * 1e3 * 32.0 / (n_eng * b_each)
print("engines %d x %d utterances of %.0f s, PF_CU_CAP=%s: %.2f ms per step of %d utterances = %.2f ms per 32 = %.0f utterances/s" %
      (n_eng, b_each, samples / 16000.0, os.environ.get("PF_CU_CAP", "-"), dt / steps * 1e3, n_eng * b_each, per32,
       n_eng * b_each * steps / dt), flush=True)
for e in engs:
    e.close()
