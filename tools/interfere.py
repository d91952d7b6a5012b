"""Does a kernel's result depend on what ELSE runs on the GPU?  Engine A (a timestamp model: persistent BiLSTM recurrence)
recognises in a loop on its own stream while engine B repeats single operators at the decoder's shapes and compares every
result with its first one.  Any difference = an operator that is not robust against foreign kernels sharing CUs / caches.
usage: python tools/interfere.py [seconds per operator]"""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
cfgA = W.paraformer_large_config(enc_layers=2, dec_layers=1, timestamp_head=True)
A = Engine(weights=W.pack_pfw(cfgA, W.synth_weights(cfgA, 1)), cmvn=W.synth_cmvn(), device=0)
cfgB = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=64)
B = Engine(weights=W.pack_pfw(cfgB, W.synth_weights(cfgB, 2)), cmvn=W.synth_cmvn(), device=0)
audio = [W.synth_audio(30 * 16000, u) for u in range(32)]
stop = False

rng = np.random.default_rng(0)
Md, D, F, T, L, Bt = 5344, 512, 2048, 500, 167, 32
x = rng.standard_normal((Md, D)).astype(np.float32)
h = np.abs(rng.standard_normal((Md, F))).astype(np.float32)
w1 = (rng.standard_normal((F, D)) / 22).astype(np.float32); w2 = (rng.standard_normal((D, F)) / 45).astype(np.float32)
wq = (rng.standard_normal((D, D)) / 22).astype(np.float32)
b1 = rng.standard_normal(F).astype(np.float32); b2 = rng.standard_normal(D).astype(np.float32)
q = rng.standard_normal((Bt, L, D)).astype(np.float32); k = rng.standard_normal((Bt, T, D)).astype(np.float32); v = rng.standard_normal((Bt, T, D)).astype(np.float32)
g = rng.standard_normal(D).astype(np.float32); be = rng.standard_normal(D).astype(np.float32)
fw = (0.1 * rng.standard_normal((D, 11))).astype(np.float32)
tn = np.full(Bt, L, np.int32)
xe = rng.standard_normal((16000, D)).astype(np.float32)
he = np.abs(rng.standard_normal((16000, F))).astype(np.float32)
speech = rng.standard_normal((8, 200, 560)).astype(np.float32)
lg = rng.standard_normal((2000, 8404)).astype(np.float32)
ops = {
  "int8 qlinear 5344x2048x512 f16-result": lambda: B.op_qlinear(x, w1, b1, relu=True, f16_result=True),
  "int8 qlinear 5344x512x2048 fp32": lambda: B.op_qlinear(h, w2, None),
  "int8 qlinear 16000x1536x512 f16-result": lambda: B.op_qlinear(xe, np.concatenate([wq, wq, wq]), None, f16_result=True),
  "ffn (up + down + residual) 5344": lambda: B.op_ffn(x, w1, b1, w2, b2, x),
  "whole encoder 8x200": lambda: B.op_encoder(speech),
  "logsoftmax + argmax 2000x8404": lambda: B.op_logsoftmax_argmax(lg)[0],
  "fsmn_enc 32x500": lambda: B.op_fsmn_enc(v, fw),
  "layernorm 16000x512": lambda: B.op_layernorm(xe, g, be),
  "gemm small 83x1536x512": lambda: B.op_gemm_ex(x[:83], np.concatenate([wq, wq, wq]), None, out_kind=1),
}
MODE = sys.argv[2] if len(sys.argv) > 2 else "recognize"
big = rng.standard_normal((16000, D)).astype(np.float32)
def disturb():
    n = 0
    while not stop:
        if MODE == "recognize": A.recognize(audio)
        else: A.op_layernorm(big, g, be)
        n += 1
    print("disturber (%s) ran %d times" % (MODE, n))
for phase in ("quiet", "disturbed"):
    th = None
    if phase == "disturbed":
        stop = False
        th = threading.Thread(target=disturb); th.start(); time.sleep(1.0)
    for name, f in ops.items():
        ref = f(); t0 = time.time(); n = bad = 0; worst = 0.0
        while time.time() - t0 < secs:
            y = f(); n += 1
            if not np.array_equal(y, ref, equal_nan=True):
                bad += 1; worst = max(worst, float(np.nanmax(np.abs(y - ref))))
        print("%-10s %-42s %4d runs, %3d differ from the first (max |d| %.3g)" % (phase, name, n, bad, worst), flush=True)
    if th:
        stop = True; th.join()
