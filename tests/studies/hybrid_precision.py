"""Study (CPU, oracle only): what would carrying the LAST N encoder layers in fp32 buy?  

The f16-operand engine moves sum(alpha) of a 30 s utterance by up to ~0.07, which flips
token_num = floor(sum alpha) for utterances whose fp32 sum lies just below an integer.  Here the oracle runs the
benchmark's full-depth graph on a few benchmark utterances with 16-bit GEMM operands in the first 50 - N encoder
layers and fp32 in the last N (+ the predictor when N > 0), and prints sum(alpha) against the all-fp32 run.

    python tests/studies/hybrid_precision.py [n_utts] > profiles/round4_hybrid_precision.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import model as om


class Hybrid(om.Oracle):
    """16-bit operands (as the engine) in encoder layers < first_fp32, fp32 from there on (predictor included)."""

    def __init__(self, cfg, w, first_fp32):
        super().__init__(cfg, w, quant="fp16")
        self.first_fp32 = first_fp32
        self.q16 = self.q
        self.q32 = om.quantizer("fp32")

    def enc_layer(self, x, p, first):
        i = int(p.rsplit(".", 1)[1])
        self.q = self.q32 if i >= self.first_fp32 else self.q16
        try:
            return super().enc_layer(x, p, first)
        finally:
            self.q = self.q16

    def cif_alphas(self, H):
        self.q = self.q32 if self.first_fp32 < self.cfg.enc_layers else self.q16
        try:
            return super().cif_alphas(H)
        finally:
            self.q = self.q16


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = W.paraformer_large_config()
    w = W.synth_weights(cfg, 42)
    cmvn = W.synth_cmvn()
    # the benchmark's near-tie utterances come first (tests/golden/bench_paraformer.npz)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "bench_paraformer.npz"))
    frac = g["alpha_sum"] - np.floor(g["alpha_sum"])
    order = np.argsort(np.minimum(frac, 1 - frac))
    utts = [int(u) for u in order[:n]]
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(W.synth_audio(480000, u), conf, *cmvn) for u in utts]
    speech = fe.pad_sequence(feats).reshape(len(utts), -1, 560)
    mc = om.ModelConfig(**cfg)
    print("utterances (benchmark seeds, nearest to a floor boundary first):", utts)
    print("golden fp32 sum(alpha):", [round(float(g["alpha_sum"][u]), 4) for u in utts])
    rows = {}
    for label, first in (("fp32 everywhere", 0), ("fp16 operands everywhere (the engine)", 50), ("last 5 layers fp32", 45),
                         ("last 10 layers fp32", 40), ("last 25 layers fp32", 25)):
        t0 = time.time()
        o = Hybrid(mc, w, first)
        H = o.encoder(speech)
        a = o.cif_alphas(H).numpy()
        s = a.astype(np.float64).sum(axis=1)
        rows[label] = s
        print("%-40s sum(alpha) = %s   floor = %s   (%.0f s)" % (label, np.round(s, 4).tolist(), np.floor(s).astype(int).tolist(), time.time() - t0),
              flush=True)
    ref = rows["fp32 everywhere"]
    for k, s in rows.items():
        d = s - ref
        print("%-40s shift vs fp32: mean %+.4f max %+.4f   token_num differs for %d of %d" %
              (k, d.mean(), np.abs(d).max() * np.sign(d[np.abs(d).argmax()]), int((np.floor(s) != np.floor(ref)).sum()), len(s)))


if __name__ == "__main__":
    main()
