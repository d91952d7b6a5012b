#!/usr/bin/env python
"""Generates tests/golden/bench_*.npz: what the fp32 CPU oracle (oracle/model.py, quant="fp32") returns for the
three benchmark workloads of bench.py at FULL model depth — so that `bench.py` and the `-m gpu` tests can check the
ids of the timed computation itself against an oracle run without repeating it (tests/test_gpu_full_depth.py also
repeats it live and cross-checks this file).

    python tests/golden/make_bench_golden.py [paraformer] [sensevoice] [seaco] [paraformer_int8] [seaco_int8]

`<name>_int8` (round 5): the same workload through `Oracle(quant="int8_ref")` — every Linear as DynamicQuantizeLinear +
MatMulInteger + rescale in exact integer / float32 arithmetic (oracle/int8.py), no 16-bit rounding point anywhere: what
onnxruntime computes on the reference's DEFAULT model.int8.onnx (Examples/Program.cs:98-101).  `bench.py --accuracy int8`
checks its ids against these files (`ids_vs_int8_oracle`).

Workloads = bench.py's: seeded synthetic weights (weights.synth_weights(cfg, 42)), synthetic audio
(weights.synth_audio(samples, utt)), CMVN weights.synth_cmvn(); the audio goes through the oracle front-end
(oracle/frontend.py, dither 0).  Stored per workload (small: ids + margins, never the [B, L, V] log-probs):
  ids        [B, L] int32   last-index arg-max of the oracle's log-probs (OfflineRecognizer.cs:139-152)
  margin     [B, L] float32 top-1 minus top-2 log-prob of the oracle row
  top1       [B, L] float32 the top-1 log-prob
  token_num  [B]    int32
  fire_count [B]    int32   (paraformer / seaco)
  alpha_sum  [B]    float32 (paraformer / seaco) sum of the CIF weights; token_num = floor(alpha_sum): an utterance whose
                            sum lies within a few 1e-2 of an integer is a near-tie of that floor (16-bit GEMM operands
                            move the sum by up to ~0.07 at T = 500)
  seaco only: us_fire [B, F] int32 fire frames of us_cif_peak (-1 padded), us_fire_clear [B, F] float32 = by how
  much the oracle's integrator clears the threshold at the fire frame and misses it on the frame before (the smaller
  of the two), hw [21, 10] the PadList'ed hotword ids.
The oracle is deterministic up to the summation order of the host BLAS (thread count): positions whose margin is
below ~1e-4 may differ between hosts; consumers mask by margin.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aliparaformerasr_amd import weights as W      # noqa: E402
from oracle import frontend as fe, glue, model as om   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def workload(name):
    """(cfg, weights, cmvn, audio, hotwords) exactly as bench.py builds them for --model <name>."""
    cmvn = W.synth_cmvn()
    hw = None
    if name == "paraformer":
        cfg = W.paraformer_large_config()
        audio = [W.synth_audio(480000, u) for u in range(32)]
    elif name == "sensevoice":
        cfg = W.sensevoice_small_config(use_itn=True)
        audio = [W.synth_audio(160000, u) for u in range(64)]
    elif name == "seaco":
        cfg = W.seaco_paraformer_config()
        audio = [W.synth_audio(480000, u) for u in range(32)]
        hrng = np.random.default_rng(99)
        hws = [list(map(int, hrng.integers(3, 8000, size=int(hrng.integers(2, 5))))) for _ in range(20)] + [[1]]
        hw = np.asarray(glue.pad_list(hws), np.int32)
    else:
        raise KeyError(name)
    return cfg, W.synth_weights(cfg, 42), cmvn, audio, hw


def speech_of(audio, cmvn, prep=None):
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    if prep is not None:
        feats = [prep(f) for f in feats]
    T = max(f.shape[0] for f in feats)
    return fe.pad_sequence(feats).reshape(len(audio), T, 560)


def top2(logits):
    part = np.partition(logits, logits.shape[-1] - 2, axis=-1)
    return part[..., -1], part[..., -1] - part[..., -2]


def us_fires(peak, alphas, thr):
    """fire frames of us_cif_peak + how clearly each one is decided: min(peak[f] - thr, thr - peak[f-1] - alpha... )
    = the smaller distance of the integrator to the threshold on the firing frame and on the frame before it."""
    B = peak.shape[0]
    fr = [np.nonzero(peak[b] > thr)[0] for b in range(B)]
    F = max(len(f) for f in fr)
    out = np.full((B, F), -1, np.int32)
    clear = np.zeros((B, F), np.float32)
    for b in range(B):
        out[b, :len(fr[b])] = fr[b]
        for i, f in enumerate(fr[b]):
            above = peak[b, f] - thr
            below = thr - peak[b, f - 1] if f > 0 else 1.0
            if f > 0 and peak[b, f - 1] > thr:          # previous frame fired too: its residue restarts the integrator
                below = thr - (peak[b, f - 1] - thr)
            clear[b, i] = min(above, below)
    return out, clear


def run(name):
    full_name = name
    quant = "fp32"
    if name.endswith("_int8"):
        name, quant = name[:-5], "int8_ref"
    elif name.endswith("_int8q"):               # the int8 graph WITH the engine's 16-bit rounding points (attention, FSMN, stored
        name, quant = name[:-6], "int8"         # f16 activations): what the kernels are built to compute, as quant="fp16" is for mode 0
    cfg, w, cmvn, audio, hw = workload(name)
    mc = om.ModelConfig(**cfg)
    orc = om.Oracle(mc, w, quant=quant)
    t0 = time.time()
    out = {}
    with torch.inference_mode():
        if name == "sensevoice":
            sp = speech_of(audio, cmvn, prep=lambda f: glue.sensevoice_prepend(f, w["embed.weight"], use_itn=True))
            r = orc.sensevoice(sp)
        elif name == "seaco":
            r = orc.seaco(speech_of(audio, cmvn), hw)
            thr = np.float32(np.float32(mc.cif_threshold) - np.float32(1e-4))
            out["us_fire"], out["us_fire_clear"] = us_fires(r["us_cif_peak"], r["us_alphas"], thr)
            out["hw"] = hw
            dha = r["dha_logits"]
            nb = cfg["seaco_nobias"]
            other = np.where(np.arange(dha.shape[-1])[None, None, :] == nb, -np.inf, dha).max(-1)
            out["nobias_clear"] = np.abs(dha[..., nb] - other).astype(np.float32)
        else:
            r = orc.paraformer(speech_of(audio, cmvn))
    lg = r["logits"]
    top1, margin = top2(lg)
    out.update(ids=om.argmax_last(lg).astype(np.int32), margin=margin.astype(np.float32), top1=top1.astype(np.float32),
               token_num=np.asarray(r["token_num"], np.int32))
    if "fire_count" in r:
        out["fire_count"] = np.asarray(r["fire_count"], np.int32)
    if "alphas" in r:
        out["alpha_sum"] = r["alphas"].astype(np.float64).sum(axis=1).astype(np.float32)
    path = os.path.join(HERE, "bench_%s.npz" % full_name)
    np.savez_compressed(path, **out)
    ids = out["ids"]
    print("%s: B=%d L=%d, %d distinct ids, margin>0.04: %.3f, >0.1: %.3f, oracle %.1f s -> %s (%d bytes)"
          % (full_name, ids.shape[0], ids.shape[1], len(np.unique(ids)), (margin > 0.04).mean(), (margin > 0.1).mean(),
             time.time() - t0, os.path.relpath(path, ROOT), os.path.getsize(path)))


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["paraformer", "sensevoice", "seaco"]):
        run(nm)
