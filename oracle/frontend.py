"""Oracle: WavFrontend (fbank -> LFR -> CMVN) and PadHelper.PadSequence.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference sites restated here
  * ``AliParaformerAsr/WavFrontend.cs:31-37``   GetFbank  (x*32768 then external OnlineFbank)
  * ``AliParaformerAsr/WavFrontend.cs:73-111``  ApplyLfr  (quirk Q1: 3 ZERO left frames, floor count)
  * ``AliParaformerAsr/WavFrontend.cs:53-71``   ApplyCmvn ((x + shift) * scale)
  * ``AliParaformerAsr/WavFrontend.cs:112-153`` LoadCmvn  (am.mvn text parse)
  * ``AliParaformerAsr/Utils/PadHelper.cs:23-65`` PadSequence (right pad, ==0 -> sentinel)
  * ``AliParaformerAsr/Model/FrontendConfEntity.cs:7-15`` defaults

The fbank itself lives in ManySpeech.SpeechFeatures 1.1.7 (kaldi-native-fbank),
which is NOT in /root/reference; `kaldi_fbank` restates the published kaldi
algorithm (feature-window.cc / mel-computations.cc / feature-fbank.cc) with
the options the C# passes (dither, snip_edges, window_type, sample_rate,
num_bins; everything else = knf defaults).  Parity unpinned for that function:
the reference holds no vector of its dependency.  What exists instead (round 5):
agreement with a second, independently written restatement of the same published
algorithm — `transformers.audio_utils.spectrogram` with the Kaldi options — to
8e-5 on log-energies (tests/test_oracle_golden.py::
test_fbank_oracle_agrees_with_an_independent_kaldi_restatement).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

F32 = np.float32

# PadHelper.cs:63 — the literal is a float32 product in C#.
PAD_SENTINEL = F32(F32(-23.025850929940457) * F32(32768.0))


@dataclass
class FrontendConf:
    """Mirror of FrontendConfEntity (Model/FrontendConfEntity.cs:7-15), defaults included."""
    fs: int = 16000
    window: str = "hamming"
    n_mels: int = 80
    frame_length: int = 25   # NOT forwarded to OnlineFbank (WavFrontend.cs:21-27)
    frame_shift: int = 10    # NOT forwarded
    dither: float = 1.0
    lfr_m: int = 7
    lfr_n: int = 6
    snip_edges: bool = False


# ----------------------------------------------------------------------------
# kaldi fbank (external dependency restated)
# ----------------------------------------------------------------------------
FRAME_LEN = 400      # 25 ms @ 16 kHz (knf default frame_length_ms)
FRAME_SHIFT = 160    # 10 ms
NFFT = 512           # round_to_power_of_two
PREEMPH = F32(0.97)
LOW_FREQ = 20.0
FLT_EPSILON = F32(1.1920929e-07)


def num_frames(n_samples: int, snip_edges: bool) -> int:
    """kaldi NumFrames(..., flush=true)."""
    if snip_edges:
        if n_samples < FRAME_LEN:
            return 0
        return 1 + (n_samples - FRAME_LEN) // FRAME_SHIFT
    return (n_samples + FRAME_SHIFT // 2) // FRAME_SHIFT


def window_function(window_type: str = "hamming") -> np.ndarray:
    """FeatureWindowFunction: computed in double, stored as float."""
    i = np.arange(FRAME_LEN, dtype=np.float64)
    a = 2.0 * math.pi / (FRAME_LEN - 1)
    if window_type == "hamming":
        w = 0.54 - 0.46 * np.cos(a * i)
    elif window_type == "hanning":
        w = 0.5 - 0.5 * np.cos(a * i)
    elif window_type == "povey":
        w = np.power(0.5 - 0.5 * np.cos(a * i), 0.85)
    elif window_type == "rectangular":
        w = np.ones(FRAME_LEN)
    else:
        raise ValueError("unsupported window " + window_type)
    return w.astype(F32)


def mel_scale(f):
    return F32(1127.0) * np.log(F32(1.0) + np.asarray(f, dtype=F32) / F32(700.0)).astype(F32)


def mel_banks(num_bins: int = 80, sample_rate: int = 16000) -> np.ndarray:
    """MelBanks::MelBanks — dense [num_bins, NFFT/2] float32 weight matrix
    (FFT bin NFFT/2 is never used by kaldi)."""
    nyquist = 0.5 * sample_rate
    high_freq = nyquist
    fft_bin_width = F32(sample_rate / NFFT)
    mel_low = mel_scale(LOW_FREQ)
    mel_high = mel_scale(high_freq)
    delta = F32((mel_high - mel_low) / F32(num_bins + 1))
    nb = NFFT // 2
    w = np.zeros((num_bins, nb), dtype=F32)
    mel = mel_scale(fft_bin_width * np.arange(nb, dtype=F32))
    for b in range(num_bins):
        left = F32(mel_low + F32(b) * delta)
        center = F32(mel_low + F32(b + 1) * delta)
        right = F32(mel_low + F32(b + 2) * delta)
        for i in range(nb):
            m = mel[i]
            if m > left and m < right:
                if m <= center:
                    w[b, i] = F32((m - left) / (center - left))
                else:
                    w[b, i] = F32((right - m) / (right - center))
    return w


def extract_frames(wave: np.ndarray, snip_edges: bool) -> np.ndarray:
    """ExtractWindow without the processing: [T80, 400] float32 raw frames,
    mirror-reflected at the edges when snip_edges == false."""
    n = wave.shape[0]
    t = num_frames(n, snip_edges)
    if t == 0:
        return np.zeros((0, FRAME_LEN), dtype=F32)
    if snip_edges:
        start = FRAME_SHIFT * np.arange(t)
    else:
        start = FRAME_SHIFT * np.arange(t) + FRAME_SHIFT // 2 - FRAME_LEN // 2
    idx = start[:, None] + np.arange(FRAME_LEN)[None, :]
    # kaldi: while (s < 0 || s >= n) { s = s < 0 ? -s-1 : 2n-1-s }
    for _ in range(64):
        neg = idx < 0
        big = idx >= n
        if not (neg.any() or big.any()):
            break
        idx = np.where(neg, -idx - 1, idx)
        idx = np.where(idx >= n, 2 * n - 1 - idx, idx)
    return wave[idx].astype(F32)


def kaldi_fbank(samples: np.ndarray, conf: FrontendConf | None = None,
                scale_to_int16: bool = True, dither_rng=None) -> np.ndarray:
    """GetFbank (WavFrontend.cs:31-37): x*32768, then kaldi fbank -> [T80, n_mels] f32.

    dither != 0 (the reference default is 1.0, Model/FrontendConfEntity.cs:12): kaldi's ProcessWindow adds
    `dither * N(0,1)` to every sample of every extracted frame window (overlapping frames draw independently)
    BEFORE the DC removal; the reference is itself non-deterministic there (quirk Q11), so parity with it can
    only be statistical.  `dither_rng` seeds the draw (numpy Generator); bit-parity work runs with dither = 0.
    """
    conf = conf or FrontendConf(dither=0.0)
    if samples is None:
        # LINQ Select on null -> ArgumentNullException("source") (WavFrontend.cs:34)
        raise ValueError("source")
    x = np.asarray(samples, dtype=F32)
    if scale_to_int16:
        x = (x * F32(32768.0)).astype(F32)
    frames = extract_frames(x, conf.snip_edges)              # [T,400]
    if frames.shape[0] == 0:
        return np.zeros((0, conf.n_mels), dtype=F32)
    if conf.dither != 0.0:
        rng = dither_rng if dither_rng is not None else np.random.default_rng()
        frames = (frames + F32(conf.dither) * rng.standard_normal(frames.shape, dtype=F32)).astype(F32)
    # remove_dc_offset: window->Add(-window->Sum() / frame_length)
    mean = (frames.sum(axis=1, dtype=F32) / F32(FRAME_LEN)).astype(F32)
    frames = (frames - mean[:, None]).astype(F32)
    # Preemphasize: for i = n-1..1: d[i] -= c*d[i-1]; d[0] -= c*d[0]
    pre = np.empty_like(frames)
    pre[:, 1:] = frames[:, 1:] - PREEMPH * frames[:, :-1]
    pre[:, 0] = frames[:, 0] - PREEMPH * frames[:, 0]
    pre = pre.astype(F32)
    win = window_function(conf.window)
    pre = (pre * win[None, :]).astype(F32)
    padded = np.zeros((pre.shape[0], NFFT), dtype=F32)
    padded[:, :FRAME_LEN] = pre
    spec = np.fft.rfft(padded.astype(np.float64), axis=1)    # double FFT, rounded below
    power = (spec.real ** 2 + spec.imag ** 2).astype(F32)[:, : NFFT // 2]
    mel = power.astype(F32) @ mel_banks(conf.n_mels, conf.fs).T.astype(F32)
    mel = np.maximum(mel.astype(F32), FLT_EPSILON)
    return np.log(mel).astype(F32)


# ----------------------------------------------------------------------------
# LFR / CMVN / pad — reference's own C#
# ----------------------------------------------------------------------------
def apply_lfr(fbank: np.ndarray, lfr_m: int = 7, lfr_n: int = 6) -> np.ndarray:
    """ApplyLfr (WavFrontend.cs:73-111), quirks preserved:
    - feature width 80 hard-coded (:75);
    - t_lfr = floor(T80 / lfr_n) with integer division (:76);
    - the intended first-frame replication is overwritten: the loop :82-85 writes
      input_0 at offset tile_x*80 and :86 then copies the input over the same
      offset, so the (lfr_m-1)/2 left-context frames stay ZERO;
    - tail branch :96-108 replicates the last frame when fewer than lfr_m remain.
    """
    flat = np.asarray(fbank, dtype=F32).reshape(-1)
    t = flat.shape[0] // 80
    t_lfr = t // lfr_n
    tile_x = (lfr_m - 1) // 2
    t = t + tile_x
    temp = np.zeros(t * 80, dtype=F32)
    temp[tile_x * 80: tile_x * 80 + flat.shape[0]] = flat
    out = np.zeros(t_lfr * lfr_m * 80, dtype=F32)
    for i in range(t_lfr):
        if lfr_m <= t - i * lfr_n:
            out[i * lfr_m * 80:(i + 1) * lfr_m * 80] = temp[i * lfr_n * 80: i * lfr_n * 80 + lfr_m * 80]
        else:
            num_padding = lfr_m - (t - i * lfr_n)
            frame = np.zeros(lfr_m * 80, dtype=F32)
            have = (t - i * lfr_n) * 80
            frame[:have] = temp[i * lfr_n * 80: i * lfr_n * 80 + have]
            for j in range(num_padding):
                frame[(lfr_m - num_padding + j) * 80:(lfr_m - num_padding + j + 1) * 80] = temp[(t - 1) * 80: t * 80]
            out[i * lfr_m * 80:(i + 1) * lfr_m * 80] = frame
    return out.reshape(t_lfr, lfr_m * 80)


def apply_cmvn(feats: np.ndarray, shift: np.ndarray, scale: np.ndarray) -> np.ndarray:
    """ApplyCmvn (WavFrontend.cs:53-71): (x + neg_mean[k]) * inv_stddev[k], float32."""
    shift = np.asarray(shift, dtype=F32)
    scale = np.asarray(scale, dtype=F32)
    dim = shift.shape[0]
    x = np.asarray(feats, dtype=F32).reshape(-1, dim)
    return ((x + shift[None, :]).astype(F32) * scale[None, :]).astype(F32)


def parse_mvn_text(text: str):
    """LoadCmvn (WavFrontend.cs:112-153): the <LearnRateCoef> line following
    <AddShift> gives the shift vector, the one following <Rescale> the scale;
    numbers = text between first '[' and last ']' split on ' '."""
    means, variances = [], []
    state = 0
    for line in text.splitlines():
        if not line:
            continue
        if line.startswith("<AddShift>"):
            state = 1
            continue
        if line.startswith("<Rescale>"):
            state = 2
            continue
        if line.startswith("<LearnRateCoef>") and state in (1, 2):
            inner = line[line.index("[") + 1: line.rindex("]")]
            vals = [float(tok.strip()) for tok in inner.split(" ") if tok != ""]
            if state == 1:
                means = vals
            else:
                variances = vals
    return np.asarray(means, dtype=F32), np.asarray(variances, dtype=F32)


def format_mvn_text(shift: np.ndarray, scale: np.ndarray) -> str:
    """Writes an am.mvn in the kaldi-nnet layout the parser above expects."""
    dim = len(shift)

    def vec(v):
        return " ".join(repr(float(F32(x))) for x in v)
    return (
        "<Nnet> \n"
        f"<Splice> {dim} {dim}\n[ 0 ]\n"
        f"<AddShift> {dim} {dim} \n"
        f"<LearnRateCoef> 0 [ {vec(shift)} ]\n"
        f"<Rescale> {dim} {dim}\n"
        f"<LearnRateCoef> 0 [ {vec(scale)} ]\n"
        "</Nnet> \n"
    )


def wav_frontend(samples: np.ndarray, conf: FrontendConf, shift, scale) -> np.ndarray:
    """OfflineStream.AddSamples numeric part (OfflineStream.cs:40-41): -> [T, 560]."""
    fb = kaldi_fbank(samples, conf)
    feats = fb
    if conf.lfr_m != 1 or conf.lfr_n != 1:
        feats = apply_lfr(fb, conf.lfr_m, conf.lfr_n)
    if shift is not None and len(shift):
        if feats.shape[0] == 0:
            return np.zeros((0, len(shift)), dtype=F32)
        feats = apply_cmvn(feats, shift, scale)
    return feats


def pad_sequence(speeches: list[np.ndarray]) -> np.ndarray:
    """PadHelper.PadSequence (PadHelper.cs:23-65): flat float arrays are
    right-padded with 0 to the batch max, stacked row-major, then EVERY value
    == 0.0f (padding and genuine zeros alike, quirk Q3) becomes the sentinel.
    Returns [B, max_len] float32 (caller reshapes to [B, Tmax, 560])."""
    flats = [np.asarray(s, dtype=F32).reshape(-1) for s in speeches]
    max_len = max(f.shape[0] for f in flats)
    out = np.zeros((len(flats), max_len), dtype=F32)
    for i, f in enumerate(flats):
        out[i, : f.shape[0]] = f
    out[out == 0] = PAD_SENTINEL
    return out
