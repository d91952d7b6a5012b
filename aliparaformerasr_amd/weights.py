"""Weight container (.pfw) and the seeded synthetic-weight generator.

The reference loads ``model.onnx`` through onnxruntime
(``AliParaformerAsr/OfflineModel.cs:35-70``); this build loads a flat
little-endian container instead:

    bytes 0..3   magic  b"PFW1"
    bytes 4..7   u32    version (1)
    bytes 8..15  u64    header_len
    header_len bytes of UTF-8 JSON:
        {"config": {...ModelConfig...},
         "tensors": [{"name", "dtype": "f32" | "u8", "shape": [...], "offset", "nbytes"}, ...]}
    zero padding to a 256-byte boundary, then tensor data; every ``offset`` is
    relative to the start of the data section and 256-byte aligned.

Linear weights use the [out, in] (= [N, K]) layout, depthwise FSMN kernels
[D, k], the CIF conv [out, in, k].  The native library converts GEMM operands
to its 16-bit MFMA layout at load time; the file always carries float32.
A container made from an int8 export (``model.int8.onnx``, the reference CLI's
default, Program.cs:98-101) additionally carries, per quantised Linear, the
stored bytes ``<linear>.weight_q`` (u8 [N, K]), ``<linear>.weight_zp`` (u8 [N])
and ``<linear>.weight_scale`` (f32 [N]); ``math_mode 2`` multiplies those bytes
as they are, ``<linear>.weight`` beside them is their de-quantised float32 image
(math modes 0 / 1).

Tensor inventory follows SURVEY.md §8a ("Tensor inventory for the
synthetic-weight generator").
"""
from __future__ import annotations

import json
import struct

import numpy as np

MAGIC = b"PFW1"
ALIGN = 256

DEFAULT_CONFIG = dict(
    kind="paraformer", feat_dim=560, d_model=512, heads=4, ffn=2048, enc_layers=50,
    tp_layers=0, kernel=11, dec_layers=16, vocab=8404, cif_threshold=1.0, cif_tail=0.45,
    cif_l_order=1, cif_r_order=1, cif_smooth=1.0, cif_noise=0.0, timestamp_head=False,
    seaco=False, use_itn=False,
    # BiCIF timestamp head (CifPredictorV3): ConvTranspose1d x3 upsample -> BiLSTM -> Linear(1024,1)
    cif_smooth2=0.25, cif_noise2=0.01, upsample=3,
    # SeACo (kind "seacoparaformer", seaco=True): hotword embedder Embedding + LSTM(512) x seaco_lstm_layers,
    # bias decoder = SAN-M decoder without input/output layer attending bias_embed, hotword_output_layer
    seaco_layers=4, seaco_ffn=1024, seaco_kernel=21, seaco_lstm_layers=2, seaco_nobias=8377,
    # CIF export variant: "loop" = sequential integrate-and-fire, "cumsum" = FunASR cif_v1_export (prefix sums)
    cif_variant="loop",
    # math_mode 2 on a container WITHOUT stored int8 bytes (the synthetic models): Linear name prefixes that stay float,
    # e.g. ["decoder.output", "seaco.output"] for the MatMuls FunASR's export excludes from quantize_dynamic.  A container
    # converted from an int8 export does not need it: a Linear is quantised iff its `<name>.weight_q` bytes are present.
    int8_exclude=(),
)


def make_config(**kw) -> dict:
    cfg = dict(DEFAULT_CONFIG)
    for k, v in kw.items():
        if k not in cfg:
            raise KeyError(k)
        cfg[k] = v
    cfg["int8_exclude"] = [str(e) for e in cfg["int8_exclude"]]      # a JSON array in the container header
    return cfg


def paraformer_large_config(**kw) -> dict:
    return make_config(**kw)


def seaco_paraformer_config(**kw) -> dict:
    base = dict(kind="seacoparaformer", seaco=True, timestamp_head=True)
    base.update(kw)
    return make_config(**base)


def sensevoice_small_config(**kw) -> dict:
    base = dict(kind="sensevoicesmall", enc_layers=50, tp_layers=20, dec_layers=0, vocab=25055)
    base.update(kw)
    return make_config(**base)


def _linear(rng, out_f, in_f, bias=True, prefix=""):
    d = {prefix + ".weight": (rng.standard_normal((out_f, in_f), dtype=np.float32)
                              / np.float32(np.sqrt(in_f))).astype(np.float32)}
    if bias:
        d[prefix + ".bias"] = (0.1 * rng.standard_normal(out_f, dtype=np.float32)).astype(np.float32)
    return d


def _ln(rng, n, prefix, jitter=True):
    # gamma = 1, beta = 0 per SURVEY §8d; a small seeded jitter keeps gamma/beta
    # handling observable in parity tests (a kernel that ignores beta would
    # otherwise pass).
    g = np.ones(n, np.float32)
    b = np.zeros(n, np.float32)
    if jitter:
        g = (g + 0.05 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)
        b = (0.05 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)
    return {prefix + ".weight": g, prefix + ".bias": b}


def _enc_layer(rng, cfg, prefix, d_in):
    D, F, K = cfg["d_model"], cfg["ffn"], cfg["kernel"]
    w = {}
    w.update(_ln(rng, d_in, prefix + ".norm1"))
    w.update(_linear(rng, 3 * D, d_in, prefix=prefix + ".attn.qkv"))
    w[prefix + ".attn.fsmn.weight"] = (0.1 * rng.standard_normal((D, K), dtype=np.float32)).astype(np.float32)
    w.update(_linear(rng, D, D, prefix=prefix + ".attn.out"))
    w.update(_ln(rng, D, prefix + ".norm2"))
    w.update(_linear(rng, F, D, prefix=prefix + ".ffn.w1"))
    w.update(_linear(rng, D, F, prefix=prefix + ".ffn.w2"))
    return w


def synth_weights(cfg: dict, seed: int = 42) -> dict:
    """Seeded synthetic weights: Linear N(0, 1/sqrt(fan_in)), FSMN N(0, 0.1),
    LN gamma~1 beta~0, CIF output bias set so mean(alpha) ~ 0.3 (5 tokens/s)."""
    rng = np.random.default_rng(seed)
    D, F, K, V = cfg["d_model"], cfg["ffn"], cfg["kernel"], cfg["vocab"]
    w = {}
    for i in range(cfg["enc_layers"]):
        w.update(_enc_layer(rng, cfg, f"encoder.layers.{i}", cfg["feat_dim"] if i == 0 else D))
    w.update(_ln(rng, D, "encoder.after_norm"))
    for i in range(cfg["tp_layers"]):
        w.update(_enc_layer(rng, cfg, f"encoder.tp_layers.{i}", D))
    if cfg["tp_layers"]:
        w.update(_ln(rng, D, "encoder.tp_norm"))
    if cfg["kind"] == "sensevoicesmall":
        w.update(_linear(rng, V, D, prefix="ctc"))
        w["embed.weight"] = (0.5 * rng.standard_normal((16, cfg["feat_dim"]), dtype=np.float32)).astype(np.float32)
        return w
    # CIF predictor
    ksz = cfg["cif_l_order"] + cfg["cif_r_order"] + 1
    w["predictor.conv.weight"] = (rng.standard_normal((D, D, ksz), dtype=np.float32)
                                  / np.float32(np.sqrt(D * ksz))).astype(np.float32)
    w["predictor.conv.bias"] = (0.1 * rng.standard_normal(D, dtype=np.float32)).astype(np.float32)
    w["predictor.out.weight"] = (rng.standard_normal((1, D), dtype=np.float32)
                                 / np.float32(np.sqrt(D))).astype(np.float32)
    # calibrated on the seed-42 paraformer-large geometry so that sum(alpha) ~ 150 for 30 s (5 tokens/s)
    w["predictor.out.bias"] = np.asarray([-1.35], np.float32)
    if cfg.get("timestamp_head"):
        up = cfg["upsample"]
        w["predictor.upsample.weight"] = (rng.standard_normal((D, D, up), dtype=np.float32) / np.float32(np.sqrt(D))).astype(np.float32)
        w["predictor.upsample.bias"] = (0.1 * rng.standard_normal(D, dtype=np.float32)).astype(np.float32)
        for sfx in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh"):
                w["predictor.blstm.%s%s" % (nm, sfx)] = (rng.standard_normal((4 * D, D), dtype=np.float32) / np.float32(np.sqrt(D))).astype(np.float32)
            for nm in ("bias_ih", "bias_hh"):
                w["predictor.blstm.%s%s" % (nm, sfx)] = (0.1 * rng.standard_normal(4 * D, dtype=np.float32)).astype(np.float32)
        w["predictor.out2.weight"] = (rng.standard_normal((1, 2 * D), dtype=np.float32) * np.float32(4.0 / np.sqrt(2 * D))).astype(np.float32)
        w["predictor.out2.bias"] = np.asarray([0.3], np.float32)
    # decoder
    for i in range(cfg["dec_layers"]):
        p = f"decoder.layers.{i}"
        w.update(_ln(rng, D, p + ".norm1"))
        w.update(_linear(rng, F, D, prefix=p + ".ffn.w1"))
        w.update(_ln(rng, F, p + ".ffn.norm"))
        w.update(_linear(rng, D, F, bias=False, prefix=p + ".ffn.w2"))
        w.update(_ln(rng, D, p + ".norm2"))
        w[p + ".fsmn.weight"] = (0.1 * rng.standard_normal((D, K), dtype=np.float32)).astype(np.float32)
        w.update(_ln(rng, D, p + ".norm3"))
        w.update(_linear(rng, D, D, prefix=p + ".src.q"))
        w.update(_linear(rng, 2 * D, D, prefix=p + ".src.kv"))
        w.update(_linear(rng, D, D, prefix=p + ".src.out"))
    p = "decoder.final"
    w.update(_ln(rng, D, p + ".norm1"))
    w.update(_linear(rng, F, D, prefix=p + ".ffn.w1"))
    w.update(_ln(rng, F, p + ".ffn.norm"))
    w.update(_linear(rng, D, F, bias=False, prefix=p + ".ffn.w2"))
    w.update(_ln(rng, D, "decoder.after_norm"))
    w.update(_linear(rng, V, D, prefix="decoder.output"))
    if cfg.get("seaco"):
        Fs, Ks = cfg["seaco_ffn"], cfg["seaco_kernel"]
        w["seaco.embed.weight"] = (0.5 * rng.standard_normal((V, D), dtype=np.float32)).astype(np.float32)
        for l in range(cfg["seaco_lstm_layers"]):
            for nm in ("weight_ih", "weight_hh"):
                w["seaco.lstm.l%d.%s" % (l, nm)] = (rng.standard_normal((4 * D, D), dtype=np.float32) / np.float32(np.sqrt(D))).astype(np.float32)
            for nm in ("bias_ih", "bias_hh"):
                w["seaco.lstm.l%d.%s" % (l, nm)] = (0.1 * rng.standard_normal(4 * D, dtype=np.float32)).astype(np.float32)
        for i in range(cfg["seaco_layers"]):
            p = f"seaco.decoder.layers.{i}"
            w.update(_ln(rng, D, p + ".norm1"))
            w.update(_linear(rng, Fs, D, prefix=p + ".ffn.w1"))
            w.update(_ln(rng, Fs, p + ".ffn.norm"))
            w.update(_linear(rng, D, Fs, bias=False, prefix=p + ".ffn.w2"))
            w.update(_ln(rng, D, p + ".norm2"))
            w[p + ".fsmn.weight"] = (0.1 * rng.standard_normal((D, Ks), dtype=np.float32)).astype(np.float32)
            w.update(_ln(rng, D, p + ".norm3"))
            w.update(_linear(rng, D, D, prefix=p + ".src.q"))
            w.update(_linear(rng, 2 * D, D, prefix=p + ".src.kv"))
            w.update(_linear(rng, D, D, prefix=p + ".src.out"))
        p = "seaco.decoder.final"
        w.update(_ln(rng, D, p + ".norm1"))
        w.update(_linear(rng, Fs, D, prefix=p + ".ffn.w1"))
        w.update(_ln(rng, Fs, p + ".ffn.norm"))
        w.update(_linear(rng, D, Fs, bias=False, prefix=p + ".ffn.w2"))
        w.update(_ln(rng, D, "seaco.decoder.after_norm"))
        w.update(_linear(rng, V, D, prefix="seaco.output"))
    return w


def synth_cmvn(dim: int = 560, seed: int = 7):
    """CMVN shift ~ N(-8, 1), scale ~ U(0.1, 0.3)."""
    rng = np.random.default_rng(seed)
    shift = (-8.0 + rng.standard_normal(dim)).astype(np.float32)
    scale = rng.uniform(0.1, 0.3, dim).astype(np.float32)
    return shift, scale


def format_mvn_text(shift, scale) -> str:
    """An am.mvn in the kaldi-nnet text layout WavFrontend.LoadCmvn reads (AliParaformerAsr/WavFrontend.cs:113-157):
    <AddShift> / <Rescale> blocks whose <LearnRateCoef> line carries the vector between '[' and ']'."""
    dim = len(shift)

    def vec(v):
        return " ".join(repr(float(np.float32(x))) for x in v)
    return ("<Nnet> \n"
            f"<Splice> {dim} {dim}\n[ 0 ]\n"
            f"<AddShift> {dim} {dim} \n"
            f"<LearnRateCoef> 0 [ {vec(shift)} ]\n"
            f"<Rescale> {dim} {dim}\n"
            f"<LearnRateCoef> 0 [ {vec(scale)} ]\n"
            "</Nnet> \n")


def synth_tokens(vocab: int):
    """A tokens.txt of `vocab` lines in the shape of the real ones: <blank> <s> </s>, CJK characters, BPE pieces with
    '@@' continuations and '▁' word starts (what DecodeMulti's branches act on, OfflineRecognizer.cs:304-418), <unk> last."""
    toks = ["<blank>", "<s>", "</s>"]
    n_cjk = min(max(vocab - 4, 0) * 3 // 4, 0x9FA5 - 0x4E00)
    toks += [chr(0x4E00 + i) for i in range(n_cjk)]
    i = 0
    while len(toks) < vocab - 1:
        w = "w%d" % i
        toks.append(w + "@@" if i % 3 == 0 else ("\u2581" + w if i % 3 == 1 else w))
        i += 1
    toks.append("<unk>")
    return toks[:vocab]


def synth_model_dir(path: str, cfg: dict, weights: dict, cmvn=None, dither: float = 0.0, int8_name: bool = False) -> dict:
    """Writes what `new OfflineRecognizer(modelFilePath, configFilePath, mvnFilePath, tokensFilePath)` takes
    (OfflineRecognizer.cs:23) for a synthetic model: model[.int8].pfw, asr.yaml, am.mvn, tokens.txt.  Returns the paths."""
    import os
    os.makedirs(path, exist_ok=True)
    shift, scale = cmvn if cmvn is not None else synth_cmvn()
    out = {"model": os.path.join(path, "model.int8.pfw" if int8_name else "model.pfw"), "config": os.path.join(path, "asr.yaml"),
           "mvn": os.path.join(path, "am.mvn"), "tokens": os.path.join(path, "tokens.txt")}
    save_pfw(out["model"], cfg, weights)
    kind = cfg.get("kind", "paraformer")
    with open(out["config"], "w") as f:
        f.write("model: %s\nuse_itn: %s\nfrontend_conf:\n  fs: 16000\n  window: hamming\n  n_mels: 80\n  dither: %r\n"
                "  lfr_m: 7\n  lfr_n: 6\n  snip_edges: false\n" % (kind, "true" if cfg.get("use_itn") else "false", float(dither)))
    with open(out["mvn"], "w") as f:
        f.write(format_mvn_text(shift, scale))
    with open(out["tokens"], "w", encoding="utf-8") as f:
        f.write("\n".join(synth_tokens(int(cfg["vocab"]))) + "\n")
    return out


def synth_audio(n_samples: int, utt: int, seed: int = 1234) -> np.ndarray:
    """SURVEY §8d synthetic utterance: 0.1*N(0,1) noise + 3 sinusoids 100-4000 Hz,
    float32 in [-1, 1), default_rng(seed + utt)."""
    rng = np.random.default_rng(seed + utt)
    x = 0.1 * rng.standard_normal(n_samples)
    t = np.arange(n_samples) / 16000.0
    for _ in range(3):
        f = rng.uniform(100.0, 4000.0)
        a = rng.uniform(0.05, 0.2)
        ph = rng.uniform(0, 2 * np.pi)
        x = x + a * np.sin(2 * np.pi * f * t + ph)
    return np.clip(x, -0.999, 0.999).astype(np.float32)


def pack_pfw(cfg: dict, weights: dict) -> bytes:
    tensors = []
    off = 0
    chunks = []
    for name, arr in weights.items():
        u8 = isinstance(arr, np.ndarray) and arr.dtype == np.uint8      # `<linear>.weight_q` / `.weight_zp`: stored bytes of an int8 export
        a = np.ascontiguousarray(arr, dtype=np.uint8 if u8 else np.float32)
        nbytes = a.nbytes
        tensors.append({"name": name, "dtype": "u8" if u8 else "f32", "shape": list(a.shape), "offset": off, "nbytes": nbytes})
        chunks.append((off, a))
        off += (nbytes + ALIGN - 1) // ALIGN * ALIGN
    header = json.dumps({"config": cfg, "tensors": tensors}).encode("utf-8")
    pre = 16 + len(header)
    pad = (-pre) % ALIGN
    out = bytearray(pre + pad + off)
    out[0:4] = MAGIC
    struct.pack_into("<IQ", out, 4, 1, len(header))
    out[16:16 + len(header)] = header
    base = pre + pad
    for o, a in chunks:
        out[base + o: base + o + a.nbytes] = a.tobytes()
    return bytes(out)


def save_pfw(path: str, cfg: dict, weights: dict) -> None:
    with open(path, "wb") as f:
        f.write(pack_pfw(cfg, weights))


def load_pfw(path_or_bytes):
    data = path_or_bytes
    if isinstance(path_or_bytes, str):
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    if data[:4] != MAGIC:
        raise ValueError("not a PFW1 container")
    _ver, hlen = struct.unpack_from("<IQ", data, 4)
    hdr = json.loads(data[16:16 + hlen].decode("utf-8"))
    base = (16 + hlen + ALIGN - 1) // ALIGN * ALIGN
    w = {}
    for t in hdr["tensors"]:
        if t.get("dtype", "f32") == "u8":
            a = np.frombuffer(data, dtype=np.uint8, count=t["nbytes"], offset=base + t["offset"])
        else:
            a = np.frombuffer(data, dtype=np.float32, count=t["nbytes"] // 4, offset=base + t["offset"])
        w[t["name"]] = a.reshape(t["shape"]).copy()
    return hdr["config"], w
