"""Oracle for the STREAMING path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Exact restatements of the reference's managed glue (citations relative to AliParaformerAsr/):
  OnlineStream.cs:79-224      AddSamples / InputSpeech / GetDecodeChunk / RemoveChunk (chunks of 60 fbank frames, a
                              chunk of SILENCE queued by the constructor, first frame repeated, 1-frame LFR splice,
                              10-frame feature cache)
  OnlineWavFrontend.cs:63-80  ApplyLfr without left context;  :152-188 SinusoidalPositionEncoder with (i + 1)
  OnlineModel.cs:141-165      DynamicMask;  :199-247 stack_states (every layer gets the streams' LAYER-0 cache)
  OnlineRecognizer.cs:126-231 PredictorProj = CIF with the carried integrator;  :336-403 Forward;  :405-437 DecodeMulti
PARITY UNPINNED for the two ONNX graphs (FunASR paraformer-online export, not under /root/reference): `online_encoder`
and `online_decoder` restate the published export (SAN-M encoder without the embed stage + CIF weights; SAN-M decoder
whose FSMN memory is a no-padding conv over cat(cache, x)), built from the same pieces as oracle/model.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as Fn

from . import frontend as fe
from . import model as om

F32 = np.float32
CHUNK_FRAMES = 10 * 5 + 10          # OnlineModel.cs:29
SENTINEL = F32(-23.025850929940457)  # OnlineRecognizer.cs:471 (NOT x 32768 here)


def apply_lfr(fbank: np.ndarray, lfr_m=7, lfr_n=6) -> np.ndarray:
    t = fbank.shape[0]
    t_lfr = 0
    if t % lfr_n < lfr_m - lfr_n:
        t_lfr = t // lfr_n - 1
    if t % lfr_n >= lfr_m - lfr_n:
        t_lfr = t // lfr_n
    out = np.zeros((max(t_lfr, 0), lfr_m * 80), F32)
    for i in range(t_lfr):
        out[i] = fbank[i * lfr_n: i * lfr_n + lfr_m].reshape(-1)
    return out


def position_encode(x: np.ndarray, start_idx: int) -> np.ndarray:
    """x [timesteps, dim] -> x + PE rows start_idx.. ; float32 / double mix of the C#."""
    timesteps, dim = x.shape
    half = dim // 2
    inc = F32(F32(math.log(float(F32(10000.0)))) / F32(half - 1))
    inv = np.exp((np.arange(1, half + 1, dtype=F32) * (-inc)).astype(F32).astype(np.float64)).astype(F32)
    out = x.copy()
    for t in range(timesteps):
        p = F32(start_idx + t + 1)
        arg = (inv * p).astype(F32).astype(np.float64)
        out[t, :half] = (out[t, :half] + np.sin(arg).astype(F32)).astype(F32)
        out[t, half:] = (out[t, half:] + np.cos(arg).astype(F32)).astype(F32)
    return out


def dynamic_mask(alphas: np.ndarray, chunk_size=5, lfr=10) -> np.ndarray:
    a = np.array(alphas, F32, copy=True)
    a[:chunk_size] = 0
    a[chunk_size + lfr:] = 0
    return a


def cif(hiddens: np.ndarray, alphas: np.ndarray, threshold=1.0):
    """OnlineRecognizer.cs:152-197 for one stream.  Returns fired [n,D], carry_alpha, carry_hidden."""
    th = F32(threshold)
    D = hiddens.shape[1]
    integrate = F32(0.0)
    frames = np.zeros(D, F32)
    fired = []
    for j in range(min(len(alphas), hiddens.shape[0])):
        alpha = F32(alphas[j])
        h = hiddens[j].astype(F32)
        if F32(alpha + integrate) < th:
            integrate = F32(integrate + alpha)
            frames = (frames + (alpha * h).astype(F32)).astype(F32)
        else:
            frames = (frames + (F32(th - integrate) * h).astype(F32)).astype(F32)
            fired.append(frames)
            integrate = F32(integrate + alpha)
            integrate = F32(integrate - th)
            frames = (integrate * h).astype(F32)
    carry_h = (frames / integrate).astype(F32) if integrate > 0 else frames
    return (np.stack(fired) if fired else np.zeros((0, D), F32)), integrate, carry_h


def decode_text(tokens, ids) -> str:
    import re
    text = ""
    for i in ids:
        if i == 2:
            break
        tk = tokens[int(i)]
        if tk not in ("</s>", "<s>", "<blank>", "<unk>"):
            text += tk if re.match(r"^[一-龥]+$", tk) else "▁" + tk + "▁"
    return text.replace("@@▁▁", "").replace("@@▁", "").replace("▁▁", " ").replace("▁", "").lower()


class OnlineGraphs:
    """The two ONNX graphs, restated over oracle.model.Oracle's weights / rounding hooks."""

    def __init__(self, orc: om.Oracle):
        self.o = orc

    def encoder(self, speech: np.ndarray):
        """speech [B,Tc,560] (already scaled + position-encoded) -> enc [B,Tc,512], alphas [B,Tc]."""
        o, c = self.o, self.o.cfg
        x = torch.as_tensor(speech, dtype=torch.float32)
        for i in range(c.enc_layers):
            x = o.enc_layer(x, f"encoder.layers.{i}", first=(i == 0))
        x = o.ln(x, "encoder.after_norm")
        a = o.cif_alphas(x)[:, :-1]                     # no tail frame in the streaming graph
        return x.numpy(), a.numpy()

    def decoder(self, enc, embeds, embeds_len, caches):
        """caches [n_layers][B,512,10] -> log-probs [B,L,V], new caches (same layout)."""
        o, c = self.o, self.o.cfg
        q = o.q
        memory = torch.as_tensor(enc, dtype=torch.float32)
        x = torch.as_tensor(embeds, dtype=torch.float32)
        B, L, D = x.shape
        ln = torch.as_tensor(np.asarray(embeds_len), dtype=torch.int64)
        mask = (torch.arange(L)[None, :] < ln[:, None]).to(torch.float32).unsqueeze(-1)
        dk = D // c.heads
        new_caches = []
        for i in range(c.dec_layers):
            p = f"decoder.layers.{i}"
            t = o.ffn_dec(o.ln(x, p + ".norm1"), p)
            tn = o.ln(t, p + ".norm2") * mask
            xc = torch.cat([torch.as_tensor(caches[i], dtype=torch.float32), tn.transpose(1, 2)], dim=2)   # [B,D,10+L]
            new_caches.append(xc[:, :, -(c.kernel - 1):].numpy().copy())
            y = Fn.conv1d(xc, o.w[p + ".fsmn.weight"].unsqueeze(1), groups=D).transpose(1, 2)              # [B,L,D]
            x = x + (y + tn) * mask
            xn = o.ln(x, p + ".norm3")
            qq = o.lin(xn, p + ".src.q")
            kv = o.lin(memory, p + ".src.kv")
            k, v = torch.split(kv, D, dim=-1)
            x = x + o.lin(o.mha(q(qq * (dk ** -0.5)), q(k), q(v)), p + ".src.out")
        x = o.ffn_dec(o.ln(x, "decoder.final.norm1"), "decoder.final")
        hid = o.ln(x, "decoder.after_norm")
        return om.log_softmax(o.lin(hid, "decoder.output")).numpy(), new_caches


class OnlineStream:
    def __init__(self, rec: "OnlineRecognizer"):
        self.rec = rec
        self.tokens = [0, 0]
        self.states = [np.zeros((512, 10), F32) for _ in range(rec.graphs.o.cfg.dec_layers)]
        self.cif_hidden = [np.zeros(512, F32)]
        self.cif_alpha = [F32(0.0)]
        self.speech = np.zeros((0, 80), F32)
        self.cache_samples = np.zeros(160 * CHUNK_FRAMES, F32)
        self.cache_feats = np.zeros((10, 560), F32)
        self.splice = None
        self.first_input = True
        self.start_idx = 0

    def add_samples(self, samples):
        self.cache_samples = np.concatenate([self.cache_samples, np.asarray(samples, F32)])
        n = 160 * CHUNK_FRAMES
        if len(self.cache_samples) > n:
            self._input_speech(self.cache_samples[:n])
            self.cache_samples = self.cache_samples[n:]

    def _input_speech(self, samples):
        fb = self.rec.fbank(samples)
        if self.first_input and fb.shape[0] > 0:
            fb = np.concatenate([fb[:1], fb])
            self.first_input = False
        self.speech = np.concatenate([self.speech, fb])

    def get_decode_chunk(self):
        if CHUNK_FRAMES > self.speech.shape[0]:
            return None
        head = self.splice if self.splice is not None else self.speech[:1]
        pad = np.concatenate([head, self.speech[:CHUNK_FRAMES]])
        self.splice = pad[-1:].copy()
        x = apply_lfr(pad)
        x = fe.apply_cmvn(x, self.rec.shift, self.rec.scale)
        x = (x.astype(np.float64) * math.pow(512, 0.5)).astype(F32)
        x = position_encode(x, self.start_idx)
        chunk = np.concatenate([self.cache_feats, x])
        self.start_idx += x.shape[0]
        self.cache_feats = chunk[-10:].copy()
        self.speech = self.speech[CHUNK_FRAMES:]
        return chunk


class OnlineRecognizer:
    def __init__(self, cfg: dict, weights: dict, cmvn, tokens, quant="fp16", snip_edges=False):
        self.graphs = OnlineGraphs(om.Oracle(om.ModelConfig(**cfg), weights, quant=quant))
        self.shift, self.scale = cmvn
        self.tokens = tokens
        self.conf = fe.FrontendConf(dither=0.0, snip_edges=snip_edges)
        self.trace = []                       # per Forward: dict(enc, alphas, embeds, lens, logits, ids)

    def fbank(self, samples):
        return fe.kaldi_fbank(samples, self.conf)

    def create_stream(self):
        return OnlineStream(self)

    def get_results(self, streams):
        work, chunks = [], []
        for s in streams:
            c = s.get_decode_chunk()
            if c is not None:
                work.append(s)
                chunks.append(c)
        if work:
            B = len(work)
            speech = np.stack(chunks).astype(F32)
            speech = np.where(speech == 0, SENTINEL, speech).astype(F32)
            nd = self.graphs.o.cfg.dec_layers
            cin = [np.stack([s.states[0] for s in work]) for _ in range(nd)]        # the layer-0 quirk
            enc, alphas = self.graphs.encoder(speech)
            fired = []
            for b, s in enumerate(work):
                s.cif_hidden += [enc[b, t] for t in range(enc.shape[1])]
                s.cif_alpha += list(dynamic_mask(alphas[b]))
            len_time = len(work[0].cif_alpha)
            for s in work:
                f, ca, ch = cif(np.stack(s.cif_hidden[:len_time]), np.asarray(s.cif_alpha[:len_time], F32),
                                self.graphs.o.cfg.cif_threshold)
                s.cif_alpha, s.cif_hidden = [ca], [ch]
                fired.append(f)
            L = max(f.shape[0] for f in fired)
            rec = {"enc": enc, "alphas": alphas, "L": L}
            if L > 0:
                emb = np.zeros((B, L, 512), F32)
                lens = np.zeros(B, np.int32)
                for b, f in enumerate(fired):
                    emb[b, : f.shape[0]] = f
                    lens[b] = f.shape[0]
                logits, cout = self.graphs.decoder(enc, emb, lens, cin)
                ids = om.argmax_last(logits)
                for b, s in enumerate(work):
                    s.tokens += [int(v) for v in ids[b]]
                    s.states = [cout[l][b] for l in range(nd)]
                rec.update(embeds=emb, lens=lens, logits=logits, ids=ids, cin=cin, cout=cout)
            self.trace.append(rec)
        return [decode_text(self.tokens, s.tokens) for s in streams]
