"""Oracle: the arithmetic inside ``InferenceSession.Run`` for the offline models.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the
arithmetic lives in downloaded FunASR ONNX graphs executed by
Microsoft.ML.OnnxRuntime 1.22.* — neither is under /root/reference.  This file
restates the published FunASR export definitions (SANMEncoder,
CifPredictorV2 export, ParaformerSANMDecoder export, SenseVoiceEncoderSmall +
CTC) and is cross-checked only against the C# config defaults
(``AliParaformerAsr/Model/EncoderConfEntity.cs:13-25``,
``DecoderConfEntity.cs:7-16``, ``PredictorConfEntity.cs:13-17``) and the call
sites ``AliParaformerAsr/OfflineProjOfParaformer.cs:39-87`` (inputs ``speech``
[B,T,560] f32 and ``speech_lengths`` = Tmax for every row, quirk Q2; outputs
[0]=logits [1]=token_num [3]=us_cif_peak).

Because ``speech_lengths[b] == Tmax`` for all rows, every encoder mask is all
ones; only the decoder's target mask (from ``token_num``) is non-trivial.

`quant` hooks let the oracle emulate the HIP engine's 16-bit GEMM-operand
rounding points ("fp32" = none = what ORT computes).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as Fn

F32 = np.float32
LN_EPS = 1e-12   # FunASR LayerNorm(nout, eps=1e-12)


@dataclass
class ModelConfig:
    kind: str = "paraformer"          # "paraformer" | "sensevoicesmall" | "seacoparaformer"
    feat_dim: int = 560
    d_model: int = 512
    heads: int = 4
    ffn: int = 2048
    enc_layers: int = 50
    tp_layers: int = 0                # SenseVoice "tp" blocks (20)
    kernel: int = 11
    dec_layers: int = 16
    vocab: int = 8404
    cif_threshold: float = 1.0
    cif_tail: float = 0.45
    cif_l_order: int = 1
    cif_r_order: int = 1
    cif_smooth: float = 1.0
    cif_noise: float = 0.0
    timestamp_head: bool = False      # BiCIF upsample head (us_alphas/us_cif_peak)
    seaco: bool = False
    use_itn: bool = False
    cif_smooth2: float = 0.25
    cif_noise2: float = 0.01
    upsample: int = 3
    seaco_layers: int = 4
    seaco_ffn: int = 1024
    seaco_kernel: int = 21
    seaco_lstm_layers: int = 2
    seaco_nobias: int = 8377
    cif_variant: str = "loop"         # "loop" (sequential integrate-and-fire) | "cumsum" (cif_v1_export)
    int8_exclude: tuple = ()          # int8 modes, fp32-only weights: Linear name prefixes that stay float

    def to_dict(self):
        return dict(self.__dict__)


def quantizer(mode: str):
    """Returns q(x): rounds a float32 tensor to the GEMM operand type and back."""
    if mode in ("fp32", None):
        return lambda x: x
    if mode == "bf16":
        return lambda x: x.to(torch.bfloat16).to(torch.float32)
    if mode == "fp16":
        return lambda x: x.to(torch.float16).to(torch.float32)
    raise ValueError(mode)


def layer_norm(x, w, b):
    """LayerNorm of the float32 input VALUES: two-pass statistics carried in float64, result
    rounded to float32.  (The sentinel rows of PadHelper.cs:63 reach |x| ~ 1.7e7 with a spread of
    a few ulps; float32 statistics are ill-conditioned there and what onnxruntime's fused
    LayerNormalization returns for them is not recoverable from the reference — the oracle
    defines the mathematically exact value.)"""
    xd = x.to(torch.float64)
    mu = xd.mean(dim=-1, keepdim=True)
    d = xd - mu
    var = (d * d).mean(dim=-1, keepdim=True)
    y = d / torch.sqrt(var + LN_EPS) * w.to(torch.float64) + b.to(torch.float64)
    return y.to(torch.float32)


def log_softmax(x):
    """LogSoftmax as onnxruntime's CPU kernel forms it (MLAS: (Input + NegativeMaximum) - Logarithm), float32:
    y = (x - max) - log(sum(exp(x - max))).  The reference's arg-max loop (OfflineRecognizer.cs:139-152) scans THIS
    tensor, so distinct logits whose log-probs round to the same float32 tie (and resolve to the larger index)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    d = x - x.max(dim=-1, keepdim=True).values
    return d - torch.log(torch.exp(d).sum(dim=-1, keepdim=True))


def sinusoidal_pe(T: int, depth: int) -> torch.Tensor:
    """SinusoidalPositionEncoder.encode: positions 1..T, [sin || cos], float32."""
    half = depth // 2
    inc = F32(math.log(10000.0) / (half - 1))
    inv = np.exp(np.arange(half, dtype=F32) * (-inc)).astype(F32)
    pos = np.arange(1, T + 1, dtype=F32)
    st = (pos[:, None] * inv[None, :]).astype(F32)
    pe = np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(F32)
    return torch.from_numpy(pe)


def fsmn(v, w, kernel, mask=None):
    """DFSMN memory block: depthwise conv1d (no bias, zero pad (k-1)/2 each side,
    sanm_shift=0) over time + identity.  v [B,T,D], w [D,k]."""
    if mask is not None:
        v = v * mask
    left = (kernel - 1) // 2
    right = kernel - 1 - left
    x = Fn.pad(v.transpose(1, 2), (left, right))
    y = Fn.conv1d(x, w.unsqueeze(1), groups=w.shape[0]).transpose(1, 2)
    y = y + v
    if mask is not None:
        y = y * mask
    return y


def mha(q, k, v, heads):
    """softmax(q k^T) v per head; q already scaled.  q [B,Lq,D], k/v [B,Lk,D].
    All-ones masks (quirk Q2) => no additive mask term."""
    B, Lq, D = q.shape
    Lk = k.shape[1]
    dk = D // heads
    qh = q.view(B, Lq, heads, dk).transpose(1, 2)
    kh = k.view(B, Lk, heads, dk).transpose(1, 2)
    vh = v.view(B, Lk, heads, dk).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-2, -1))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vh)
    return o.transpose(1, 2).reshape(B, Lq, D)


class Oracle:
    def __init__(self, cfg: ModelConfig, weights: dict, quant: str = "fp32", fast: bool = False):
        """fast=True: the same graph with torch's fused fp32 CPU kernels (layer_norm, scaled_dot_product_attention)
        instead of the float64-statistics LayerNorm and the explicit softmax — the leaner CPU stand-in that
        bench.py times as `cpu_baseline` (parity work always uses fast=False)."""
        self.fast = fast
        self.cfg = cfg
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()
                  if isinstance(v, np.ndarray) and v.dtype == np.float32}
        # "int8": the graph of the reference's default model.int8.onnx (every Linear = DynamicQuantizeLinear +
        # MatMulInteger, oracle/int8.py) with the engine's 16-bit storage points (q / k / v, attention context, encoder
        # FFN hidden); "int8_ref": the same graph without any 16-bit rounding = what onnxruntime computes.
        self.int8 = quant in ("int8", "int8_ref")
        self.q = quantizer("fp16" if quant == "int8" else ("fp32" if quant == "int8_ref" else quant))
        self.quant = quant
        if self.int8:
            from .int8 import QuantizedLinears
            self.qlin = QuantizedLinears({k: np.ascontiguousarray(v) for k, v in weights.items()
                                          if isinstance(v, np.ndarray) and v.dtype in (np.float32, np.uint8)},
                                         exclude=tuple(cfg.int8_exclude))

    # -- helpers -----------------------------------------------------------
    def lin(self, x, name, bias=True):
        if self.int8 and self.qlin.quantised(name):
            return torch.from_numpy(self.qlin(np.ascontiguousarray(x.detach().numpy(), dtype=np.float32), name, bias))
        w = self.q(self.w[name + ".weight"])
        y = torch.matmul(self.q(x), w.t())
        if bias:
            y = y + self.w[name + ".bias"]
        return y

    def ln(self, x, name):
        if self.fast:
            w = self.w[name + ".weight"]
            return Fn.layer_norm(x, (w.shape[0],), w, self.w[name + ".bias"], LN_EPS)
        return layer_norm(x, self.w[name + ".weight"], self.w[name + ".bias"])

    def mha(self, q, k, v):
        if self.fast:
            B, Lq, D = q.shape
            H, dk = self.cfg.heads, D // self.cfg.heads
            sp = lambda t: t.view(B, t.shape[1], H, dk).transpose(1, 2)
            o = Fn.scaled_dot_product_attention(sp(q), sp(k), sp(v), scale=1.0)      # q is pre-scaled
            return o.transpose(1, 2).reshape(B, Lq, D)
        return mha(q, k, v, self.cfg.heads)

    # -- encoder -----------------------------------------------------------
    def enc_layer(self, x, p, first):
        c = self.cfg
        q = self.q
        xn = self.ln(x, p + ".norm1")
        qkv = self.lin(xn, p + ".attn.qkv")
        qh, kh, vh = torch.split(qkv, c.d_model, dim=-1)
        vq = q(vh)                                   # engine stores q/k/v as 16-bit
        f = fsmn(vq, self.w[p + ".attn.fsmn.weight"], c.kernel)
        dk = c.d_model // c.heads
        ctx = self.mha(q(qh * (dk ** -0.5)), q(kh), vq)
        if self.int8:
            ctx = q(ctx)                             # the engine's attention writes f16; it is quantised from there
        att = self.lin(ctx, p + ".attn.out") + f
        x = att if first else x + att
        xn = self.ln(x, p + ".norm2")
        h = torch.relu(self.lin(xn, p + ".ffn.w1"))
        if self.int8:
            h = q(h)                                 # f16 FFN hidden in the engine
        return x + self.lin(h, p + ".ffn.w2")

    def encoder(self, speech):
        """speech [B,T,feat] f32 -> H [B,T,512]."""
        c = self.cfg
        x = torch.as_tensor(speech, dtype=torch.float32)
        B, T, Fd = x.shape
        x = x * F32(math.sqrt(c.d_model))            # Mul, then Add (separate roundings)
        x = x + sinusoidal_pe(T, Fd)[None]
        for i in range(c.enc_layers):
            x = self.enc_layer(x, f"encoder.layers.{i}", first=(i == 0))
        x = self.ln(x, "encoder.after_norm")
        for i in range(c.tp_layers):
            x = self.enc_layer(x, f"encoder.tp_layers.{i}", first=False)
        if c.tp_layers:
            x = self.ln(x, "encoder.tp_norm")
        return x

    # -- CIF predictor -----------------------------------------------------
    def cif_alphas(self, H):
        """CifPredictorV2 (export): alphas [B,T+1] incl. tail frame, token_num [B]."""
        c = self.cfg
        x = Fn.pad(self.q(H).transpose(1, 2), (c.cif_l_order, c.cif_r_order))
        y = Fn.conv1d(x, self.q(self.w["predictor.conv.weight"]), self.w["predictor.conv.bias"])
        y = torch.relu(y).transpose(1, 2)
        z = torch.matmul(y, self.w["predictor.out.weight"].t()) + self.w["predictor.out.bias"]
        a = torch.sigmoid(z).squeeze(-1)
        a = torch.relu(a * c.cif_smooth - c.cif_noise)
        tail = torch.full((a.shape[0], 1), c.cif_tail, dtype=a.dtype)
        a = torch.cat([a, tail], dim=1)
        return a

    def cif(self, H, alphas):
        if self.cfg.cif_variant == "cumsum":
            return self.cif_fire_cumsum(H, alphas, self.cfg.cif_threshold)
        return self.cif_fire(H, alphas, self.cfg.cif_threshold)

    @staticmethod
    def cif_fire(H, alphas, threshold=1.0):
        """Sequential integrate-and-fire (same statement as the reference's
        streaming C# loop, AliParaformerAsr/OnlineRecognizer.cs:147-200, and
        FunASR cif_export).  H [B,T,D] (a zero frame is appended for the tail),
        alphas [B,T+1].  float32 arithmetic, non-fused mul/add.
        Returns embeds [B,Lmax,D], fire_count [B], token_num [B] (floor sum)."""
        Hn = np.asarray(H, dtype=F32)
        an = np.asarray(alphas, dtype=F32)
        B, T1 = an.shape
        D = Hn.shape[2]
        Hn = np.concatenate([Hn, np.zeros((B, T1 - Hn.shape[1], D), F32)], axis=1)
        th = F32(threshold)
        frames_all, counts, tnum = [], [], []
        for b in range(B):
            integrate = F32(0.0)
            frame = np.zeros(D, F32)
            fired = []
            s = F32(0.0)
            for t in range(T1):
                alpha = an[b, t]
                s = F32(s + alpha)
                completion = F32(F32(1.0) - integrate)
                integrate = F32(integrate + alpha)
                if integrate >= th:
                    frame = (frame + (completion * Hn[b, t]).astype(F32)).astype(F32)
                    fired.append(frame)
                    integrate = F32(integrate - F32(1.0))
                    remain = F32(alpha - completion)
                    frame = (remain * Hn[b, t]).astype(F32)
                else:
                    frame = (frame + (alpha * Hn[b, t]).astype(F32)).astype(F32)
            frames_all.append(fired)
            counts.append(len(fired))
            tnum.append(int(np.floor(s)))
        L = max(counts) if counts else 0
        E = np.zeros((B, L, D), F32)
        for b in range(B):
            if counts[b]:
                E[b, : counts[b]] = np.stack(frames_all[b])
        return E, np.asarray(counts, np.int32), np.asarray(tnum, np.int32)

    @staticmethod
    def cif_fire_cumsum(H, alphas, threshold=1.0):
        """The OTHER CIF formulation found in FunASR exports (`cif_v1_export`: no loop, prefix sums) — SURVEY §8a
        flags "loop vs cumsum export" as something only the real ONNX graph can settle, so both are built
        (container key `cif_variant`).  threshold 1:
          prefix = float32(cumsum_float64(alphas));  fire at t  <=>  floor(prefix[t]) > floor(prefix[t-1])
          remain[t] = (1 + (prefix[t] - floor(prefix[t]))) - 1        (the part of alpha[t] carried to the NEXT token)
          psh[t] = fp32 running sum of alpha[t] * H[t]                 (ONNX CumSum: sequential)
          E[l] = ((psh[f_l] - psh[f_{l-1}]) + remain[f_{l-1}] * H[f_{l-1}]) - remain[f_l] * H[f_l]
        Fire decisions and embeddings round differently from the sequential definition (`cif_fire`)."""
        Hn = np.asarray(H, dtype=F32)
        an = np.asarray(alphas, dtype=F32)
        B, T1 = an.shape
        D = Hn.shape[2]
        Hn = np.concatenate([Hn, np.zeros((B, T1 - Hn.shape[1], D), F32)], axis=1)
        prefix = np.cumsum(an.astype(np.float64), axis=1).astype(F32)
        pfl = np.floor(prefix)
        prev = np.concatenate([np.zeros((B, 1), F32), pfl[:, :-1]], axis=1)
        fire = (pfl - prev) > 0
        fires = (fire.astype(F32) + (prefix - pfl).astype(F32)).astype(F32)
        remain = (fires - np.floor(fires)).astype(F32)
        counts = fire.sum(axis=1).astype(np.int32)
        L = int(counts.max()) if B else 0
        E = np.zeros((B, L, D), F32)
        for b in range(B):
            psh = np.zeros(D, F32)
            last_psh = np.zeros(D, F32)
            last_rem = np.zeros(D, F32)
            l = 0
            for t in range(T1):
                psh = (psh + (an[b, t] * Hn[b, t]).astype(F32)).astype(F32)
                if fire[b, t]:
                    rem_h = (remain[b, t] * Hn[b, t]).astype(F32)
                    E[b, l] = (((psh - last_psh).astype(F32) + last_rem).astype(F32) - rem_h).astype(F32)
                    last_psh = psh.copy()
                    last_rem = rem_h
                    l += 1
        tnum = []
        for b in range(B):                                # token_num: the sequential fp32 sum, as in cif_fire
            sacc = F32(0.0)
            for t in range(T1):
                sacc = F32(sacc + an[b, t])
            tnum.append(int(np.floor(sacc)))
        return E, counts, np.asarray(tnum, np.int32)

    # -- BiCIF timestamp head ----------------------------------------------
    def us_alphas_peak(self, H, token_num):
        """CifPredictorV3.get_upsample_timestmap (FunASR export; external to /root/reference, consumed by
        AliParaformerAsr/OfflineRecognizer.cs:172-183 as output [3] `us_cif_peak`):
          ConvTranspose1d(D, D, k=3, stride=3) -> BiLSTM(D) -> Linear(2D, 1) -> sigmoid ->
          relu(a * smooth2 - noise2) (mask all ones) -> renormalise each row to token_num ->
          cif_wo_hidden(alphas, threshold - 1e-4): running integrate, recorded BEFORE the reset.
        H [B,T,D] torch, token_num [B] ints.  Returns us_alphas, us_cif_peak  [B, 3T] float32."""
        c = self.cfg
        q = self.q
        B, T, D = H.shape
        up = c.upsample
        Wt = q(self.w["predictor.upsample.weight"])                 # [in, out, k]
        # out[b, 3t+j, o] = sum_c H[b,t,c] * W[c,o,j] + bias[o]
        y = torch.einsum("btc,coj->btjo", q(H), Wt).reshape(B, T * up, D) + self.w["predictor.upsample.bias"]
        yq = q(y)                                                   # engine keeps it as a 16-bit GEMM operand
        outs = []
        for sfx, rev in (("", False), ("_reverse", True)):
            Wih = q(self.w["predictor.blstm.weight_ih" + sfx]); Whh = q(self.w["predictor.blstm.weight_hh" + sfx])
            bias = self.w["predictor.blstm.bias_ih" + sfx] + self.w["predictor.blstm.bias_hh" + sfx]
            xg = torch.matmul(yq, Wih.t()) + bias                  # [B, 3T, 4D]
            h = torch.zeros(B, D); cst = torch.zeros(B, D)
            hs = [None] * (T * up)
            order = range(T * up - 1, -1, -1) if rev else range(T * up)
            for t in order:
                g = xg[:, t] + torch.matmul(q(h), Whh.t())
                i_, f_, g_, o_ = torch.split(g, D, dim=-1)
                cst = torch.sigmoid(f_) * cst + torch.sigmoid(i_) * torch.tanh(g_)
                h = torch.sigmoid(o_) * torch.tanh(cst)
                hs[t] = h
            outs.append(torch.stack(hs, dim=1))
        hcat = torch.cat(outs, dim=-1)                              # [B, 3T, 2D]
        z = torch.matmul(hcat, self.w["predictor.out2.weight"].t()).squeeze(-1) + self.w["predictor.out2.bias"]
        a2 = torch.relu(torch.sigmoid(z) * c.cif_smooth2 - c.cif_noise2)
        a2 = a2.numpy().astype(F32)
        tn = np.asarray(token_num, dtype=F32)
        ssum = a2.sum(axis=1, dtype=F32)
        a2 = (a2 * (tn / ssum)[:, None].astype(F32)).astype(F32)
        thr = F32(F32(c.cif_threshold) - F32(1e-4))
        peak = np.zeros_like(a2)
        for b in range(B):
            integ = F32(0.0)
            for t in range(a2.shape[1]):
                integ = F32(integ + a2[b, t])
                peak[b, t] = integ
                if integ >= thr:
                    integ = F32(integ - thr)
        return a2, peak

    # -- decoder -----------------------------------------------------------
    def ffn_dec(self, x, p):
        h = torch.relu(self.lin(x, p + ".ffn.w1"))
        if not self.int8:
            h = self.q(h)                                     # 16-bit modes: the engine stores this hidden as f16 (fp32 in int8 mode)
        h = self.ln(h, p + ".ffn.norm")
        return self.lin(h, p + ".ffn.w2", bias=False)

    def _sanm_decoder(self, x, memory, token_num, prefix, n_layers, kernel):
        """ParaformerSANMDecoder body shared by the ASR decoder (memory = encoder output) and the SeACo bias
        decoder (wo_input_layer, memory = bias_embed): n_layers x [FFNdec -> FSMN(k) -> cross-attention],
        then decoders3 (FFNdec, no residual) and after_norm.  Returns the hidden [B,L,D]."""
        c = self.cfg
        q = self.q
        x = torch.as_tensor(x, dtype=torch.float32)
        memory = torch.as_tensor(memory, dtype=torch.float32)
        B, L, D = x.shape
        if L == 0:                                   # no CIF fire at all: nothing to decode
            return x
        tn = torch.as_tensor(np.asarray(token_num), dtype=torch.int64)
        mask = (torch.arange(L)[None, :] < tn[:, None]).to(torch.float32).unsqueeze(-1)  # [B,L,1]
        dk = D // c.heads
        for i in range(n_layers):
            p = f"{prefix}.layers.{i}"
            t = self.ffn_dec(self.ln(x, p + ".norm1"), p)
            tn2 = self.ln(t, p + ".norm2")
            x = x + fsmn(tn2, self.w[p + ".fsmn.weight"], kernel, mask)
            xn = self.ln(x, p + ".norm3")
            qq = self.lin(xn, p + ".src.q")
            kv = self.lin(memory, p + ".src.kv")
            k, v = torch.split(kv, D, dim=-1)
            ctx = self.mha(q(qq * (dk ** -0.5)), q(k), q(v))
            if self.int8:
                ctx = q(ctx)
            x = x + self.lin(ctx, p + ".src.out")
        x = self.ffn_dec(self.ln(x, prefix + ".final.norm1"), prefix + ".final")
        return self.ln(x, prefix + ".after_norm")

    def decoder(self, E, H, token_num, return_hidden=False):
        """E [B,L,512] acoustic embeds, H [B,T,512] memory, token_num [B] -> logits [B,L,V]
        (log_softmax applied, as the ONNX graph does)."""
        hid = self._sanm_decoder(E, H, token_num, "decoder", self.cfg.dec_layers, self.cfg.kernel)
        logp = log_softmax(self.lin(hid, "decoder.output"))
        return (logp, hid) if return_hidden else logp

    # -- SeACo ---------------------------------------------------------------
    def seaco_embed(self, hotwords):
        """model_eb (ContextualEmbedderExport of the FunASR export, external to /root/reference; call site
        AliParaformerAsr/EmbedSeacoModel.cs:70-123): hotword int [N,10] -> Embedding -> LSTM (time-major,
        no length masking: pad id 0 is embedded like any token) -> ALL 10 outputs, hw_embed [10, N, D]."""
        c = self.cfg
        q = self.q
        ids = torch.as_tensor(np.asarray(hotwords, dtype=np.int64))
        x = self.w["seaco.embed.weight"][ids]                      # [N,10,D]
        N, J, D = x.shape
        for l in range(c.seaco_lstm_layers):
            Wih = q(self.w["seaco.lstm.l%d.weight_ih" % l]); Whh = q(self.w["seaco.lstm.l%d.weight_hh" % l])
            bias = self.w["seaco.lstm.l%d.bias_ih" % l] + self.w["seaco.lstm.l%d.bias_hh" % l]
            xg = torch.matmul(q(x), Wih.t()) + bias
            h = torch.zeros(N, D); cst = torch.zeros(N, D)
            outs = []
            for t in range(J):
                g = xg[:, t] + torch.matmul(q(h), Whh.t())
                i_, f_, g_, o_ = torch.split(g, D, dim=-1)
                cst = torch.sigmoid(f_) * cst + torch.sigmoid(i_) * torch.tanh(g_)
                h = torch.sigmoid(o_) * torch.tanh(cst)
                outs.append(h)
            x = torch.stack(outs, dim=1)                            # [N,10,D]
        return x.transpose(0, 1).contiguous()                       # [10,N,D]

    def seaco(self, speech, hotwords):
        """SeACo-paraformer graph (export_forward of the FunASR SeACo export; reference call site
        AliParaformerAsr/OfflineProjOfSeacoParaformer.cs:48-135).  hotwords: int [N,10] (PadList output).
        ASR branch as paraformer(); bias branch: the bias decoder is run on the CIF embeds AND on the ASR
        decoder hidden, both attending bias_embed; merged -> hotword_output_layer -> log_softmax (dha);
        where argmax(dha) == NO_BIAS the ASR log-probs are kept, elsewhere the dha log-probs replace them
        (seaco_weight = 1)."""
        c = self.cfg
        H = self.encoder(speech)
        a = self.cif_alphas(H)
        E, counts, tnum = self.cif(H.numpy(), a.numpy())
        logp, hid = self.decoder(E, H, tnum, return_hidden=True)
        out = {"token_num": tnum, "fire_count": counts, "asr_logits": logp.numpy(), "alphas": a.numpy()}
        hw = np.asarray(hotwords)
        if hw.shape[0] > 0:
            B = H.shape[0]
            hw_embed = self.seaco_embed(hw)                         # [10,N,D]
            J, N, D = hw_embed.shape
            bias = hw_embed.transpose(0, 1).reshape(1, N * J, D).expand(B, N * J, D)   # row n*10+j (EmbedSeacoModel / :83-111)
            cif_att = self._sanm_decoder(E, bias, tnum, "seaco.decoder", c.seaco_layers, c.seaco_kernel)
            dec_att = self._sanm_decoder(hid, bias, tnum, "seaco.decoder", c.seaco_layers, c.seaco_kernel)
            dha = log_softmax(self.lin(cif_att + dec_att, "seaco.output"))
            nobias = (torch.argmax(dha, dim=-1) == c.seaco_nobias).unsqueeze(-1)
            logp = torch.where(nobias, logp, dha)
            out["dha_logits"] = dha.numpy()
        out["logits"] = logp.numpy()
        if c.timestamp_head:
            out["us_alphas"], out["us_cif_peak"] = self.us_alphas_peak(H, tnum)
        return out

    # -- full graphs ---------------------------------------------------------
    def paraformer(self, speech):
        """= InferenceSession.Run for paraformer-large: returns dict with
        logits [B,L,V] (log-probs), token_num [B], alphas, H, E."""
        H = self.encoder(speech)
        a = self.cif_alphas(H)
        E, counts, tnum = self.cif(H.numpy(), a.numpy())
        logits = self.decoder(E, H, tnum)
        out = {"logits": logits.numpy(), "token_num": tnum, "fire_count": counts,
               "alphas": a.numpy(), "H": H.numpy(), "E": E}
        if self.cfg.timestamp_head:
            out["us_alphas"], out["us_cif_peak"] = self.us_alphas_peak(H, tnum)
        return out

    def sensevoice(self, speech):
        """SenseVoice-small (speech already carries the 4 prompt frames):
        encoder (50 + 20 tp blocks) -> CTC linear -> log_softmax, [B,T+4,V]."""
        H = self.encoder(speech)
        logits = self.lin(H, "ctc")
        return {"logits": log_softmax(logits).numpy(), "H": H.numpy(),
                "token_num": np.full((H.shape[0],), H.shape[1], np.int32)}


def argmax_last(logits: np.ndarray) -> np.ndarray:
    """OfflineRecognizer.cs:139-152: cur = (x[cur] > x[k]) ? cur : k  for k = 1..V-1
    => ties and NaN compares resolve to the LARGER index (quirk Q4). [..., V] -> int64."""
    x = np.asarray(logits)
    V = x.shape[-1]
    flat = x.reshape(-1, V)
    out = np.zeros(flat.shape[0], dtype=np.int64)
    if np.isnan(flat).any():
        for r in range(flat.shape[0]):
            cur = 0
            for k in range(1, V):
                cur = cur if flat[r, cur] > flat[r, k] else k
            out[r] = cur
    else:
        rev = flat[:, ::-1]
        out = (V - 1 - np.argmax(rev, axis=1)).astype(np.int64)
    return out.reshape(x.shape[:-1])
