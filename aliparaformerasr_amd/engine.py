"""numpy-facing wrapper of the pf_engine C ABI (device engine + stand-alone device ops)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class BatchResult:
    def __init__(self, token_ids, token_num, L, V, logits=None, cif_peak=None):
        self.token_ids = token_ids      # [B, L] int64
        self.token_num = token_num      # [B] int32
        self.L = L
        self.V = V
        self.logits = logits            # [B, L, V] float32 log-probs or None
        self.cif_peak = cif_peak        # [B, 3*Tmax] float32 us_cif_peak (timestamp models) or None


def _build_config(weights, weights_path, weights_device_ptr, weights_bytes, cmvn, mvn_path, device, dither, snip_edges,
                  lfr_m, lfr_n, n_mels, fs, window, use_itn, frame_length_ms, frame_shift_ms, dither_seed, math_mode):
    """pf_engine_config + the Python objects whose memory it points into."""
    cfg = N.PfEngineConfig()
    cfg.struct_size = C.sizeof(N.PfEngineConfig)
    cfg.device = device
    keep = []
    if weights_path is not None:
        cfg.weights_path = weights_path.encode()
    elif weights_device_ptr is not None:
        cfg.weights_device = C.c_void_p(weights_device_ptr)
        cfg.weights_bytes = weights_bytes
    elif weights is not None:
        buf = (C.c_char * len(weights)).from_buffer_copy(weights) if not isinstance(weights, np.ndarray) else None
        if buf is None:
            arr = np.ascontiguousarray(weights, dtype=np.uint8)
            keep.append(arr)
            cfg.weights_host = arr.ctypes.data_as(C.c_void_p)
            cfg.weights_bytes = arr.nbytes
        else:
            keep.append(buf)
            cfg.weights_host = C.cast(buf, C.c_void_p)
            cfg.weights_bytes = len(weights)
    if mvn_path is not None:
        cfg.mvn_path = mvn_path.encode()
    elif cmvn is not None:
        sh, sc = _f32(cmvn[0]), _f32(cmvn[1])
        keep += [sh, sc]
        cfg.cmvn_shift, cfg.cmvn_scale, cfg.cmvn_dim = _fp(sh), _fp(sc), sh.shape[0]
    cfg.fs, cfg.n_mels, cfg.lfr_m, cfg.lfr_n = fs, n_mels, lfr_m, lfr_n
    cfg.snip_edges = 1 if snip_edges else 0
    cfg.dither = dither
    cfg.window = window.encode()
    cfg.use_itn = 1 if use_itn else 0
    cfg.frame_length_ms, cfg.frame_shift_ms = frame_length_ms, frame_shift_ms
    cfg.dither_seed, cfg.math_mode = dither_seed, math_mode
    return cfg, keep


def _collect_result(lib, fetch_fn, call, B, want_logits):
    """The learn-L-then-fetch protocol shared by pf_engine and pf_group handles."""
    out = N.PfBatchOut()
    out.struct_size = C.sizeof(N.PfBatchOut)
    N.check(call(out))                       # first pass: learn L, V (no buffers)
    L, V, P = out.L, out.V, out.cif_peak_len
    peak = None
    if P > 0:
        peak = np.zeros((B, P), np.float32)
        out.cif_peak = _fp(peak)
        out.cif_peak_cap = peak.size
    ids = np.zeros((B, max(L, 1)), np.int64)
    tn = np.zeros(B, np.int32)
    out.token_ids = ids.ctypes.data_as(C.POINTER(C.c_int64))
    out.token_num = tn.ctypes.data_as(C.POINTER(C.c_int32))
    out.l_cap = max(L, 1)
    logits = None
    if want_logits:
        logits = np.zeros((B, L, V), np.float32)
        out.logits = _fp(logits)
        out.logits_cap = logits.size
    N.check(fetch_fn(C.byref(out)))
    return BatchResult(ids[:, :L].copy(), tn, L, V, logits, peak)


class Engine:
    """Device engine = OfflineModel + WavFrontend replacement (see include/paraformer_hip.h)."""

    def __init__(self, weights=None, weights_path=None, weights_device_ptr=None, weights_bytes=0,
                 cmvn=None, mvn_path=None, device=0, dither=0.0, snip_edges=False, lfr_m=7, lfr_n=6,
                 n_mels=80, fs=16000, window="hamming", use_itn=False, frame_length_ms=0, frame_shift_ms=0,
                 dither_seed=0, math_mode=0):
        self._lib = N.load()
        cfg, self._keep = _build_config(weights, weights_path, weights_device_ptr, weights_bytes, cmvn, mvn_path, device,
                                        dither, snip_edges, lfr_m, lfr_n, n_mels, fs, window, use_itn, frame_length_ms,
                                        frame_shift_ms, dither_seed, math_mode)
        h = C.c_void_p()
        N.check(self._lib.pf_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        kind, vocab, feat, ts = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        N.check(self._lib.pf_engine_info(self._h, kind, vocab, feat, ts))
        self.kind, self.vocab, self.feat_dim = kind.value, vocab.value, feat.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pf_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- front-end ----------------------------------------------------------
    def num_frames(self, n_samples: int) -> int:
        t = C.c_int32()
        N.check(self._lib.pf_frontend_num_frames(self._h, n_samples, t))
        return t.value

    def fbank(self, samples) -> np.ndarray:
        x = _f32(samples)
        cap = (x.shape[0] // 160 + 2) * 128
        out = np.zeros(cap, np.float32)
        t = C.c_int32()
        N.check(self._lib.pf_fbank(self._h, _fp(x), x.shape[0], _fp(out), cap, t))
        return out[: t.value * 80].reshape(t.value, 80).copy()

    def frontend(self, samples) -> np.ndarray:
        x = _f32(samples)
        t = self.num_frames(x.shape[0])
        w = self.feat_dim
        out = np.zeros(max(t, 1) * w, np.float32)
        tt = C.c_int32()
        N.check(self._lib.pf_frontend(self._h, _fp(x), x.shape[0], _fp(out), out.shape[0], tt))
        return out[: tt.value * w].reshape(tt.value, w).copy()

    # ---- forward ------------------------------------------------------------
    def _collect(self, call, B, want_logits):
        return _collect_result(self._lib, lambda o: self._lib.pf_fetch(self._h, o), call, B, want_logits)

    @staticmethod
    def _hw(hotwords):
        """SeACo hotword ids [N,10] int32 (PadList output) -> (pointer, N); None -> (NULL, 0)."""
        if hotwords is None:
            return None, 0, None
        a = np.ascontiguousarray(hotwords, dtype=np.int32).reshape(-1, 10)
        return a.ctypes.data_as(C.POINTER(C.c_int32)), a.shape[0], a

    def forward_feats(self, speech, want_logits=False, hotwords=None) -> BatchResult:
        sp = _f32(speech)
        B, T, _ = sp.shape
        hp, hn, _keep = self._hw(hotwords)
        if want_logits:
            # logits must be requested at forward time: pass a 1-float dummy capacity marker
            dummy = np.zeros(1, np.float32)

            def call(out):
                out.logits = _fp(dummy)
                out.logits_cap = 1
                rc = self._lib.pf_forward_feats(self._h, _fp(sp), B, T, hp, hn, C.byref(out))
                # capacity error on the dummy buffer is expected; L and V are filled in
                return 0 if rc == N.PF_ERR_CAPACITY else rc
        else:
            def call(out):
                return self._lib.pf_forward_feats(self._h, _fp(sp), B, T, hp, hn, C.byref(out))
        return self._collect(call, B, want_logits)

    def model_proj(self, speeches, want_logits=False, hotwords=None) -> BatchResult:
        hp, hn, _keep = self._hw(hotwords)
        arrs = [_f32(s).reshape(-1) for s in speeches]
        B = len(arrs)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fp(a) for a in arrs])
        lens = (C.c_int32 * B)(*[a.shape[0] for a in arrs])
        dummy = np.zeros(1, np.float32)

        def call(out):
            if want_logits:
                out.logits = _fp(dummy)
                out.logits_cap = 1
            rc = self._lib.pf_model_proj(self._h, ptrs, lens, B, hp, hn, C.byref(out))
            return 0 if (want_logits and rc == N.PF_ERR_CAPACITY) else rc
        return self._collect(call, B, want_logits)

    def recognize(self, samples_list, want_logits=False, hotwords=None) -> BatchResult:
        hp, hn, _keep = self._hw(hotwords)
        arrs = [_f32(s) for s in samples_list]
        B = len(arrs)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fp(a) for a in arrs])
        ns = (C.c_int64 * B)(*[a.shape[0] for a in arrs])
        dummy = np.zeros(1, np.float32)

        def call(out):
            if want_logits:
                out.logits = _fp(dummy)
                out.logits_cap = 1
            rc = self._lib.pf_recognize(self._h, ptrs, ns, B, hp, hn, C.byref(out))
            return 0 if (want_logits and rc == N.PF_ERR_CAPACITY) else rc
        return self._collect(call, B, want_logits)

    # split form (bench): audio resident in HBM before the timed region
    def stage_audio(self, samples_list):
        arrs = [_f32(s) for s in samples_list]
        B = len(arrs)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fp(a) for a in arrs])
        ns = (C.c_int64 * B)(*[a.shape[0] for a in arrs])
        N.check(self._lib.pf_stage_audio(self._h, ptrs, ns, B))
        self._staged_B = B

    def set_hotwords(self, hotwords):
        """SeACo hotword ids [N,10] for the following run_staged() calls."""
        hp, hn, _keep = self._hw(hotwords)
        N.check(self._lib.pf_engine_set_hotwords(self._h, hp, hn))

    def run_staged(self):
        N.check(self._lib.pf_run_staged(self._h))

    def sync(self):
        N.check(self._lib.pf_sync(self._h))

    def fetch(self) -> BatchResult:
        return self._collect(lambda out: self._lib.pf_fetch(self._h, C.byref(out)), self._staged_B, False)

    def fetch_ids_device(self, dev_ptr: int, l_cap: int) -> int:
        """The staged result's ids [B, l_cap] int64 (-1 padded) into caller-owned DEVICE memory (e.g. a torch tensor's
        data_ptr on this engine's GPU) — what a multi-GPU caller hands to the RCCL all-gather.  Returns L."""
        L = C.c_int32(0)
        N.check(self._lib.pf_fetch_ids_device(self._h, C.c_void_p(dev_ptr), l_cap, C.byref(L)))
        return L.value

    def profile(self, on: bool):
        N.check(self._lib.pf_profile_enable(self._h, 1 if on else 0))

    def profile_select(self, cls: str = ""):
        N.check(self._lib.pf_profile_select(self._h, cls.encode()))

    def profile_reset(self):
        N.check(self._lib.pf_profile_reset(self._h))

    def profile_get(self, cls: str):
        ms, n, fpl = C.c_double(), C.c_int64(), C.c_double()
        N.check(self._lib.pf_profile_get(self._h, cls.encode(), ms, n, fpl))
        return ms.value, n.value, fpl.value

    def profile_kernel(self, cls: str) -> str:
        buf = C.create_string_buffer(256)
        N.check(self._lib.pf_profile_kernel(self._h, cls.encode(), buf, 256))
        return buf.value.decode()

    def last_flops(self) -> float:
        f = C.c_double()
        N.check(self._lib.pf_last_flops(self._h, f))
        return f.value

    # ---- stand-alone ops ------------------------------------------------------
    def op_lfr_cmvn_pad(self, fbanks, sentinel=True) -> np.ndarray:
        arrs = [_f32(f).reshape(-1, 80) for f in fbanks]
        B = len(arrs)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fp(a) for a in arrs])
        t80 = (C.c_int32 * B)(*[a.shape[0] for a in arrs])
        tmax = max([a.shape[0] // 6 for a in arrs] + [0])
        out = np.zeros((B, tmax, self.feat_dim), np.float32)
        tm = C.c_int32()
        N.check(self._lib.pf_op_lfr_cmvn_pad(self._h, ptrs, t80, B, 1 if sentinel else 0, _fp(out), out.size, tm))
        return out

    def op_qlinear(self, x, W, bias=None, relu=False, x_is_f16=False, details=False, f16_result=False):
        """One dynamically quantised Linear on the int8 MFMA (pf_op_qlinear).  details=True also returns the uint8
        activations, (x_scale, x_zp), the uint8 weights, w_scale and w_zp.  f16_result: through the f16-result kernel
        (QKV / FFN-up in the pipeline): y = float32 of the f16 values it stored; the call fails if the range the kernel's
        epilogue reports for the next quantiser differs from a min / max pass over its output."""
        x, W = _f32(x), _f32(W)
        rows, depth = x.shape
        cols = W.shape[0]
        y = np.zeros((rows, cols), np.float32)
        b = _f32(bias) if bias is not None else None
        xq = np.zeros((rows, depth), np.uint8)
        wq = np.zeros((cols, depth), np.uint8)
        ap = np.zeros(2, np.float32)
        ws = np.zeros(cols, np.float32)
        wz = np.zeros(cols, np.int32)
        u8 = C.POINTER(C.c_uint8)
        N.check(self._lib.pf_op_qlinear(self._h, _fp(x), _fp(W), _fp(b) if b is not None else None, rows, cols, depth,
                                        1 if relu else 0, (1 if x_is_f16 else 0) | (2 if f16_result else 0), _fp(y),
                                        xq.ctypes.data_as(u8), _fp(ap),
                                        wq.ctypes.data_as(u8), _fp(ws), wz.ctypes.data_as(C.POINTER(C.c_int32))))
        return (y, xq, (float(ap[0]), int(ap[1])), wq, ws, wz) if details else y

    def op_argmax(self, x) -> np.ndarray:
        a = _f32(x)
        V = a.shape[-1]
        rows = a.size // V if V else 0
        ids = np.zeros(rows, np.int64)
        N.check(self._lib.pf_op_argmax(self._h, _fp(a), rows, V, ids.ctypes.data_as(C.POINTER(C.c_int64))))
        return ids.reshape(a.shape[:-1])

    def op_gemm(self, A, W, bias=None, relu=False, f16_out=False) -> np.ndarray:
        A, W = _f32(A), _f32(W)
        M, K = A.shape
        Nn = W.shape[0]
        out = np.zeros((M, Nn), np.float32)
        b = _f32(bias) if bias is not None else None
        N.check(self._lib.pf_op_gemm(self._h, _fp(A), _fp(W), _fp(b) if b is not None else None, M, Nn, K,
                                     2 if f16_out else (1 if relu else 0), _fp(out)))
        return out

    def op_gemm_ex(self, A, W, bias=None, resid=None, add2=None, relu=False, out_kind=0, a_blocked=False,
                   tile_rows=0, scale_cols=0, scale=1.0) -> np.ndarray:
        """The GEMM as the pipeline launches it (out_kind 0 fp32 / 1 f16 / 2 f16 blocked)."""
        A, W = _f32(A), _f32(W)
        M, K = A.shape
        Nn = W.shape[0]
        d = N.PfGemmDesc()
        d.struct_size = C.sizeof(N.PfGemmDesc)
        d.M, d.N, d.K = M, Nn, K
        d.relu, d.out_kind, d.a_blocked, d.tile_rows = int(relu), out_kind, int(a_blocked), tile_rows
        d.scale_cols, d.scale = scale_cols, scale
        keep = [_f32(t) if t is not None else None for t in (bias, resid, add2)]
        d.bias, d.resid, d.add2 = [_fp(t) if t is not None else None for t in keep]
        out = np.zeros((M, Nn), np.float32)
        N.check(self._lib.pf_op_gemm_ex(self._h, C.byref(d), _fp(A), _fp(W), _fp(out)))
        return out

    def op_gemm_rc(self, A, W, bias=None, resid=None, fsmn_v=None, fsmn_w=None, T=0, ln=None, a_blocked=False,
                   want_x=True, want_n16=True, want_n32=True, short_input=False):
        """Row-complete GEMM (N = 512) + fused epilogue; returns (x, n16, n32) (n* = None without `ln`).
        short_input: the short-input kernels of the same graph nodes (k_gemm_small.hip)."""
        A, W = _f32(A), _f32(W)
        M, K = A.shape
        d = N.PfGemmRcDesc()
        d.struct_size = C.sizeof(N.PfGemmRcDesc)
        d.M, d.K, d.a_blocked, d.T = M, K, int(a_blocked), T
        d.short_input = int(short_input)
        d.split_k = 0
        keep = [_f32(t) if t is not None else None for t in (bias, resid, fsmn_v, fsmn_w)]
        d.bias, d.resid, d.fsmn_v, d.fsmn_w = [_fp(t) if t is not None else None for t in keep]
        d.fsmn_k = keep[3].shape[1] if keep[3] is not None else 0
        g = b = None
        if ln is not None:
            g, b = _f32(ln[0]), _f32(ln[1])
            d.ln_gamma, d.ln_beta = _fp(g), _fp(b)
        x = np.zeros((M, 512), np.float32) if (want_x or ln is None) else None
        n16 = np.zeros((M, 512), np.float32) if (ln is not None and want_n16) else None
        n32 = np.zeros((M, 512), np.float32) if (ln is not None and want_n32) else None
        N.check(self._lib.pf_op_gemm_rc(self._h, C.byref(d), _fp(A), _fp(W), _fp(x) if x is not None else None,
                                        _fp(n16) if n16 is not None else None, _fp(n32) if n32 is not None else None))
        return x, n16, n32

    def op_qkv_attention(self, x, w, bias, B, T):
        """Fused Q | K | V projection (persistent 256 x 192 kernel: Q, K blocked, V row-major) + self-attention on that
        layout, as the encoder launches them for long inputs; x [B*T, K], w [1536, K]; returns (q, k, v, ctx) [B*T, 512]."""
        x, w = _f32(x), _f32(w)
        M, K = x.shape
        assert M == B * T and w.shape == (1536, K)
        bias = _f32(bias) if bias is not None else None
        outs = [np.zeros((M, 512), np.float32) for _ in range(4)]
        N.check(self._lib.pf_op_qkv_attention(self._h, _fp(x), _fp(w), _fp(bias) if bias is not None else None, B, T, K,
                                              *[_fp(o) for o in outs]))
        return tuple(outs)

    def op_ffn(self, x, w1, b1, w2, b2, resid) -> np.ndarray:
        x, w1, b1, w2, b2, resid = map(_f32, (x, w1, b1, w2, b2, resid))
        M, D = x.shape
        F = w1.shape[0]
        y = np.zeros((M, D), np.float32)
        N.check(self._lib.pf_op_ffn(self._h, _fp(x), _fp(w1), _fp(b1), _fp(w2), _fp(b2), _fp(resid), M, D, F, _fp(y)))
        return y

    def op_ffn_fused(self, x, w1, b1, w2, b2, resid=None, ln=None):
        """The encoder FFN block as ONE launch (k_ffn.hip): returns (x_out, n16_out or None);
        ln = (gamma, beta) of the LayerNorm that follows."""
        x, w1, b1, w2, b2 = map(_f32, (x, w1, b1, w2, b2))
        M, D = x.shape
        assert D == 512 and w1.shape == (2048, 512) and w2.shape == (512, 2048)
        resid = _f32(resid) if resid is not None else None
        g = _f32(ln[0]) if ln is not None else None
        be = _f32(ln[1]) if ln is not None else None
        xo = np.zeros((M, D), np.float32)
        no = np.zeros((M, D), np.float32) if ln is not None else None
        N.check(self._lib.pf_op_ffn_fused(self._h, _fp(x), _fp(w1), _fp(b1), _fp(w2), _fp(b2),
                                          _fp(resid) if resid is not None else None, _fp(g) if g is not None else None,
                                          _fp(be) if be is not None else None, M, _fp(xo), _fp(no) if no is not None else None))
        return xo, no

    def op_dec_ffn_fused(self, x, w1, b1, ln_hidden, w2, ln=None, splits=0, out_proj=None):
        """The decoder's FFN block (LayerNorm over the 2048 hidden columns between the products, w2 without bias) in the split
        form of the fused kernel (k_ffn.hip): returns (t, LayerNorm(t; ln) or None).  x = the block's normalised input;
        ln_hidden = (gamma, beta) [2048]; splits: 0 = the pipeline's choice for the row count, or 1 | 2 | 3 | 4 | 8.
        out_proj = (ctx, wo, bo, resid, (g1, b1)): the previous layer's cross-attention out-projection in front of the block
        in the same launch (x is ignored): returns (t, n, x_out) with x_out = resid + ctx wo^T + bo, the block's input =
        LayerNorm(x_out; g1, b1)."""
        arrs = dict(w1=_f32(w1), b1=_f32(b1), gamma_f=_f32(ln_hidden[0]), beta_f=_f32(ln_hidden[1]), w2=_f32(w2))
        if out_proj is None:
            arrs["x"] = _f32(x)
            M = arrs["x"].shape[0]
        else:
            ctx, wo, bo, resid, ln1 = out_proj
            arrs.update(ctx=_f32(ctx), wo=_f32(wo), bo=_f32(bo), resid=_f32(resid), ln1_gamma=_f32(ln1[0]), ln1_beta=_f32(ln1[1]))
            M = arrs["ctx"].shape[0]
        assert arrs["w1"].shape == (2048, 512) and arrs["w2"].shape == (512, 2048) and arrs["gamma_f"].shape == (2048,)
        if ln is not None:
            arrs["ln_gamma"], arrs["ln_beta"] = _f32(ln[0]), _f32(ln[1])
        d = N.PfDecFfnDesc()
        d.struct_size, d.M, d.splits = C.sizeof(N.PfDecFfnDesc), M, int(splits)
        for k, a in arrs.items():
            setattr(d, k, _fp(a))
        t = np.zeros((M, 512), np.float32)
        n = np.zeros((M, 512), np.float32) if ln is not None else None
        xo = np.zeros((M, 512), np.float32) if out_proj is not None else None
        N.check(self._lib.pf_op_dec_ffn_fused(self._h, C.byref(d), _fp(t), _fp(n) if n is not None else None,
                                              _fp(xo) if xo is not None else None))
        return (t, n, xo) if out_proj is not None else (t, n)

    def op_attn_ffn_fused(self, ctx, wo, bo, v, fsmn_w, T, ln2, w1, b1, w2, b2, resid=None, ln=None, qkv=None):
        """Out-projection + FSMN + norm2 + the FFN block + the next LayerNorm as the ONE launch the pipeline runs
        (k_ffn.hip, OP = 1): returns (x_out, n16_out or None); with qkv = (wqkv [1536,512], bqkv) also the next layer's
        Q | K | V projection in the same launch: returns (x_out, n16_out, q, k, v)."""
        arrs = {k: _f32(a) for k, a in dict(ctx=ctx, wo=wo, bo=bo, v=v, fsmn_w=fsmn_w, ln2_gamma=ln2[0], ln2_beta=ln2[1],
                                             w1=w1, b1=b1, w2=w2, b2=b2).items()}
        if resid is not None:
            arrs["resid"] = _f32(resid)
        if ln is not None:
            arrs["ln_gamma"], arrs["ln_beta"] = _f32(ln[0]), _f32(ln[1])
        M = arrs["ctx"].shape[0]
        d = N.PfAttnFfnDesc()
        d.struct_size, d.M, d.T = C.sizeof(N.PfAttnFfnDesc), M, T
        for k, a in arrs.items():
            setattr(d, k, _fp(a))
        xo = np.zeros((M, 512), np.float32)
        no = np.zeros((M, 512), np.float32) if ln is not None else None
        outs = []
        if qkv is not None:
            keep = (_f32(qkv[0]), _f32(qkv[1]))
            d.wqkv, d.bqkv = _fp(keep[0]), _fp(keep[1])
            outs = [np.zeros((M, 512), np.float32) for _ in range(3)]
            d.q_out, d.k_out, d.v_out = [_fp(o) for o in outs]
        N.check(self._lib.pf_op_attn_ffn_fused(self._h, C.byref(d), _fp(xo), _fp(no) if no is not None else None))
        return (xo, no, *outs) if outs else (xo, no)

    def op_fsmn_enc(self, v, w) -> np.ndarray:
        v, w = _f32(v), _f32(w)
        B, T, D = v.shape
        y = np.zeros_like(v)
        N.check(self._lib.pf_op_fsmn_enc(self._h, _fp(v), _fp(w), B, T, D, w.shape[1], _fp(y)))
        return y

    def op_linear32(self, x, W, bias=None, resid=None, relu=False) -> np.ndarray:
        """A Linear of the fp32 graph as math_mode 1 / 3 runs it (pf_op_linear32)."""
        x, W = _f32(x), _f32(W)
        M, K = x.shape
        Nn = W.shape[0]
        b = _f32(bias) if bias is not None else None
        r = _f32(resid) if resid is not None else None
        y = np.zeros((M, Nn), np.float32)
        N.check(self._lib.pf_op_linear32(self._h, _fp(x), _fp(W), _fp(b) if b is not None else None,
                                         _fp(r) if r is not None else None, M, Nn, K, 1 if relu else 0, _fp(y)))
        return y

    def op_ffn32(self, x, W1, b1, W2, b2) -> np.ndarray:
        """x + relu(x W1^T + b1) W2^T + b2 as math_mode 1 / 3 runs the FFN block (pf_op_ffn32)."""
        x, W1, b1, W2, b2 = _f32(x), _f32(W1), _f32(b1), _f32(W2), _f32(b2)
        M, D = x.shape
        F = W1.shape[0]
        y = np.zeros((M, D), np.float32)
        N.check(self._lib.pf_op_ffn32(self._h, _fp(x), _fp(W1), _fp(b1), _fp(W2), _fp(b2), M, D, F, _fp(y)))
        return y

    def op_fsmn_dec(self, tn, w, token_num, x) -> np.ndarray:
        tn, w, x = _f32(tn), _f32(w), _f32(x).copy()
        B, L, D = tn.shape
        t = np.ascontiguousarray(token_num, dtype=np.int32)
        N.check(self._lib.pf_op_fsmn_dec(self._h, _fp(tn), _fp(w), t.ctypes.data_as(C.POINTER(C.c_int32)), B, L, D,
                                         w.shape[1], _fp(x)))
        return x

    def op_logsoftmax_argmax(self, x, store=True):
        a = _f32(x)
        V = a.shape[-1]
        rows = a.size // V
        ids = np.zeros(rows, np.int64)
        y = np.zeros_like(a) if store else None
        N.check(self._lib.pf_op_logsoftmax_argmax(self._h, _fp(a), rows, V, _fp(y) if store else None,
                                                  ids.ctypes.data_as(C.POINTER(C.c_int64))))
        return (y, ids.reshape(a.shape[:-1])) if store else ids.reshape(a.shape[:-1])

    # ---- streaming seams (include/paraformer_hip.h section 7) ----------------------
    def online_encoder(self, speech):
        sp = _f32(speech)
        B, Tc, _ = sp.shape
        enc = np.zeros((B, Tc, 512), np.float32)
        al = np.zeros((B, Tc), np.float32)
        N.check(self._lib.pf_online_encoder(self._h, _fp(sp), B, Tc, _fp(enc), _fp(al)))
        return enc, al

    def online_decoder(self, enc, embeds, embeds_len, caches, want_logits=True):
        enc, emb = _f32(enc), _f32(embeds)
        B, Tc, _ = enc.shape
        L = emb.shape[1]
        ln = np.ascontiguousarray(embeds_len, dtype=np.int32)
        cin = _f32(np.stack(caches))                       # [n_layers, B, 512, 10]
        cout = np.zeros_like(cin)
        ids = np.zeros((B, L), np.int64)
        logits = np.zeros((B, L, self.vocab), np.float32) if want_logits else None
        N.check(self._lib.pf_online_decoder(self._h, _fp(enc), B, Tc, _fp(emb), L, ln.ctypes.data_as(C.POINTER(C.c_int32)),
                                            _fp(cin), cin.shape[0], _fp(logits) if want_logits else None,
                                            ids.ctypes.data_as(C.POINTER(C.c_int64)), _fp(cout)))
        return logits, ids, [cout[i] for i in range(cout.shape[0])]

    def op_layernorm(self, x, gamma, beta) -> np.ndarray:
        x, g, b = _f32(x), _f32(gamma), _f32(beta)
        D = x.shape[-1]
        y = np.zeros_like(x)
        N.check(self._lib.pf_op_layernorm(self._h, _fp(x), _fp(g), _fp(b), x.size // D, D, _fp(y)))
        return y

    def op_attention(self, q, k, v, heads=4) -> np.ndarray:
        q, k, v = _f32(q), _f32(k), _f32(v)
        B, Lq, _ = q.shape
        Lk = k.shape[1]
        o = np.zeros_like(q)
        N.check(self._lib.pf_op_attention(self._h, _fp(q), _fp(k), _fp(v), B, Lq, Lk, heads, _fp(o)))
        return o

    def op_fsmn(self, v, w, mask=None) -> np.ndarray:
        v, w = _f32(v), _f32(w)
        B, T, D = v.shape
        y = np.zeros_like(v)
        m = _f32(mask) if mask is not None else None
        N.check(self._lib.pf_op_fsmn(self._h, _fp(v), _fp(w), _fp(m) if m is not None else None, B, T, D,
                                     w.shape[1], _fp(y)))
        return y

    def op_cif(self, H, alphas, threshold=1.0, Lcap=None):
        H, a = _f32(H), _f32(alphas)
        B, T, D = H.shape
        if Lcap is None:
            Lcap = int(np.ceil(a.sum(axis=1).max())) + 2
        E = np.zeros((B, Lcap, D), np.float32)
        fc, tn, L = np.zeros(B, np.int32), np.zeros(B, np.int32), C.c_int32()
        N.check(self._lib.pf_op_cif(self._h, _fp(H), _fp(a), B, T, D, threshold, Lcap, _fp(E),
                                    fc.ctypes.data_as(C.POINTER(C.c_int32)), tn.ctypes.data_as(C.POINTER(C.c_int32)), L))
        return E[:, : L.value].copy(), fc, tn

    def op_encoder(self, speech) -> np.ndarray:
        sp = _f32(speech)
        B, T, _ = sp.shape
        H = np.zeros((B, T, 512), np.float32)
        N.check(self._lib.pf_op_encoder(self._h, _fp(sp), B, T, _fp(H)))
        return H


class EngineGroup:
    """pf_group: one engine per listed device inside this process, utterance shards, RCCL weight broadcast and
    hypothesis gather (include/paraformer_hip.h section 4b).  `devices` may repeat a device."""

    def __init__(self, devices, weights=None, weights_path=None, cmvn=None, mvn_path=None, dither=0.0, snip_edges=False,
                 lfr_m=7, lfr_n=6, n_mels=80, fs=16000, window="hamming", use_itn=False, dither_seed=0, math_mode=0):
        self._lib = N.load()
        cfg, self._keep = _build_config(weights, weights_path, None, 0, cmvn, mvn_path, 0, dither, snip_edges, lfr_m,
                                        lfr_n, n_mels, fs, window, use_itn, 0, 0, dither_seed, math_mode)
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        N.check(self._lib.pf_group_create(C.byref(cfg), devs, len(devices), C.byref(h)))
        self._h = h
        n, r = C.c_int32(), C.c_int32()
        N.check(self._lib.pf_group_info(self._h, n, r))
        self.size, self.uses_rccl = n.value, bool(r.value)

    def recognize(self, samples_list, want_logits=False, hotwords=None) -> BatchResult:
        hp, hn, _keep = Engine._hw(hotwords)
        arrs = [_f32(s) for s in samples_list]
        B = len(arrs)
        ptrs = (C.POINTER(C.c_float) * max(B, 1))(*[_fp(a) for a in arrs])
        ns = (C.c_int64 * max(B, 1))(*[a.shape[0] for a in arrs])
        dummy = np.zeros(1, np.float32)

        def call(out):
            if want_logits:
                out.logits = _fp(dummy)
                out.logits_cap = 1
            rc = self._lib.pf_group_recognize(self._h, ptrs, ns, B, hp, hn, C.byref(out))
            return 0 if (want_logits and rc == N.PF_ERR_CAPACITY) else rc
        return _collect_result(self._lib, lambda o: self._lib.pf_group_fetch(self._h, o), call, B, want_logits)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pf_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
