"""Python face of the drop-in API: same names, argument meaning and error behaviour as the
reference's public C# classes

    OfflineRecognizer  AliParaformerAsr/OfflineRecognizer.cs:13-477
    OfflineStream      AliParaformerAsr/OfflineStream.cs:7-121
    OfflineRecognizerResultEntity  AliParaformerAsr/Model/OfflineRecognizerResultEntity.cs:9-29

All logic lives in libparaformer_hip.so (C++ host mirror + HIP engine); this file is the
ctypes stub a Python caller uses, exactly as INTEGRATION.md's P/Invoke stub is for C#.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _native as N


class ObjectDisposedException(RuntimeError):
    def __init__(self, object_name: str):
        super().__init__(f"Cannot access a disposed object.\nObject name: '{object_name}'.")
        self.ObjectName = object_name


class ArgumentNullException(ValueError):
    def __init__(self, param_name: str):
        super().__init__(f"Value cannot be null. (Parameter '{param_name}')")
        self.ParamName = param_name


class RecognizerException(Exception):
    """`throw new Exception(message, inner)` of the reference."""


def _raise(code: int):
    msg = N.load().pf_last_error()
    msg = msg.decode("utf-8", "replace") if msg else ""
    if code == N.PF_ERR_TOKENS:
        raise RecognizerException("tokens invalid")
    if code == N.PF_ERR_DISPOSED:
        raise ObjectDisposedException(msg or "OfflineRecognizer")
    if code == N.PF_ERR_NULL_SAMPLES:
        raise ArgumentNullException(msg or "source")
    if code == N.PF_ERR_RECOGNITION:
        raise RecognizerException("Offline recognition failed" if not msg.startswith("Offline recognition failed")
                                  else msg)
    raise N.PfError(code, msg)


def _ck(code: int) -> int:
    if code < 0:
        _raise(code)
    return code


@dataclass
class OfflineRecognizerResultEntity:
    Text: Optional[str] = None
    TextLen: int = 0
    Tokens: List[str] = field(default_factory=list)
    Timestamps: List[List[int]] = field(default_factory=list)


@dataclass
class FrontendConfEntity:
    """Model/FrontendConfEntity.cs:6-28 (defaults included)."""
    fs: int = 16000
    window: str = "hamming"
    n_mels: int = 80
    frame_length: int = 25
    frame_shift: int = 10
    dither: float = 1.0
    lfr_m: int = 7
    lfr_n: int = 6
    snip_edges: bool = False


@dataclass
class ConfEntity:
    """Model/ConfEntity.cs: only what OfflineStream's constructor reads (frontend_conf)."""
    frontend_conf: FrontendConfEntity = field(default_factory=FrontendConfEntity)


@dataclass
class OfflineInputEntity:
    """Model/OfflineInputEntity.cs:6-11."""
    Speech: Optional[np.ndarray] = None
    SpeechLength: int = 0
    Hotwords: Optional[List[List[int]]] = field(default_factory=list)


class OfflineStream:
    Hyp: List[int] = [0, 0]                 # OfflineStream.cs:24,31 (per instance below)

    def __init__(self, mvnFilePath=None, confEntity=None, *, _lib=None, _handle=None, _recognizer=None):
        """Public form: OfflineStream(mvnFilePath, confEntity) (OfflineStream.cs:20-28) — a stream that belongs to no
        recognizer yet; the first GetResults that receives it adopts it.  CreateOfflineStream uses the private form."""
        self.Hyp = [0, 0]                   # OfflineStream.cs:24,31: never read again by the reference
        if _handle is not None:
            self._lib, self._h, self._recognizer = _lib, _handle, _recognizer
            return
        self._lib = N.load()
        self._recognizer = None
        f = (confEntity or ConfEntity()).frontend_conf
        h = C.c_void_p()
        _ck(self._lib.pf_stream_create((mvnFilePath or "").encode("utf-8"), f.fs, f.n_mels, f.lfr_m, f.lfr_n,
                                       1 if f.snip_edges else 0, float(f.dither), (f.window or "").encode("utf-8"),
                                       C.byref(h)))
        self._h = h

    def AddSamples(self, samples) -> None:
        if samples is None:
            _ck(self._lib.pf_stream_add_samples(self._h, None, 0))
            return
        x = np.ascontiguousarray(samples, dtype=np.float32)
        _ck(self._lib.pf_stream_add_samples(self._h, x.ctypes.data_as(C.POINTER(C.c_float)), x.shape[0]))

    @property
    def Hotwords(self) -> Optional[List[List[int]]]:
        n = C.c_int32()
        ids = (C.c_int32 * 4096)()
        lens = (C.c_int32 * 1024)()
        _ck(self._lib.pf_stream_get_hotwords(self._h, ids, 4096, lens, 1024, n))
        if n.value < 0:
            return None
        out, off = [], 0
        for i in range(n.value):
            out.append(list(ids[off: off + lens[i]]))
            off += lens[i]
        return out

    @Hotwords.setter
    def Hotwords(self, value: Optional[List[List[int]]]) -> None:
        if value is None:
            _ck(self._lib.pf_stream_set_hotwords(self._h, None, None, -1))
            return
        flat = [v for hw in value for v in hw]
        ids = (C.c_int32 * max(len(flat), 1))(*flat)
        lens = (C.c_int32 * max(len(value), 1))(*[len(hw) for hw in value])
        _ck(self._lib.pf_stream_set_hotwords(self._h, ids, lens, len(value)))

    @property
    def Tokens(self) -> List[int]:
        p = C.POINTER(C.c_int64)()
        n = C.c_int32()
        _ck(self._lib.pf_stream_tokens(self._h, C.byref(p), n))
        return [p[i] for i in range(n.value)]

    @Tokens.setter
    def Tokens(self, value: List[int]) -> None:
        a = (C.c_int64 * max(len(value), 1))(*value)
        _ck(self._lib.pf_stream_set_tokens(self._h, a, len(value)))

    @property
    def Timestamps(self) -> List[List[int]]:
        n = C.c_int32()
        _ck(self._lib.pf_stream_num_timestamps(self._h, n))
        out = []
        for j in range(n.value):
            p = C.POINTER(C.c_int32)()
            k = C.c_int32()
            _ck(self._lib.pf_stream_timestamp(self._h, j, C.byref(p), k))
            out.append([p[m] for m in range(k.value)])
        return out

    @Timestamps.setter
    def Timestamps(self, value: List[List[int]]) -> None:
        flat = [v for t in value for v in t]
        ints = (C.c_int32 * max(len(flat), 1))(*flat)
        lens = (C.c_int32 * max(len(value), 1))(*[len(t) for t in value])
        _ck(self._lib.pf_stream_set_timestamps(self._h, ints, lens, len(value)))

    @property
    def OfflineInputEntity(self) -> OfflineInputEntity:
        n = C.c_int32()
        rc = self._lib.pf_stream_get_speech(self._h, None, 0, n)
        speech = None
        if rc == N.PF_ERR_CAPACITY:
            speech = np.empty(n.value, np.float32)
            _ck(self._lib.pf_stream_get_speech(self._h, speech.ctypes.data_as(C.POINTER(C.c_float)), speech.size, n))
        else:
            _ck(rc)
            speech = None if n.value < 0 else np.empty(0, np.float32)
        return OfflineInputEntity(Speech=speech, SpeechLength=self.SpeechLength, Hotwords=self.Hotwords)

    @OfflineInputEntity.setter
    def OfflineInputEntity(self, value: OfflineInputEntity) -> None:
        if value is None or value.Speech is None:
            _ck(self._lib.pf_stream_set_speech(self._h, None, -1, 0 if value is None else value.SpeechLength))
        else:
            x = np.ascontiguousarray(value.Speech, dtype=np.float32)
            _ck(self._lib.pf_stream_set_speech(self._h, x.ctypes.data_as(C.POINTER(C.c_float)), x.size, value.SpeechLength))
        self.Hotwords = None if value is None else value.Hotwords

    def GetDecodeChunk(self) -> OfflineInputEntity:        # OfflineStream.cs:58-68
        return self.OfflineInputEntity

    def RemoveChunk(self) -> None:                         # OfflineStream.cs:69-79
        if len(self.Tokens) > 2:
            _ck(self._lib.pf_stream_set_speech(self._h, None, -1, 0))

    @property
    def SpeechLength(self) -> int:
        n = C.c_int32()
        _ck(self._lib.pf_stream_num_feature_floats(self._h, n))
        return n.value

    def Dispose(self) -> None:
        # the native handle stays valid: later calls raise the reference's ObjectDisposedException
        # (PF_ERR_DISPOSED "OfflineStream"), not a null-handle error
        if self._h:
            self._lib.pf_stream_dispose(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pf_stream_free(self._h)
                self._h = None
        except Exception:
            pass


class OfflineRecognizer:
    def __init__(self, modelFilePath: str, configFilePath: str, mvnFilePath: str, tokensFilePath: str,
                 modelebFilePath: str = "", hotwordFilePath: str = "", batchSize: int = 1, threadsNum: int = 1,
                 device: int = 0):
        self._lib = N.load()
        h = C.c_void_p()
        enc = lambda s: (s or "").encode("utf-8")
        _ck(self._lib.pf_recognizer_create(enc(modelFilePath), enc(configFilePath), enc(mvnFilePath),
                                           enc(tokensFilePath), enc(modelebFilePath), enc(hotwordFilePath),
                                           batchSize, threadsNum, device, C.byref(h)))
        self._h = h

    def CreateOfflineStream(self) -> OfflineStream:
        s = C.c_void_p()
        _ck(self._lib.pf_recognizer_create_stream(self._h, C.byref(s)))
        return OfflineStream(_lib=self._lib, _handle=s, _recognizer=self)

    def GetResult(self, stream: OfflineStream) -> OfflineRecognizerResultEntity:
        return self.GetResults([stream])[0]

    def GetResults(self, streams: List[OfflineStream]) -> List[OfflineRecognizerResultEntity]:
        n = len(streams)
        arr = (C.c_void_p * max(n, 1))(*[s._h for s in streams])
        _ck(self._lib.pf_recognizer_get_results(self._h, arr, n))
        out = []
        for i in range(n):
            txt = C.c_char_p()
            tl = C.c_int32()
            _ck(self._lib.pf_result_text(self._h, i, C.byref(txt), tl))
            r = OfflineRecognizerResultEntity(Text=(txt.value or b"").decode("utf-8"), TextLen=tl.value)
            nt = C.c_int32()
            _ck(self._lib.pf_result_num_tokens(self._h, i, nt))
            for j in range(nt.value):
                t = C.c_char_p()
                _ck(self._lib.pf_result_token(self._h, i, j, C.byref(t)))
                r.Tokens.append((t.value or b"").decode("utf-8"))
            nts = C.c_int32()
            _ck(self._lib.pf_result_num_timestamps(self._h, i, nts))
            for j in range(nts.value):
                p = C.POINTER(C.c_int32)()
                k = C.c_int32()
                _ck(self._lib.pf_result_timestamp(self._h, i, j, C.byref(p), k))
                r.Timestamps.append([p[m] for m in range(k.value)])
            out.append(r)
        return out

    def DisposeOfflineStream(self, offlineStream: Optional[OfflineStream]) -> None:
        if offlineStream is not None:
            offlineStream.Dispose()

    def Dispose(self) -> None:
        if self._h:
            self._lib.pf_recognizer_dispose(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pf_recognizer_free(self._h)
                self._h = None
        except Exception:
            pass
