"""ctypes mirror of the reference's streaming API (AliParaformerAsr/OnlineRecognizer.cs, OnlineStream.cs) over
libparaformer_hip.so (include/paraformer_hip.h section 7)."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np

from . import _native as N
from .offline_recognizer import _ck


class OnlineRecognizerResultEntity:
    def __init__(self, text: str):
        self.Text = text


class OnlineStream:
    def __init__(self, lib, handle, recognizer):
        self._lib, self._h, self._recognizer = lib, handle, recognizer

    def AddSamples(self, samples) -> None:
        if samples is None:
            _ck(self._lib.pf_online_stream_add_samples(self._h, None, 0))
            return
        x = np.ascontiguousarray(samples, dtype=np.float32)
        _ck(self._lib.pf_online_stream_add_samples(self._h, x.ctypes.data_as(C.POINTER(C.c_float)), x.shape[0]))

    @property
    def Tokens(self) -> List[int]:
        p, n = C.POINTER(C.c_int64)(), C.c_int32()
        _ck(self._lib.pf_online_stream_tokens(self._h, C.byref(p), n))
        return [p[i] for i in range(n.value)]

    def Dispose(self) -> None:
        if self._h:
            self._lib.pf_online_stream_dispose(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pf_online_stream_free(self._h)
                self._h = None
        except Exception:
            pass


class OnlineRecognizer:
    def __init__(self, encoderFilePath: str, decoderFilePath: str, configFilePath: str, mvnFilePath: str,
                 tokensFilePath: str, threadsNum: int = 1, device: int = 0):
        self._lib = N.load()
        h = C.c_void_p()
        enc = lambda s: (s or "").encode("utf-8")
        _ck(self._lib.pf_online_recognizer_create(enc(encoderFilePath), enc(decoderFilePath), enc(configFilePath),
                                                  enc(mvnFilePath), enc(tokensFilePath), threadsNum, device, C.byref(h)))
        self._h = h

    def CreateOnlineStream(self) -> OnlineStream:
        s = C.c_void_p()
        _ck(self._lib.pf_online_create_stream(self._h, C.byref(s)))
        return OnlineStream(self._lib, s, self)

    def GetResult(self, stream: OnlineStream) -> OnlineRecognizerResultEntity:
        return self.GetResults([stream])[0]

    def GetResults(self, streams: List[OnlineStream]) -> List[OnlineRecognizerResultEntity]:
        n = len(streams)
        arr = (C.c_void_p * max(n, 1))(*[s._h for s in streams])
        _ck(self._lib.pf_online_get_results(self._h, arr, n))
        out = []
        for i in range(n):
            t = C.c_char_p()
            _ck(self._lib.pf_online_result_text(self._h, i, C.byref(t)))
            out.append(OnlineRecognizerResultEntity((t.value or b"").decode("utf-8")))
        return out

    def engine_handle(self):
        return self._lib.pf_online_recognizer_engine(self._h)

    def Dispose(self) -> None:
        if self._h:
            self._lib.pf_online_recognizer_dispose(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pf_online_recognizer_free(self._h)
                self._h = None
        except Exception:
            pass
