"""aliparaformerasr_amd — MI355X (gfx950) native offline Paraformer / SenseVoice path.

Drop-in for the hot path behind the reference's OfflineRecognizer / OfflineStream
(manyeyes/AliParaformerAsr): front-end, SAN-M encoder, CIF predictor, parallel decoder and
greedy arg-max run as hand-written HIP kernels inside ``libparaformer_hip.so`` behind the C
ABI of ``include/paraformer_hip.h``.  Nothing here computes on the CPU: without the built
shared library (and a gfx950 device) every entry point fails loudly.
"""
from . import weights  # noqa: F401  (pure-numpy container + synthetic weights)

__all__ = ["weights", "OfflineRecognizer", "OfflineStream", "OfflineRecognizerResultEntity", "Engine"]


def __getattr__(name):
    if name in ("OfflineRecognizer", "OfflineStream", "OfflineRecognizerResultEntity"):
        from . import offline_recognizer as _m
        return getattr(_m, name)
    if name == "Engine":
        from .engine import Engine
        return Engine
    raise AttributeError(name)
