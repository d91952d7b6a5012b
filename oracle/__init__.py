"""CPU oracle for the offline Paraformer / SenseVoice path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``aliparaformerasr_amd``) never imports this package and
fails loudly if its HIP library is missing.

What it restates (reference = manyeyes/AliParaformerAsr, paths relative to
``/root/reference``):

* glue arithmetic that lives in the reference's own C# — LFR, CMVN, padding +
  sentinel, last-index arg-max, CIF-peak timestamps, DecodeMulti, hotword id
  lookup, SenseVoice prompt ids, SeACo bias_embed layout.  These are pinned by
  known-answer vectors hand-evaluated from the reference source
  (``tests/golden/*.json``; see ``tests/golden/make_golden.py``).
* arithmetic that lives in the reference's un-vendored dependencies —
  ``ManySpeech.SpeechFeatures 1.1.7`` (kaldi-native-fbank) and the FunASR ONNX
  graphs executed by ``Microsoft.ML.OnnxRuntime 1.22.*``
  (``AliParaformerAsr/AliParaformerAsr.csproj:49-50``).  Neither package nor any
  model file is present, and the reference's tests hold no numeric vectors, so
  for this part the oracle restates the *published* algorithms (kaldi fbank,
  FunASR SANMEncoder / CifPredictorV2 / ParaformerSANMDecoder export code, the
  CifPredictorV3 BiCIF timestamp head, the SeACo hotword embedder / bias decoder /
  NO-BIAS merge).

PARITY UNPINNED for the model arithmetic (encoder / predictor / decoder /
BiCIF head / SeACo branch / fbank): there is no runnable reference and no golden vector for it.  The glue
functions are pinned.
"""
