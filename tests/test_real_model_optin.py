"""Opt-in real-model parity (SURVEY.md §8c plan).  Skipped unless PF_MODEL_DIR points at a downloaded reference
model directory (model.onnx / model.int8.onnx, asr.yaml, am.mvn, tokens.txt, optional model_eb*.onnx, *.wav).
Neither such a directory nor onnxruntime exists in the build image, so this test has never run there; it documents
and automates the check a user with the files can make:
  1. ONNX -> PFW through aliparaformerasr_amd.convert (graph walk; an int8 file's bytes carried beside their float image),
  2. recognise every *.wav under the directory with the HIP path,
  3. if `onnxruntime` is importable: run the same padded features through the ONNX graph on CPU and require the
     arg-max token ids to be identical wherever the ORT top-1/top-2 log-prob margin exceeds 0.1."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MODEL_DIR = os.environ.get("PF_MODEL_DIR", "")


@pytest.mark.skipif(not MODEL_DIR or not os.path.isdir(MODEL_DIR), reason="PF_MODEL_DIR not set")
def test_real_model_tokens_vs_onnxruntime(tmp_path):
    from aliparaformerasr_amd import convert as cv, examples as ex, weights as W
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer
    sel = ex.select_model_files(os.path.dirname(MODEL_DIR.rstrip("/")), os.path.basename(MODEL_DIR.rstrip("/")), os.environ.get("PF_ACCURACY", "int8"))
    onnx_path = sel["modelFilePath"] if sel["modelFilePath"].endswith(".onnx") else glob.glob(os.path.join(MODEL_DIR, "model*.onnx"))[0]
    conf = open(sel["configFilePath"]).read() if sel["configFilePath"] else ""
    kind = "sensevoicesmall" if "sensevoice" in conf.lower() else ("seacoparaformer" if "seaco" in conf.lower() else "paraformer")
    cfg, wts = cv.onnx_to_pfw(onnx_path, sel["modelebFilePath"] or None, kind)
    pfw = str(tmp_path / "model.pfw")
    W.save_pfw(pfw, cfg, wts)
    rec = OfflineRecognizer(pfw, sel["configFilePath"], sel["mvnFilePath"], sel["tokensFilePath"], hotwordFilePath=sel["hotwordFilePath"])
    wavs = sorted(glob.glob(os.path.join(MODEL_DIR, "**", "*.wav"), recursive=True))[:4]
    assert wavs, "no wav files under PF_MODEL_DIR"
    streams = []
    for wv in wavs:
        s, _ = ex.get_file_sample(wv)
        st = rec.CreateOfflineStream()
        st.AddSamples(s)
        streams.append(st)
    feats = [np.asarray(st.Speech, np.float32).reshape(-1, 560) for st in streams]
    results = rec.GetResults(streams)
    for wv, r in zip(wavs, results):
        print(wv, r.Text)
    ort = pytest.importorskip("onnxruntime")
    from oracle import frontend as fe
    T = max(f.shape[0] for f in feats)
    speech = fe.pad_sequence(feats).reshape(len(feats), T, 560)
    sess = ort.InferenceSession(onnx_path, providers=["CPUExecutionProvider"])
    feed = {"speech": speech, "speech_lengths": np.full(len(feats), T, np.int32)}
    feed = {k: v for k, v in feed.items() if k in [i.name for i in sess.get_inputs()]}
    out = sess.run(None, feed)[0]
    ids_ref = np.argmax(out, -1)
    srt = np.sort(out, -1)
    safe = (srt[..., -1] - srt[..., -2]) > 0.1
    for b, st in enumerate(streams):
        ids = np.asarray(st.Tokens)
        L = min(len(ids), ids_ref.shape[1])
        assert np.array_equal(ids[:L][safe[b, :L]], ids_ref[b, :L][safe[b, :L]])
