"""CPU: the C-ABI library loads, exports every symbol include/paraformer_hip.h declares, and
its host-side text stage (pure CPU: timestamps, DecodeMulti, hotword ids, contracts)
reproduces the known-answer vectors.  No device compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from aliparaformerasr_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return N.load()


def test_library_is_in_tree_and_loads(lib):
    assert os.path.dirname(N.LIB_PATH) == os.path.join(ROOT, "aliparaformerasr_amd")
    assert lib.pf_version() == 6


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "paraformer_hip.h"), encoding="utf-8").read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 40
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert missing == []
    # and the ctypes table covers the header exactly
    assert declared == set(N.SIGNATURES)


def test_no_cpu_fallback_engine_create_fails_without_weights(lib):
    cfg = N.PfEngineConfig()
    cfg.struct_size = C.sizeof(N.PfEngineConfig)
    h = C.c_void_p()
    rc = lib.pf_engine_create(C.byref(cfg), C.byref(h))
    assert rc < 0 and not h.value          # PF_ERR_DEVICE here (no GPU) / PF_ERR_INVALID_ARG on a GPU box
    assert lib.pf_last_error()


def _cstrs(lst):
    arr = (C.c_char_p * max(len(lst), 1))(*[s.encode("utf-8") for s in lst])
    return arr


def test_host_timestamps(lib, kat):
    for c in kat["timestamps"]["cases"]:
        peak = np.zeros(c["len"], np.float32)
        peak[c["fires"]] = 1.0
        toks = (C.c_int64 * len(c["tokens"]))(*c["tokens"])
        out = (C.c_int32 * 64)()
        n = lib.pf_host_timestamps(peak.ctypes.data_as(C.POINTER(C.c_float)), c["len"], toks, len(c["tokens"]), out, 32)
        if c["expected"] == "throws":
            assert n == N.PF_ERR_RECOGNITION
        else:
            assert n == len(c["expected"])
            assert [[out[2 * i], out[2 * i + 1]] for i in range(n)] == c["expected"]


def _decode(lib, table, ids, ts):
    tarr = _cstrs(table)
    idarr = (C.c_int64 * max(len(ids), 1))(*ids)
    flat = [v for t in ts for v in t]
    tsi = (C.c_int32 * max(len(flat), 1))(*flat)
    tsl = (C.c_int32 * max(len(ts), 1))(*[len(t) for t in ts])
    d = C.c_void_p()
    N.check(lib.pf_host_decode(tarr, len(table), idarr, len(ids), tsi, tsl, len(ts), C.byref(d)))
    txt, tl = C.c_char_p(), C.c_int32()
    N.check(lib.pf_decoded_text(d, C.byref(txt), tl))
    nt = C.c_int32()
    N.check(lib.pf_decoded_num_tokens(d, nt))
    toks = []
    for j in range(nt.value):
        t = C.c_char_p()
        N.check(lib.pf_decoded_token(d, j, C.byref(t)))
        toks.append(t.value.decode("utf-8"))
    nts = C.c_int32()
    N.check(lib.pf_decoded_num_timestamps(d, nts))
    tss = []
    for j in range(nts.value):
        p, k = C.POINTER(C.c_int32)(), C.c_int32()
        N.check(lib.pf_decoded_timestamp(d, j, C.byref(p), k))
        tss.append([p[m] for m in range(k.value)])
    res = (txt.value.decode("utf-8"), tl.value, toks, tss)
    lib.pf_decoded_free(d)
    return res


def test_host_decode_multi(lib, kat):
    for c in kat["decode_multi"]["cases"]:
        text, tlen, toks, ts = _decode(lib, c["tokens_table"], c["ids"], c["timestamps"])
        assert text == c["text"]
        assert tlen == c["text_len"]
        assert toks == c["tokens"]
        assert ts == c["out_timestamps"]


def test_host_decode_matches_oracle_on_random_sequences(lib):
    from oracle import glue
    rng = np.random.default_rng(0)
    table = ["<blank>", "<s>", "</s>", "<unk>", "你", "好", "世", "he@@", "llo", "▁wor", "ld", "▁a", "b@@", "c", "x▁y",
             "<|en|>", "▁", "@@", "foo\tbar", "的"]
    for _ in range(300):
        n = int(rng.integers(0, 12))
        ids = [int(v) for v in rng.integers(0, len(table), n)]
        ts = [[int(i), int(i + 1)] for i in range(n + int(rng.integers(0, 2)))]
        try:
            exp = glue.decode_multi_one(table, ids, ts)
        except glue.RecognitionFailed:
            exp = None
        try:
            got = _decode(lib, table, ids, ts)
        except N.PfError as e:
            assert e.code == N.PF_ERR_RECOGNITION
            got = None
        assert (got is None) == (exp is None), (ids,)
        if exp is not None:
            assert got == (exp[0], exp[1], exp[2], exp[3]), (ids,)


def test_host_hotword_ids(lib, kat):
    k = kat["hotword_ids"]
    ids = (C.c_int32 * 64)()
    lens = (C.c_int32 * 16)()
    n = C.c_int32()
    N.check(lib.pf_host_hotword_ids(_cstrs(k["tokens_table"]), len(k["tokens_table"]), _cstrs(k["lines"]),
                                    len(k["lines"]), ids, 64, lens, 16, n))
    out, off = [], 0
    for i in range(n.value):
        out.append(list(ids[off: off + lens[i]]))
        off += lens[i]
    assert out == k["expected"]


def test_recognizer_missing_tokens_is_tokens_invalid(tmp_path):
    """Tests/OfflineRecognizerTests .cs:185 — tokensFilePath "" -> Exception("*tokens invalid*");
    raised before any device work, so it holds without a GPU."""
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer, RecognizerException
    with pytest.raises(RecognizerException, match="tokens invalid"):
        OfflineRecognizer(modelFilePath=str(tmp_path / "model.pfw"), configFilePath="", mvnFilePath="",
                          tokensFilePath="")
    empty = tmp_path / "tokens.txt"
    empty.write_text("")
    with pytest.raises(RecognizerException, match="tokens invalid"):
        OfflineRecognizer(str(tmp_path / "model.pfw"), "", "", str(empty))


def test_recognizer_without_device_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer
    tok = tmp_path / "tokens.txt"
    tok.write_text("<blank>\n<s>\n</s>\n")
    with pytest.raises(N.PfError) as ei:
        OfflineRecognizer(str(tmp_path / "model.pfw"), "", "", str(tok))
    assert ei.value.code == N.PF_ERR_DEVICE


@pytest.mark.timeout(300)
def test_host_parsers_under_sanitizers(tmp_path):
    """csrc/json.h + csrc/hostutil.cpp (PFW header / asr.json / asr.yaml / am.mvn / RIFF-WAVE / resampler / UTF-8) built
    with -fsanitize=address,undefined and fed mutated files (tests/native/host_fuzz.cpp): every input parses or is
    rejected with a pf::Error, no report.  (Round 3: `"lfr_n": 1e400` used to be cast to int — now PF_ERR_FORMAT.)"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    cs = os.path.join(root, "aliparaformerasr_amd", "csrc")
    exe = str(tmp_path / "host_fuzz")
    b = subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        "-std=c++17", "-I" + cs, os.path.join(root, "tests", "native", "host_fuzz.cpp"), os.path.join(cs, "hostutil.cpp"),
                        "-o", exe], capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + b.stderr[-300:])
    r = subprocess.run([exe, "3000", str(tmp_path)], capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.startswith("ok ")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("san", ["thread", "address"])
def test_copy_helpers_under_sanitizers(san, tmp_path):
    """csrc/copycrew.cpp (the helper threads that share the host side of a staged AddSamples upload: spinning and sleeping helpers,
    shares taken by helpers and callers alike, several callers on one crew) under -fsanitize=thread / address: every copy exact,
    nothing written outside the destination, no report."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / ("copycrew_" + san))
    cs = os.path.join(root, "aliparaformerasr_amd", "csrc")
    b = subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-g", "-O1", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-std=c++17",
                        "-I" + cs, os.path.join(root, "tests", "native", "copycrew_sanitize.cpp"), os.path.join(cs, "copycrew.cpp"),
                        "-o", exe, "-lpthread"], capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + b.stderr[-300:])
    r = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.startswith("ok ")


@pytest.mark.skipif(not os.environ.get("PF_SANITIZE_FULL"), reason="set PF_SANITIZE_FULL=1: rebuilds the whole library with ASan + UBSan (minutes)")
@pytest.mark.timeout(3000)
def test_every_host_entry_point_under_sanitizers(tmp_path):
    """tools/sanitize_host.sh: the full library with -fsanitize=address,undefined, every pf_host_* entry point driven
    with random arguments (tests/native/abi_host_fuzz.c), plus the shard-runner and parser harnesses."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "tools", "sanitize_host.sh"), str(tmp_path), "3000"], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "runtime error" not in r.stdout + r.stderr and "Sanitizer" not in r.stdout + r.stderr


def test_header_is_plain_c99_and_cxx11(tmp_path):
    """include/paraformer_hip.h is the contract a C# / C / C++ caller binds: it must compile on its own as strict C99 and
    as C++11, and the abi_host_fuzz driver (C) must compile against it."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "paraformer_hip.h"\nint main(void) { pf_engine_config c; (void)c; return sizeof(pf_batch_out) > 0 ? 0 : 1; }\n')
    inc = "-I" + os.path.join(root, "include")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", inc, str(src)],
                ["g++", "-x", "c++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", inc, str(src)],
                ["gcc", "-std=c99", "-Wall", "-fsyntax-only", inc, os.path.join(root, "tests", "native", "abi_host_fuzz.c")]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, (cmd, r.stderr[-2000:])


def test_public_constructor_stream_without_a_recognizer(tmp_path):
    """ABI 6: new OfflineStream(mvnFilePath, confEntity) (OfflineStream.cs:20-34) needs no device — the
    samples wait on the host, SpeechLength follows the sample counts call by call (LFR floor per call, as AddSamples
    appends the features of each call), Tokens / Timestamps / OfflineInputEntity are plain state."""
    from aliparaformerasr_amd import weights as W
    from aliparaformerasr_amd.offline_recognizer import (ArgumentNullException, ConfEntity, FrontendConfEntity,
                                                         OfflineInputEntity, OfflineStream)
    from oracle import frontend as fe
    shift, scale = W.synth_cmvn()
    mvn = tmp_path / "am.mvn"
    mvn.write_text(fe.format_mvn_text(shift, scale))
    s = OfflineStream(str(mvn), ConfEntity(FrontendConfEntity(dither=0.0)))
    assert s.Tokens == [0, 0] and s.Hyp == [0, 0] and s.Timestamps == [] and s.Hotwords == []
    assert s.OfflineInputEntity.Speech is None and s.SpeechLength == 0
    with pytest.raises(ArgumentNullException):
        s.AddSamples(None)
    want = 0
    for n in (16000, 7, 32000 + 159):
        s.AddSamples(np.zeros(n, np.float32))
        want += ((n + 80) // 160) // 6 * 560
        assert s.SpeechLength == want
    s.Tokens = [5, 6, 7]
    assert s.Tokens == [5, 6, 7]
    s.Timestamps = [[0, 60], [60, 120, 120, 180]]
    assert s.Timestamps == [[0, 60], [60, 120, 120, 180]]
    s.RemoveChunk()                                  # more than two tokens: Speech = null, SpeechLength = 0 (:69-79)
    assert s.OfflineInputEntity.Speech is None and s.SpeechLength == 0
    e = OfflineInputEntity(Speech=np.arange(1120, dtype=np.float32), SpeechLength=1120, Hotwords=[[3, 4], [9]])
    s.OfflineInputEntity = e
    got = s.GetDecodeChunk()
    np.testing.assert_array_equal(got.Speech, e.Speech)
    assert got.SpeechLength == 1120 and got.Hotwords == [[3, 4], [9]]
    s.Dispose()
    from aliparaformerasr_amd.offline_recognizer import ObjectDisposedException
    with pytest.raises(ObjectDisposedException):
        s.AddSamples(np.zeros(10, np.float32))
