"""Command-line harness mirroring AliParaformerAsr.Examples (`-type offline` and `-type online`).

    python -m aliparaformerasr_amd.examples -type offline -method batch -base <dir> -model <name> \
        [-accuracy int8] [-threads 2] -files a.wav b.wav
    python -m aliparaformerasr_amd.examples -type online -method one -base <dir> -model <name> -files a.wav

Mirrors (file:line in /root/reference/AliParaformerAsr.Examples):
  * argument handling and defaults — Program.cs:93-104, ParseArgs :197-252: the MANYSPEECH_BASE / _TYPE / _BATCH / _MODEL /
    _ACCURACY / _THREADS environment variables are the defaults (Program.cs:20-28), command-line parameters overwrite
    them; a recognizer type must come from one of the two, unknown flags are an error;
  * model-directory file selection — OfflineAliParaformerAsrRecognizer.cs:24-100: `model*` (not `_eb`)
    preferring a name containing ".<accuracy>.", else the last; last `asr*.yaml|json`, `am*.mvn`, `tokens*.txt`,
    `hotword*.txt` (the model file here is a .pfw container instead of .onnx);
  * sample loading — IsAudioByHeader + GetFileSample (:121-160, Utils/AudioHelper.cs:12-32) through the native
    pf_host_wav_read / pf_host_is_audio; default file list = every *.wav under the model directory;
  * "-method one" / "-method batch" loops, the JSON-ish result line and the timing lines — :169-249 (the timed
    window includes stream creation, AddSamples, GetResults, printing and Dispose, as upstream);
  * `-type online` (Program.cs:290-297) — OnlineAliParaformerAsrRecognizer.cs:8-279: encoder / decoder file selection
    (:43-63), at most TWO media files (:121-171), 9600-sample chunks (AudioHelper.GetFileChunkSamples :80-127) plus six
    400-sample silence chunks (:160-163), one AddSamples + GetResult + printed text per chunk (:181-194; only the "one"
    method exists upstream — the batch loop is commented out there), timing lines :272-279."""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np

from . import _native as N


def get_file_sample(path: str):
    """AudioHelper.GetFileSample -> (float32 samples, duration_ms)."""
    lib = N.load()
    n = C.c_int64(); sr = C.c_int32(); ch = C.c_int32(); dur = C.c_double()
    N.check(lib.pf_host_wav_read(path.encode(), None, 0, n, sr, ch, dur))
    out = np.zeros(max(n.value, 1), np.float32)
    N.check(lib.pf_host_wav_read(path.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, n, sr, ch, dur))
    return out[: n.value], dur.value


def is_audio_by_header(path: str) -> bool:
    v = C.c_int32(0)
    N.check(N.load().pf_host_is_audio(path.encode(), v))
    return bool(v.value)


def select_model_files(base: str, model: str, accuracy: str):
    folder = os.path.join(base, model)
    if not os.path.isdir(folder):
        print("Error: folder does not exist - %s" % folder)
        return None
    names = sorted(f for f in os.listdir(folder) if os.path.isfile(os.path.join(folder, f)))
    full = lambda f: os.path.join(base, model, f)

    def last(pred):
        m = [f for f in names if pred(f)]
        return full(m[-1]) if m else ""

    cands = [f for f in names if f.startswith("model") and "_eb" not in f]
    pref = [f for f in cands if (".%s." % accuracy) in f]
    model_path = full(pref[-1]) if pref else (full(cands[-1]) if cands else "")
    eb = [f for f in names if f.startswith("model_eb")]
    ebp = [f for f in eb if (".%s." % accuracy) in f]
    return dict(
        modelFilePath=model_path,
        modelebFilePath=full(ebp[-1]) if ebp else (full(eb[-1]) if eb else ""),
        configFilePath=last(lambda f: f.startswith("asr") and (f.endswith(".yaml") or f.endswith(".json"))),
        mvnFilePath=last(lambda f: f.startswith("am") and f.endswith(".mvn")),
        tokensFilePath=last(lambda f: f.startswith("tokens") and f.endswith(".txt")),
        hotwordFilePath=last(lambda f: f.startswith("hotword") and f.endswith(".txt")),
    )


# ---- the GUI's display step for SenseVoice results (AliParaformerAsr.Examples.MauiApp/Utils/AEDEmojiHelper.cs:7-47; callers
#      RecognitionForFiles.xaml.cs:478,583: `ReplaceTagsWithEmojis(result.Text.Replace("> ", ">"))`) — SURVEY §8f row 3
_EMOJI_MAP = {
    "Laughter": "\U0001F606", "Applause": "\U0001F44F", "HAPPY": "\U0001F600", "SAD": "\U0001F622", "ANGRY": "\U0001F621",
    "NEUTRAL": "\U0001F610", "FEARFUL": "\U0001F628", "DISGUSTED": "\U0001F922", "SURPRISED": "\U0001F632", "Cry": "\U0001F62D",
    "Sneeze": "\U0001F443\U0001F927", "Cough": "\U0001F912", "Sing": "\U0001F3A4",
}


def replace_tags_with_emojis(text: str) -> str:
    """AEDEmojiHelper.ReplaceTagsWithEmojis (:7-36): every `<|word|>` tag becomes its emoji, or nothing when the tag is
    not in the table (language / event / itn tags such as <|zh|>, <|Speech|>, <|woitn|> simply disappear)."""
    import re
    return re.sub(r"<\|(\w+)\|>", lambda m: _EMOJI_MAP.get(m.group(1), ""), text)


def replace_tags_with_empty(text: str) -> str:
    """AEDEmojiHelper.ReplaceTagsWithEmpty (:38-45): `<|...|>` (shortest match, anything but a newline inside) removed."""
    import re
    return re.sub(r"<\|.*?\|>", "", text)


def display_text(text: str) -> str:
    """What the GUI shows for a result (RecognitionForFiles.xaml.cs:478): DecodeMulti separates tags with "> "."""
    return replace_tags_with_emojis(text.replace("> ", ">"))


def _result_line(r) -> str:
    toks = ",".join('"%s"' % t for t in r.Tokens)
    ts = ",".join("[%d,%d]" % (t[0], t[-1]) for t in r.Timestamps)
    return '{"text": "%s","tokens":[%s],"timestamps":[%s]}' % (r.Text, toks, ts)


def offline_recognizer(method="one", model="paraformer-seaco-large-zh-timestamp-onnx-offline", accuracy="int8",
                       threads=2, files=None, base=None, out=sys.stdout):
    from .offline_recognizer import OfflineRecognizer
    base = base or os.getcwd()
    sel = select_model_files(base, model, accuracy)
    if sel is None or not sel["modelFilePath"] or not sel["tokensFilePath"]:
        print("Init models failure!", file=out)
        return None
    t0 = time.perf_counter()
    rec = OfflineRecognizer(threadsNum=threads, **sel)
    print("init_models_elapsed_milliseconds:%s" % ((time.perf_counter() - t0) * 1e3), file=out)
    if not files:
        files = []
        for d, _dirs, fs in os.walk(os.path.join(base, model)):
            files += [os.path.join(d, f) for f in sorted(fs) if f.lower().endswith(".wav")]
    samples, paths, total_ms = [], [], 0.0
    for f in files:
        if not os.path.isfile(f) or not is_audio_by_header(f):
            continue
        s, dur = get_file_sample(f)
        paths.append(f); samples.append(s); total_ms += dur
    if not samples:
        print("No media file is read!", file=out)
        return None
    print("Automatic speech recognition in progress!", file=out)
    t0 = time.perf_counter()
    method = method or "batch"
    results = []
    print("Recognition results:\r\n", file=out)
    try:
        if method == "one":
            for p, s in zip(paths, samples):
                st = rec.CreateOfflineStream()
                st.AddSamples(s)
                r = rec.GetResult(st)
                results.append(r)
                print(p, file=out); print(_result_line(r), file=out); print("", file=out)
        elif method == "batch":
            streams = []
            for s in samples:
                st = rec.CreateOfflineStream()
                st.AddSamples(s)
                streams.append(st)
            results = rec.GetResults(streams)
            for p, r in zip(paths, results):
                print(p, file=out); print(_result_line(r), file=out); print("", file=out)
    except Exception as ex:          # the reference prints the message and carries on to the timing lines
        print(str(ex), file=out)
    rec.Dispose()
    elapsed = (time.perf_counter() - t0) * 1e3
    print("recognition_elapsed_milliseconds:%s" % elapsed, file=out)
    print("total_duration_milliseconds:%s" % total_ms, file=out)
    print("rtf:%s" % (elapsed / total_ms if total_ms else float("inf")), file=out)
    print("end!", file=out)
    return results


def select_online_model_files(base: str, model: str, accuracy: str):
    """OnlineAliParaformerAsrRecognizer.cs:15-84 (the containers here are .pfw instead of .onnx)."""
    folder = os.path.join(base, model)
    if not os.path.isdir(folder):
        print("Error: folder does not exist - %s" % folder)
        return None
    names = sorted(f for f in os.listdir(folder) if os.path.isfile(os.path.join(folder, f)))
    full = lambda f: os.path.join(base, model, f)

    def pick(cands):
        pref = [f for f in cands if (".%s." % accuracy) in f]
        return full(pref[-1]) if pref else (full(cands[-1]) if cands else "")

    def last(pred):
        m = [f for f in names if pred(f)]
        return full(m[-1]) if m else ""
    return dict(
        encoderFilePath=pick([f for f in names if f.startswith("model") or f.startswith("encoder")]),
        decoderFilePath=pick([f for f in names if f.startswith("decoder")]),
        configFilePath=last(lambda f: f.startswith("asr") and (f.endswith(".yaml") or f.endswith(".json"))),
        mvnFilePath=last(lambda f: f.startswith("am") and f.endswith(".mvn")),
        tokensFilePath=last(lambda f: f.startswith("tokens")),
    )


def get_file_chunk_samples(path: str, chunk: int = 160 * 6 * 10):
    """AudioHelper.GetFileChunkSamples (:80-127): GetFileSample's samples cut into 9600-sample pieces, the last one shorter."""
    s, dur = get_file_sample(path)
    return [s[i: i + chunk] for i in range(0, len(s), chunk)], dur


def online_recognizer(method="one", model="speech_paraformer-large_asr_nat-zh-cn-16k-common-vocab8404-online-onnx",
                      accuracy="int8", threads=2, files=None, base=None, out=sys.stdout):
    from .online_recognizer import OnlineRecognizer
    base = base or os.getcwd()
    if not model:
        print("Init models failure!", file=out)
        return None
    sel = select_online_model_files(base, model, accuracy)
    if sel is None or not sel["encoderFilePath"] or not sel["tokensFilePath"]:
        print("Init models failure!", file=out)
        return None
    t0 = time.perf_counter()
    try:
        rec = OnlineRecognizer(threadsNum=threads, **sel)
    except Exception as ex:                                 # :97-100
        print("Error occurred: %s" % ex, file=out)
        print("Init models failure!", file=out)
        return None
    print("init_models_elapsed_milliseconds:%s" % ((time.perf_counter() - t0) * 1e3), file=out)
    t0 = time.perf_counter()                                # :126: the window opens before the files are read
    if not files:
        files = []
        for d, _dirs, fs in os.walk(os.path.join(base, model)):
            files += [os.path.join(d, f) for f in sorted(fs) if f.lower().endswith(".wav")]
    samples_list, total_ms, n = [], 0.0, 0
    for f in files:
        if n >= 2:                                          # batchSize = 2 (:129, :152-155)
            break
        if not os.path.isfile(f):
            continue
        if is_audio_by_header(f):
            chunks, dur = get_file_chunk_samples(f)
            if chunks:
                chunks += [np.zeros(400, np.float32) for _ in range(6)]          # :160-163
                samples_list.append(chunks)
                total_ms += dur
        n += 1
    if not samples_list:
        print("No media file is read!", file=out)
        return None
    method = method or "batch"
    texts = []
    if method == "one":                                     # :176-194 (the only method the reference runs)
        for chunks in samples_list:
            st = rec.CreateOnlineStream()
            for c in chunks:
                st.AddSamples(c)
                r = rec.GetResult(st)
                print(r.Text, file=out)
                texts.append(r.Text)
    rec.Dispose()
    elapsed = (time.perf_counter() - t0) * 1e3
    print("elapsed_milliseconds:%s" % elapsed, file=out)
    print("total_duration:%s" % total_ms, file=out)
    print("rtf:%s" % (elapsed / total_ms if total_ms else float("inf")), file=out)
    print("Hello, World!", file=out)
    return texts


def parse_args(argv, env=None):
    # environment variables are the defaults, command-line parameters overwrite them (Program.cs:20-28, :93-104)
    env = os.environ if env is None else env
    try:
        threads = int(env.get("MANYSPEECH_THREADS", "2"))
    except ValueError:
        raise ValueError("The number of threads must be a valid integer")
    cfg = dict(modelBasePath=env.get("MANYSPEECH_BASE", ""), recognizerType=env.get("MANYSPEECH_TYPE"),
               methodType=env.get("MANYSPEECH_BATCH", "one"), modelName=env.get("MANYSPEECH_MODEL", "default-model"),
               modelAccuracy=env.get("MANYSPEECH_ACCURACY", "int8"), threads=threads, files=[])
    i = 0
    while i < len(argv):
        a = argv[i].lower()
        if a in ("-base", "-type", "-method", "-model", "-accuracy"):
            key = {"-base": "modelBasePath", "-type": "recognizerType", "-method": "methodType", "-model": "modelName",
                   "-accuracy": "modelAccuracy"}[a]
            if i + 1 < len(argv):
                i += 1
                cfg[key] = argv[i]
        elif a == "-threads":
            try:
                i += 1
                cfg["threads"] = int(argv[i])
            except (IndexError, ValueError):
                raise ValueError("The number of threads must be a valid integer")
        elif a == "-files":
            fs = []
            while i + 1 < len(argv) and not argv[i + 1].startswith("-"):
                i += 1
                fs.append(argv[i].strip('"'))
            cfg["files"] = fs
        else:
            raise ValueError("Unknown parameters: %s" % argv[i])
        i += 1
    if cfg["recognizerType"] is None:
        raise ValueError("You must specify the recognizer type (-type online/offline)")
    return cfg


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    try:
        cfg = parse_args(argv)
    except ValueError as ex:
        print("parameter error: %s" % ex)
        return 2
    # Program.cs:290-308
    if cfg["recognizerType"] == "online":
        online_recognizer(cfg["methodType"], cfg["modelName"], cfg["modelAccuracy"], cfg["threads"], cfg["files"],
                          cfg["modelBasePath"] or None)
    elif cfg["recognizerType"] == "offline":
        offline_recognizer(cfg["methodType"], cfg["modelName"], cfg["modelAccuracy"], cfg["threads"], cfg["files"],
                           cfg["modelBasePath"] or None)
    else:
        print("the recognizer type must be online or offline")
        return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
