"""Generates the committed golden fixtures under tests/golden/.

Run in the authoring container only (it reads /root/reference, which does not exist on the
GPU box; the tests read the generated files, never the reference):

    python tests/golden/make_golden.py

1. sensevoice_embed.npy — the 16x560 float32 table held by the reference's embedded resource
   AliParaformerAsr/data/embed.onnx (a single Gather over `weight`; used by
   AliParaformerAsr/EmbedSVModel.cs:45-77).  This is reference DATA, extracted with a
   hand-rolled protobuf walk (no onnx package in the image).
2. kat.json — known-answer vectors for the reference's own C# glue, hand-evaluated from the
   reference source with float32 emulation; each entry cites the lines it
   was derived from.  They pin the oracle AND the native host code.
"""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/AliParaformerAsr"


def _varint(buf, i):
    v, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        s += 7
        if not b & 0x80:
            return v, i


def _fields(buf):
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("wire type %d" % wt)
        yield fno, wt, v


def extract_embed():
    data = open(os.path.join(REF, "data", "embed.onnx"), "rb").read()
    graph = next(v for f, w, v in _fields(data) if f == 7 and w == 2)          # ModelProto.graph
    for f, w, v in _fields(graph):
        if f == 5 and w == 2:                                                   # GraphProto.initializer
            dims, raw, name, dtype = [], None, None, None
            for ff, ww, vv in _fields(v):
                if ff == 1 and ww == 0:
                    dims.append(vv)
                elif ff == 1 and ww == 2:                                       # packed dims
                    j = 0
                    while j < len(vv):
                        d, j = _varint(vv, j)
                        dims.append(d)
                elif ff == 2:
                    dtype = vv
                elif ff == 8:
                    name = vv.decode()
                elif ff == 9:
                    raw = vv
            if raw is not None and dtype == 1:
                arr = np.frombuffer(raw, dtype="<f4").reshape(dims)
                print("initializer", name, arr.shape, "mean %.4f std %.4f" % (arr.mean(), arr.std()), arr[0, :3])
                return arr.copy()
    raise RuntimeError("no float initializer found")


F32 = np.float32
S = float(F32(F32(-23.025850929940457) * F32(32768.0)))

KAT = {
    "_comment": "hand-evaluated from the reference C# source; see make_golden.py and SURVEY.md 8c",
    "pad_sentinel": {
        "cite": "AliParaformerAsr/Utils/PadHelper.cs:23-65 (sentinel :63)",
        "value": S,
        "inputs": [[1.0, 0.0, 2.0], [3.0]],
        "expected": [[1.0, S, 2.0], [3.0, S, S]],
    },
    "lfr": {
        "cite": "AliParaformerAsr/WavFrontend.cs:73-111 (m=7, n=6; frame f has all 80 values = f, f=1..13)",
        "t80": 13,
        "expected_frame_values": [[0, 0, 0, 1, 2, 3, 4], [4, 5, 6, 7, 8, 9, 10]],
        "frame_counts": {"1": 0, "5": 0, "6": 1, "11": 1, "12": 2, "2998": 499, "3000": 500},
    },
    "cmvn": {
        "cite": "AliParaformerAsr/WavFrontend.cs:53-71: (x + shift[k]) * scale[k]",
        "x": [1.0, -2.0, 0.5], "shift": [-8.0, 1.0, 0.25], "scale": [0.5, 2.0, 4.0],
        "expected": [-3.5, -2.0, 3.0],
    },
    "argmax": {
        "cite": "AliParaformerAsr/OfflineRecognizer.cs:139-152: cur = x[cur] > x[k] ? cur : k (ties -> larger index)",
        "cases": [
            {"x": [0.5, 0.7, 0.7, 0.1], "expected": 2},
            {"x": [1.0, 1.0, 1.0, 1.0], "expected": 3},
            {"x": [3.0, 1.0, 2.0], "expected": 0},
            {"x": [0.0, "nan", 1.0, 0.5], "expected": 2},
            {"x": [5.0, 1.0, "nan"], "expected": 2},
            {"x": ["nan", 2.0, 1.0], "expected": 1},
        ],
    },
    "timestamps": {
        "cite": "AliParaformerAsr/OfflineRecognizer.cs:200-302 (float32 arithmetic, (int)(t*1000) truncation)",
        "cases": [
            {"len": 90, "fires": [10, 22, 40, 75], "tokens": [100, 200, 300, 2],
             "expected": [[170, 410], [410, 770], [770, 1635]]},
            {"len": 60, "fires": [3, 15, 58], "tokens": [100, 200, 2, 7],
             "expected": [[30, 269], [269, 1199]]},
            {"len": 40, "fires": [], "tokens": [5, 2], "expected": "throws"},
        ],
    },
    "decode_multi": {
        "cite": "AliParaformerAsr/OfflineRecognizer.cs:304-418",
        "cases": [
            {"tokens_table": ["<blank>", "<s>", "</s>", "欢", "迎", "he@@", "llo", "world", "x"],
             "ids": [3, 4, 5, 6, 7, 2, 8],
             "timestamps": [[0, 1], [1, 2], [2, 3], [3, 4], [4, 5], [5, 6], [6, 7]],
             "text": "欢迎 hello world ", "text_len": 15,
             "tokens": ["欢", "迎", "hello", "world"],
             "out_timestamps": [[0, 1], [1, 2], [2, 3, 3, 4], [4, 5]]},
            {"tokens_table": ["<blank>", "<s>", "</s>", "<unk>", "▁hello", "world", "▁foo", "bar"],
             "ids": [3, 4, 5, 6, 7, 3],
             "timestamps": [[0, 1], [1, 2], [2, 3], [3, 4], [4, 5], [5, 6]],
             "text": "helloworld foobar", "text_len": 17,
             "tokens": ["helloworld", "foobar"],
             "out_timestamps": [[1, 2, 2, 3], [3, 4, 4, 5]]},
            {"tokens_table": ["<blank>", "<s>", "</s>", "<|zh|>", "<|NEUTRAL|>", "<|Speech|>", "<|woitn|>", "你", "好"],
             "ids": [3, 4, 5, 6, 7, 8],
             "timestamps": [[0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 0]],
             "text": " <|zh|> <|NEUTRAL|> <|Speech|> <|woitn|> 你好", "text_len": 43,
             "tokens": ["<|zh|>", "<|NEUTRAL|>", "<|Speech|>", "<|woitn|>", "你", "好"],
             "out_timestamps": [[0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 0]]},
        ],
    },
    "sensevoice_ids": {
        "cite": "AliParaformerAsr/OfflineProjOfSenseVoiceSmall.cs:57-74 (languageId overwritten by the textnorm lookup)",
        "use_itn_true": {"language": 14, "textnorm": 15, "prompt_rows": [14, 1, 2, 15]},
        "use_itn_false": {"language": 15, "textnorm": 15, "prompt_rows": [15, 1, 2, 15]},
    },
    "seaco_padlist": {
        "cite": "AliParaformerAsr/EmbedSeacoModel.cs:110-123 PadList(hotwords, 0, 10)",
        "hotwords": [[5, 6], [1], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]],
        "expected": [[5, 6, 0, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]],
        "bias_embed_rule": "bias_embed[b, n*10 + j, :] = hw_embed[j, n, :] (OfflineProjOfSeacoParaformer.cs:92-107)",
    },
    "hotword_ids": {
        "cite": "AliParaformerAsr/OfflineRecognizer.cs:72-90 (per-char Array.IndexOf, -1 dropped, [1] appended)",
        "tokens_table": ["<blank>", "<s>", "</s>", "魔", "搭", "a", "b\tx"],
        "lines": ["魔搭", "a?b", ""],
        "expected": [[3, 4], [5], [], [1]],
    },
}


def main():
    emb = extract_embed()
    assert emb.shape == (16, 560)
    np.save(os.path.join(HERE, "sensevoice_embed.npy"), emb.astype(np.float32))
    with open(os.path.join(HERE, "kat.json"), "w", encoding="utf-8") as f:
        json.dump(KAT, f, ensure_ascii=True, indent=1)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
