"""Oracle: the reference's managed post-processing (timestamps, text decoding, hotwords).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function follows the C# source line
by line; pinned by tests/golden/kat.json.

  * time_stamp_lfr6_onnx  AliParaformerAsr/OfflineRecognizer.cs:200-302
  * DecodeMulti           AliParaformerAsr/OfflineRecognizer.cs:304-418 (+ IsChinese :428-439)
  * GetHotwords           AliParaformerAsr/OfflineRecognizer.cs:72-90
  * PadList               AliParaformerAsr/EmbedSeacoModel.cs:110-123
  * bias_embed layout     AliParaformerAsr/OfflineProjOfSeacoParaformer.cs:83-111
  * SenseVoice prompt ids AliParaformerAsr/OfflineProjOfSenseVoiceSmall.cs:57-106
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
BAR = "▁"


class RecognitionFailed(Exception):
    """Stands for `throw new Exception("Offline recognition failed", ex)` (:194-197)."""


def time_stamp_lfr6_onnx(us_cif_peak, tokens, begin_time=0.0, total_offset=-1.5):
    us = np.asarray(us_cif_peak, dtype=F32)
    tokens = [int(t) for t in tokens]
    START_END_THRESHOLD = 5
    MAX_TOKEN_DURATION = 30
    TIME_RATE = F32(F32(F32(10.0) * F32(6)) / F32(1000)) / F32(3)
    num_frames = len(us)
    if not tokens:
        raise RecognitionFailed("tokens.Last() on empty")
    if tokens[-1] == 2:
        tokens = tokens[:-1]
    fire_place = [F32(F32(i) + F32(total_offset)) for i in range(num_frames) if float(us[i]) > float(F32(1.0)) - 1e-4]
    if not fire_place:
        raise RecognitionFailed("fire_place[0] on empty list")
    timestamp_list = []
    new_char_list = []
    if fire_place[0] > START_END_THRESHOLD:
        timestamp_list.append([F32(0.0), F32(fire_place[0] * TIME_RATE)])
        new_char_list.append(False)
    n = len(fire_place)
    for i in range(n - 1):
        if i >= len(tokens):
            raise RecognitionFailed("tokens index out of range")
        new_char_list.append(tokens[i] != 1)
        if i == n - 2 or MAX_TOKEN_DURATION < 0 or F32(fire_place[i + 1] - fire_place[i]) < MAX_TOKEN_DURATION:
            timestamp_list.append([F32(fire_place[i] * TIME_RATE), F32(fire_place[i + 1] * TIME_RATE)])
        else:
            split = F32(fire_place[i] + F32(MAX_TOKEN_DURATION))
            timestamp_list.append([F32(fire_place[i] * TIME_RATE), F32(split * TIME_RATE)])
            timestamp_list.append([F32(split * TIME_RATE), F32(fire_place[i + 1] * TIME_RATE)])
            new_char_list.append(False)
    if F32(F32(num_frames) - fire_place[-1]) > START_END_THRESHOLD:
        end = F32(F32(F32(num_frames) + fire_place[-1]) / F32(2))
        if not timestamp_list:
            raise RecognitionFailed("timestamp_list.Last() on empty")
        timestamp_list[-1][1] = F32(end * TIME_RATE)
        timestamp_list.append([F32(end * TIME_RATE), F32(F32(num_frames) * TIME_RATE)])
        new_char_list.append(False)
    else:
        if not timestamp_list:
            raise RecognitionFailed("timestamp_list.Last() on empty")
        timestamp_list[-1][1] = F32(F32(num_frames) * TIME_RATE)
    if begin_time > 0.0:
        for t in timestamp_list:
            t[0] = F32(t[0] + F32(F32(begin_time) / F32(1000.0)))
            t[1] = F32(t[1] + F32(F32(begin_time) / F32(1000.0)))
    new_char_list.append(True)
    out = []
    for c, t in zip(new_char_list, timestamp_list):
        if c:
            out.append([int(F32(t[0] * F32(1000))), int(F32(t[1] * F32(1000)))])
    return out


def is_chinese_all(s: str) -> bool:
    return len(s) > 0 and all("一" <= ch <= "龥" for ch in s)


def _remove_first_equal_to_last(lst):
    if not lst:
        raise RecognitionFailed("Last() on empty")
    lst.remove(lst[-1])      # list.remove drops the FIRST equal element, like List<T>.Remove


def decode_multi_one(token_table, ids, timestamps):
    """Returns (Text, TextLen, Tokens, Timestamps) for one stream."""
    text = ""
    last_token = ""
    last_ts = None
    out_tokens, out_ts = [], []
    for token, ts in zip(ids, timestamps):
        token = int(token)
        if token == 2:
            break
        cur = token_table[token].split("\t")[0]
        if cur in ("</s>", "<s>", "<blank>", "<unk>"):
            continue
        if is_chinese_all(cur):
            text += cur
            out_tokens.append(cur)
            out_ts.append(list(ts))
            continue
        text += BAR + cur + BAR
        comb = last_token + BAR + cur + BAR
        if comb.find("@@" + BAR + BAR) > 0:
            cur_token = comb.replace("@@" + BAR + BAR, "")
            cur_ts = list(ts) if last_ts is None else list(last_ts) + list(ts)
            _remove_first_equal_to_last(out_tokens)
            out_tokens.append(cur_token.replace(BAR, ""))
            if not out_ts:
                raise RecognitionFailed("Last() on empty")
            out_ts.pop()
            out_ts.append(cur_ts)
            last_token, last_ts = cur_token, cur_ts
        elif comb.count(BAR) in (3, 5) and comb.find(BAR * 3) < 0:
            cur_token = comb.replace(BAR + BAR, "")
            cur_ts = list(ts) if last_ts is None else list(last_ts) + list(ts)
            if out_tokens:
                _remove_first_equal_to_last(out_tokens)
            out_tokens.append(cur_token.replace(BAR, ""))
            if out_ts:
                out_ts.pop()
            out_ts.append(cur_ts)
            last_token, last_ts = cur_token, cur_ts
        else:
            out_tokens.append(cur.replace(BAR, ""))
            out_ts.append(list(ts))
            last_token, last_ts = BAR + cur + BAR, list(ts)
    if text.find("@@" + BAR + BAR) > 0 or text.find(BAR * 3) < 0:
        text = text.replace("@@" + BAR + BAR, "").replace(BAR + BAR, " ").replace("@@", " ").replace(BAR, " ")
    else:
        text = text.replace(BAR * 3, " ").replace(BAR + BAR, "").replace(BAR, "")
    text_len = len(text.encode("utf-16-le")) // 2
    return text, text_len, out_tokens, out_ts


def hotword_ids(token_table, lines, sos_eos_id=1):
    hw = []
    for sentence in lines:
        ids = []
        for unit in sentence.encode("utf-16-le").decode("utf-16-le", "surrogatepass"):
            if ord(unit) >= 0x10000:
                continue     # ToCharArray yields two surrogate halves; neither is a token
            try:
                ids.append(token_table.index(unit))
            except ValueError:
                pass
        hw.append(ids)
    hw.append([sos_eos_id])
    return hw


def pad_list(hotwords, padding_value=0, max_length=10):
    out = []
    for hw in hotwords:
        hw = list(hw)
        out.append(hw[:max_length] if len(hw) > max_length else hw + [padding_value] * (max_length - len(hw)))
    return out


def bias_embed(hw_embed: np.ndarray, batch: int) -> np.ndarray:
    """hw_embed [10, N, 512] -> bias_embed [B, 10N, 512] with row n*10 + j = hw_embed[j, n]."""
    j, n, d = hw_embed.shape
    flat = np.transpose(hw_embed, (1, 0, 2)).reshape(n * j, d)
    return np.tile(flat[None], (batch, 1, 1)).astype(F32)


def sensevoice_prompt_ids(use_itn: bool):
    """Effective ids (quirk Q7): languageId is overwritten by the textnorm lookup."""
    language = 14 if use_itn else 15
    textnorm = 15
    return language, textnorm, [language, 1, 2, textnorm]


def sensevoice_prepend(speech: np.ndarray, embed: np.ndarray, use_itn: bool) -> np.ndarray:
    """speech [T,560] -> [T+4,560] (OfflineProjOfSenseVoiceSmall.cs:78-106)."""
    rows = sensevoice_prompt_ids(use_itn)[2]
    return np.concatenate([embed[rows].astype(F32), np.asarray(speech, F32)], axis=0)
