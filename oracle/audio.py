"""Oracle restatement of the Examples harness audio helpers (test infrastructure only).

Reference: AliParaformerAsr.Examples/Utils/AudioHelper.cs — GetFileSample (:12-32), Resample with channel
down-mix (:223-279), IsAudioByHeader / IsWavHeader (:286-340).  NAudio's AudioFileReader (external NuGet
package) converts every PCM width to IEEE float; the conversions restated here are NAudio's published
sample providers: PCM8 b/128-1, PCM16 /32768, PCM24 /8388608, PCM32 /2147483648, float32 unchanged.  G.711 files
(WAVE_FORMAT_ALAW 6 / MULAW 7, 8 bits) reach AudioFileReader through a codec that expands them to 16-bit PCM first: the ITU-T
G.711 tables (restated as their closed forms below), then /32768; IEEE float64 narrows to float32."""
import struct

import numpy as np

F32 = np.float32


def decode_wav(data: bytes):
    """-> (interleaved float32 samples, sample_rate, channels, duration_ms)"""
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, sz = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8: pos + 8 + sz]
        if cid == b"fmt ":
            tag, ch, sr, _br, align, bits = struct.unpack_from("<HHIIHH", body, 0)
            if tag == 0xFFFE and sz >= 26:
                tag = struct.unpack_from("<H", body, 24)[0]
            fmt = (tag, ch, sr, align, bits)
        elif cid == b"data":
            payload = body
            break
        pos += 8 + sz + (sz & 1)
    tag, ch, sr, align, bits = fmt
    if tag == 7 and bits == 8:                       # mu-law (G.711): ~byte = sign | exponent (3) | mantissa (4)
        u = (~np.frombuffer(payload, np.uint8)).astype(np.int32) & 0xFF
        mag = ((((u & 0x0F) << 3) + 0x84) << ((u >> 4) & 7)) - 0x84
        x = (np.where(u & 0x80, -mag, mag).astype(F32) / F32(32768.0)).astype(F32)
    elif tag == 6 and bits == 8:                     # A-law (G.711): byte ^ 0x55, sign bit SET = positive
        a = np.frombuffer(payload, np.uint8).astype(np.int32) ^ 0x55
        e, m = (a >> 4) & 7, a & 0x0F
        mag = np.where(e == 0, (m << 4) + 8, ((m << 4) + 0x108) << np.maximum(e - 1, 0))
        x = (np.where(a & 0x80, mag, -mag).astype(F32) / F32(32768.0)).astype(F32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(payload[: len(payload) // 8 * 8], "<f8").astype(F32)
    elif tag == 3:
        x = np.frombuffer(payload[: len(payload) // 4 * 4], "<f4").astype(F32)
    elif bits == 16:
        x = (np.frombuffer(payload[: len(payload) // 2 * 2], "<i2").astype(F32) / F32(32768.0)).astype(F32)
    elif bits == 24:
        b = np.frombuffer(payload[: len(payload) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = (v.astype(F32) / F32(8388608.0)).astype(F32)
    elif bits == 32:
        x = (np.frombuffer(payload[: len(payload) // 4 * 4], "<i4").astype(F32) / F32(2147483648.0)).astype(F32)
    else:
        x = (np.frombuffer(payload, np.uint8).astype(F32) / F32(128.0) - F32(1.0)).astype(F32)
    return x, sr, ch, (len(payload) // align) * 1000.0 / sr


def resample(source, sr_in, sr_out, channels=1):
    """AudioHelper.Resample (:223-279), float64 interpolation, Math.Round (half to even) target length."""
    src = np.asarray(source, F32)
    if src.size == 0:
        return np.zeros(0, F32)
    if channels == 2:
        n = src.size // 2
        src = ((src[0:2 * n:2] + src[1:2 * n:2]) * F32(0.5)).astype(F32)
    ratio = float(sr_in) / float(sr_out)
    n_out = int(np.round(src.size / ratio))          # numpy rounds half to even, like Math.Round
    out = np.zeros(n_out, F32)
    for i in range(n_out):
        pos = i * ratio
        idx = int(pos)
        fr = pos - idx
        if idx >= src.size - 1:
            out[i] = src[-1]
        else:
            out[i] = F32((1 - fr) * float(src[idx]) + fr * float(src[idx + 1]))
    return out


def get_file_sample(data):
    """GetFileSample (:12-32): None (missing file) -> float[1]; resample/down-mix only when rate != 16000."""
    if data is None:
        return np.zeros(1, F32), 0.0
    x, sr, ch, dur = decode_wav(data)
    if sr != 16000:
        x = resample(x, sr, 16000, ch)
    return x, dur
