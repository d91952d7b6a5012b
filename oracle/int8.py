"""Oracle: the arithmetic of a dynamically quantised Linear layer, as the reference's DEFAULT models compute it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference's CLI default is
``-accuracy int8`` (``AliParaformerAsr.Examples/Program.cs:98-101``; ``model.int8.onnx``,
``Examples/OfflineAliParaformerAsrRecognizer.cs:17-22``) — FunASR exports passed through
``onnxruntime.quantization.quantize_dynamic(..., op_types_to_quantize=["MatMul"], per_channel=True,
weight_type=QuantType.QUInt8)``.  Neither onnxruntime (1.22.*, ``AliParaformerAsr.csproj:50``) nor a model file is
under /root/reference, so this file restates the published operator definitions the quantised graph consists of:

    DynamicQuantizeLinear-11 (ONNX):  uint8, range widened to include 0,
        y_scale = (max' - min') / 255,  y_zero_point = round_half_even(clamp(0 - min' / y_scale, 0, 255)),
        y = saturate(round_half_even(x / y_scale) + y_zero_point)           (MLAS: scale = 1 when max' == min')
    MatMulInteger-10:  int32 accumulation of (a - a_zp) * (b - b_zp), per-column b_zp
    Cast(int32 -> float), Mul(a_scale * b_scale[n]), Add(bias)               (= onnxruntime's MatMulIntegerToFloat)

and the weight side of quantize_dynamic (per output channel, uint8, asymmetric, range including 0) for the synthetic
models.  Every intermediate is float32 / exact integer, so the device kernels (csrc/k_quant.hip, csrc/k_gemm.hip
``gemm_i8_pp3``) can be — and are — compared bit for bit at the operator level (tests/test_gpu_int8.py).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _qparams(mn: np.ndarray, mx: np.ndarray):
    """scale, zero_point (both float32 arrays) from min / max; the range always includes 0."""
    mn = np.minimum(mn.astype(F32), F32(0))
    mx = np.maximum(mx.astype(F32), F32(0))
    rng = (mx - mn).astype(F32)
    scale = np.where(mx == mn, F32(1), (rng / F32(255)).astype(F32)).astype(F32)
    z = (F32(0) - (mn / scale).astype(F32)).astype(F32)
    zp = np.rint(np.clip(z, F32(0), F32(255))).astype(F32)              # round half to even
    return scale, zp


def _quantize(x: np.ndarray, scale, zp) -> np.ndarray:
    """uint8 values (as int32) of DynamicQuantizeLinear / quantize_dynamic's weight quantiser."""
    v = (np.rint((x.astype(F32) / scale).astype(F32)).astype(F32) + zp).astype(F32)
    return np.clip(v, F32(0), F32(255)).astype(np.int32)


def quantize_activation(x: np.ndarray):
    """DynamicQuantizeLinear over the WHOLE tensor: (q int32 in 0..255, scale float32, zero_point int)."""
    x = np.asarray(x, F32)
    scale, zp = _qparams(x.min() if x.size else F32(0), x.max() if x.size else F32(0))
    return _quantize(x, scale, zp), F32(scale), int(zp)


def quantize_weight(w: np.ndarray):
    """per output channel (row of W [N, K]): (q int32 [N, K], scale float32 [N], zero_point int32 [N])."""
    w = np.asarray(w, F32)
    scale, zp = _qparams(w.min(axis=1), w.max(axis=1))
    return _quantize(w, scale[:, None], zp[:, None]), scale.astype(F32), zp.astype(np.int32)


def qlinear(x: np.ndarray, wq: np.ndarray, wscale: np.ndarray, wzp: np.ndarray, bias=None) -> np.ndarray:
    """y = float32(sum_k (x_q - x_zp)(w_q - w_zp[n])) * (x_scale * w_scale[n]) (+ bias): x [..., K], wq [N, K]."""
    x = np.asarray(x, F32)
    xq, xs, xz = quantize_activation(x)
    a = (xq.reshape(-1, x.shape[-1]) - xz).astype(np.float64)            # integers: exact in float64 (|sum| < 2^53)
    b = (wq - wzp[:, None]).astype(np.float64)
    acc = np.rint(a @ b.T).astype(np.int64)
    assert np.abs(acc).max(initial=0) < 2 ** 31
    y = acc.astype(np.int32).astype(F32) * (xs * wscale.astype(F32)).astype(F32)[None, :]
    y = y.astype(F32)
    if bias is not None:
        y = (y + np.asarray(bias, F32)[None, :]).astype(F32)
    return y.reshape(x.shape[:-1] + (wq.shape[0],))


class QuantizedLinears:
    """Per tensor name for oracle.model.Oracle(quant="int8"): the stored bytes of an int8 export when the container
    carries them (`<name>.weight_q` / `.weight_zp` / `.weight_scale`, aliparaformerasr_amd/convert.py), otherwise
    quantize_weight() of the float tensor (the synthetic models)."""

    def __init__(self, weights: dict, exclude=()):
        self.w = weights
        self.cache = {}
        self.exclude = tuple(exclude)
        self.any_stored = any(k.endswith(".weight_q") for k in weights)

    def quantised(self, name: str) -> bool:
        """Is `<name>` a DynamicQuantizeLinear + MatMulInteger pair?  An export's container answers by the bytes it
        carries (FunASR runs quantize_dynamic with nodes_to_exclude: such MatMuls have no `.weight_q` and stay float);
        fp32-only weights quantise everything but the prefixes in the `int8_exclude` config key."""
        if self.any_stored:
            return name + ".weight_q" in self.w
        return not any(name.startswith(e) for e in self.exclude)

    def __call__(self, x, name: str, bias: bool):
        if name not in self.cache:
            if name + ".weight_q" in self.w:
                self.cache[name] = (self.w[name + ".weight_q"].astype(np.int32), self.w[name + ".weight_scale"].astype(F32),
                                    self.w[name + ".weight_zp"].astype(np.int32))
            else:
                self.cache[name] = quantize_weight(self.w[name + ".weight"])
        wq, ws, wz = self.cache[name]
        return qlinear(x, wq, ws, wz, self.w[name + ".bias"] if bias else None)
