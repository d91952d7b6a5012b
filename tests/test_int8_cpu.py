"""CPU: oracle/int8.py against the known-answer vectors of the ONNX operator it restates.

DynamicQuantizeLinear-11 is what onnxruntime executes in front of every MatMulInteger of the reference's default
`model.int8.onnx` (Examples/Program.cs:98-101).  Its ONNX backend test cases (onnx/backend/test/case/node/
dynamicquantizelinear.py: "expanded", "max_adjusted", "min_adjusted") publish inputs and the expected uint8 outputs,
scale and zero point — the only reference-side golden vectors that exist for any arithmetic on this path."""
import numpy as np

from oracle import int8 as q8


def test_dynamicquantizelinear_onnx_vectors():
    # test_dynamicquantizelinear
    x = np.array([0, 2, -3, -2.5, 1.34, 0.5], np.float32)
    q, s, z = q8.quantize_activation(x)
    assert z == 153 and np.float32(s) == np.float32(5.0) / np.float32(255)
    np.testing.assert_array_equal(q, [153, 255, 0, 26, 221, 179])
    # test_dynamicquantizelinear_max_adjusted: all negative -> max becomes 0
    x = np.array([-1.0, -2.1, -1.3, -2.5, -3.34, -4.0], np.float32)
    q, s, z = q8.quantize_activation(x)
    assert z == 255 and np.float32(s) == np.float32(4.0) / np.float32(255)
    np.testing.assert_array_equal(q, [191, 121, 172, 96, 42, 0])
    # test_dynamicquantizelinear_min_adjusted: all positive -> min becomes 0
    x = np.array([1, 2.1, 1.3, 2.5, 3.34, 4.0, 1.5, 2.6, 3.9, 4.0, 3.0, 2.345], np.float32).reshape(3, 4)
    q, s, z = q8.quantize_activation(x)
    assert z == 0 and np.float32(s) == np.float32(4.0) / np.float32(255)
    np.testing.assert_array_equal(q.reshape(-1), [64, 134, 83, 159, 213, 255, 96, 166, 249, 255, 191, 149])


def test_round_half_to_even_and_saturation():
    # range 255 -> scale exactly 1, zero point 100: values on a .5 boundary round to the even neighbour
    x = np.array([-100.0, 155.0, 0.5, 1.5, 2.5, -0.5, 154.5], np.float32)
    q, s, z = q8.quantize_activation(x)
    assert s == 1.0 and z == 100
    assert list(q) == [0, 255, 100, 102, 102, 100, 254]
    # float32 arithmetic of the zero point: 1 / (2 / 255) = 127.49999 in float32 -> 127, not 128
    q, s, z = q8.quantize_activation(np.array([-1.0, 1.0, 0.0], np.float32))
    assert z == 127 and list(q) == [0, 254, 127]
    q, s, z = q8.quantize_activation(np.zeros(5, np.float32))
    assert s == 1.0 and z == 0 and not q.any()          # MLAS: a zero range gives scale 1


def test_qlinear_is_matmulinteger_plus_rescale():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((7, 24)).astype(np.float32)
    w = rng.standard_normal((5, 24)).astype(np.float32)
    b = rng.standard_normal(5).astype(np.float32)
    wq, ws, wz = q8.quantize_weight(w)
    assert wq.min() >= 0 and wq.max() <= 255 and ws.shape == (5,) and wz.shape == (5,)
    # per channel: the extremes of every row hit 0 and 255 unless the row's range had to be widened to include 0
    for n in range(5):
        assert wq[n].min() == 0 or w[n].min() > 0
        assert wq[n].max() == 255 or w[n].max() < 0
    y = q8.qlinear(x, wq, ws, wz, b)
    xq, xs, xz = q8.quantize_activation(x)
    ref = np.zeros((7, 5), np.float32)
    for m in range(7):
        for n in range(5):
            acc = int(((xq[m] - xz) * (wq[n] - wz[n])).sum())
            ref[m, n] = np.float32(np.float32(acc) * np.float32(xs * ws[n])) + b[n]
    np.testing.assert_array_equal(y, ref)
    assert np.abs(y - (x @ w.T + b)).max() < 0.15


def test_matmulinteger_onnx_vector_through_qlinear():
    """The ONNX backend test of MatMulInteger (onnx/backend/test/case/node/matmulinteger.py): A uint8 [4, 3] with
    a_zero_point 12, B uint8 [3, 2] with b_zero_point 0 -> Y int32.  Fed through the oracle's public Linear: float rows
    x = A - 12 plus one row that pins the dynamic range to [-12, 243], so that DynamicQuantizeLinear returns scale 1, zero
    point 12 and the codes A themselves; unit weight scales then make y the published integers."""
    A = np.array([[11, 7, 3], [10, 6, 2], [9, 5, 1], [8, 4, 0]], np.int32)
    B = np.array([[1, 4], [2, 5], [3, 6]], np.int32)
    Y = np.array([[-38, -83], [-44, -98], [-50, -113], [-56, -128]], np.int32)
    x = np.concatenate([A - 12, np.array([[243, -12, 0]], np.int32)]).astype(np.float32)
    xq, xs, xz = q8.quantize_activation(x)
    assert xs == 1.0 and xz == 12
    np.testing.assert_array_equal(xq[:4], A)
    y = q8.qlinear(x, B.T.copy(), np.ones(2, np.float32), np.zeros(2, np.int32))
    np.testing.assert_array_equal(y[:4], Y.astype(np.float32))
