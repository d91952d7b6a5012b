"""Minimal ONNX (protobuf) reader / writer — no `onnx` package exists in the build image.

Only what the converters need: ModelProto.graph -> nodes (op_type, name, inputs, outputs, int/ints/string attributes)
and initializers (float32, float16, int8, uint8, int32, int64; raw_data or the typed repeated fields).  Field numbers
follow onnx.proto3: ModelProto.graph = 7; GraphProto.node = 1, .name = 2, .initializer = 5, .input = 11, .output = 12;
NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5; AttributeProto.name = 1, .f = 2, .i = 3,
.s = 4, .ints = 8, .type = 20; TensorProto.dims = 1, .data_type = 2, .float_data = 4, .int32_data = 5,
.int64_data = 7, .name = 8, .raw_data = 9; ValueInfoProto.name = 1.
The writer emits the same subset and exists so that tests can build synthetic graphs."""
from __future__ import annotations

import struct

import numpy as np

_DT = {1: "<f4", 2: "u1", 3: "i1", 6: "<i4", 7: "<i8", 10: "<f2", 11: "<f8"}
_DT_REV = {np.dtype("float32"): 1, np.dtype("uint8"): 2, np.dtype("int8"): 3, np.dtype("int32"): 6,
           np.dtype("int64"): 7, np.dtype("float16"): 10}


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
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, v


def _packed_varints(v):
    out, j = [], 0
    while j < len(v):
        d, j = _varint(v, j)
        out.append(d)
    return out


def _signed(v, bits=64):
    return v - (1 << bits) if v >= 1 << (bits - 1) else v


def parse_tensor(buf):
    dims, dtype, name, raw = [], None, "", None
    f32, i32, i64 = [], [], []
    for f, w, v in _fields(buf):
        if f == 1:
            dims += _packed_varints(v) if w == 2 else [v]
        elif f == 2:
            dtype = v
        elif f == 4:
            f32 += list(np.frombuffer(v, "<f4")) if w == 2 else [struct.unpack("<f", v)[0]]
        elif f == 5:
            i32 += [_signed(x, 64) for x in _packed_varints(v)] if w == 2 else [_signed(v, 64)]
        elif f == 7:
            i64 += [_signed(x, 64) for x in _packed_varints(v)] if w == 2 else [_signed(v, 64)]
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = v
    if dtype not in _DT:
        raise ValueError("tensor '%s': unsupported ONNX data_type %s" % (name, dtype))
    if raw is not None:
        arr = np.frombuffer(raw, _DT[dtype])
    elif dtype == 1:
        arr = np.asarray(f32, "<f4")
    elif dtype == 7:
        arr = np.asarray(i64, "<i8")
    else:
        arr = np.asarray(i32).astype(_DT[dtype])          # int8/uint8/int32/float16 bit patterns live in int32_data
        if dtype == 10:
            arr = np.asarray(i32, np.uint16).view("<f2")
    return name, arr.reshape(dims).copy()


class Node:
    __slots__ = ("op_type", "name", "inputs", "outputs", "attrs")

    def __init__(self, op_type, name, inputs, outputs, attrs):
        self.op_type, self.name, self.inputs, self.outputs, self.attrs = op_type, name, inputs, outputs, attrs

    def __repr__(self):
        return "Node(%s %s %s -> %s)" % (self.op_type, self.name, self.inputs, self.outputs)


def _parse_node(buf):
    ins, outs, name, op, attrs = [], [], "", "", {}
    for f, w, v in _fields(buf):
        if f == 1:
            ins.append(v.decode())
        elif f == 2:
            outs.append(v.decode())
        elif f == 3:
            name = v.decode()
        elif f == 4:
            op = v.decode()
        elif f == 5:
            an, val = "", None
            for ff, ww, vv in _fields(v):
                if ff == 1:
                    an = vv.decode()
                elif ff == 3:
                    val = _signed(vv)
                elif ff == 2:
                    val = struct.unpack("<f", vv)[0]
                elif ff == 4:
                    val = vv.decode(errors="replace")
                elif ff == 8:
                    val = (val or []) + ([_signed(x) for x in _packed_varints(vv)] if ww == 2 else [_signed(vv)])
            attrs[an] = val
    return Node(op, name, ins, outs, attrs)


class Graph:
    def __init__(self):
        self.nodes, self.initializers, self.inputs, self.outputs, self.name = [], {}, [], [], ""

    def consumers(self, value_name):
        return [n for n in self.nodes if value_name in n.inputs]

    def producer(self, value_name):
        for n in self.nodes:
            if value_name in n.outputs:
                return n
        return None


def load(path_or_bytes) -> Graph:
    data = path_or_bytes
    if isinstance(path_or_bytes, str):
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    gbuf = None
    for f, w, v in _fields(data):
        if f == 7 and w == 2:
            gbuf = v
    if gbuf is None:
        raise ValueError("not an ONNX ModelProto (no graph)")
    g = Graph()
    for f, w, v in _fields(gbuf):
        if f == 1:
            g.nodes.append(_parse_node(v))
        elif f == 2:
            g.name = v.decode()
        elif f == 5:
            name, arr = parse_tensor(v)
            g.initializers[name] = arr
        elif f in (11, 12):
            nm = next((vv.decode() for ff, ww, vv in _fields(v) if ff == 1), "")
            (g.inputs if f == 11 else g.outputs).append(nm)
    g.inputs = [n for n in g.inputs if n not in g.initializers]
    return g


# ------------------------------------------------------------------ writer (tests) ---------
def _vi(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fno, payload):
    return _vi((fno << 3) | 2) + _vi(len(payload)) + payload


def _tensor(name, arr):
    arr = np.ascontiguousarray(arr)
    body = b"".join(_vi((1 << 3) | 0) + _vi(d) for d in arr.shape)
    body += _vi((2 << 3) | 0) + _vi(_DT_REV[arr.dtype])
    body += _ld(8, name.encode()) + _ld(9, arr.tobytes())
    return body


def _node(op, name, ins, outs, attrs=None):
    body = b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs)
    body += _ld(3, name.encode()) + _ld(4, op.encode())
    for k, v in (attrs or {}).items():
        a = _ld(1, k.encode())
        if v is None:
            continue                                   # (tensor / graph attributes the reader does not keep)
        if isinstance(v, str):
            a += _ld(4, v.encode()) + _vi((20 << 3) | 0) + _vi(3)
        elif isinstance(v, float):
            a += _vi((2 << 3) | 5) + struct.pack("<f", v) + _vi((20 << 3) | 0) + _vi(1)
        elif isinstance(v, (list, tuple)):
            a += b"".join(_vi((8 << 3) | 0) + _vi(x) for x in v) + _vi((20 << 3) | 0) + _vi(7)
        else:
            a += _vi((3 << 3) | 0) + _vi(int(v)) + _vi((20 << 3) | 0) + _vi(2)
        body += _ld(5, a)
    return body


def dump(nodes, initializers, inputs, outputs, name="g") -> bytes:
    """nodes: [(op_type, name, inputs, outputs, attrs)], initializers: {name: ndarray}."""
    g = b"".join(_ld(1, _node(*n)) for n in nodes) + _ld(2, name.encode())
    g += b"".join(_ld(5, _tensor(k, v)) for k, v in initializers.items())
    g += b"".join(_ld(11, _ld(1, i.encode())) for i in inputs) + b"".join(_ld(12, _ld(1, o.encode())) for o in outputs)
    return _vi((1 << 3) | 0) + _vi(8) + _ld(7, g)          # ir_version = 8, graph
