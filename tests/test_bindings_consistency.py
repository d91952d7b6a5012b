"""CPU: the three statements of the C ABI must agree — `include/paraformer_hip.h` (the contract), the C# P/Invoke text a
maintainer of the reference would add (`csharp/*.cs`; never compiled here: no .NET toolchain in the image) and the
library's export table.  Mechanical checks only: every `static extern` names a function the header declares, with the
same number of parameters and a compatible type class per parameter (pointer / 32-bit / 64-bit / float / double);
`PfEngineConfig` / `PfBatchOut` list the header structs' fields in order with matching widths; the status constants
match; every declared function is exported by the built library."""
import ctypes as C
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def _split_params(p):
    out, depth, cur = [], 0, ""
    for ch in p:
        if ch in "([<":
            depth += 1
        if ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _c_class(t):
    t = t.strip()
    if "*" in t:
        return "ptr"
    base = t.split()[:-1] if len(t.split()) > 1 else t.split()
    b = " ".join(x for x in base if x != "const")
    return {"int32_t": "i32", "int": "i32", "uint32_t": "i32", "int64_t": "i64", "float": "f32", "double": "f64"}.get(b, b)


def _cs_class(t):
    t = re.sub(r"^(\[[A-Za-z0-9_.()= ,]+\]\s*)+", "", t.strip()).strip()  # leading attributes: [Out], [MarshalAs(...)]
    if t.startswith(("ref ", "out ")) or "[]" in t.split()[0] or t.split()[0].rstrip("?") in ("IntPtr", "string"):
        return "ptr"
    return {"int": "i32", "uint": "i32", "long": "i64", "float": "f32", "double": "f64"}.get(t.split()[0], t.split()[0])


def _header_functions():
    src = _strip_comments(open(os.path.join(ROOT, "include", "paraformer_hip.h")).read())
    fns = {}
    for m in re.finditer(r"\b(?:int|void|const char\*|pf_engine\*)\s+(pf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ps = [] if m.group(2).strip() in ("", "void") else _split_params(" ".join(m.group(2).split()))
        fns[m.group(1)] = [_c_class(x) for x in ps]
    return src, fns


def test_csharp_externs_match_the_header():
    _, fns = _header_functions()
    assert len(fns) > 80
    seen = 0
    for path in glob.glob(os.path.join(ROOT, "csharp", "*.cs")):
        src = _strip_comments(open(path, encoding="utf-8-sig").read())
        for m in re.finditer(r"static\s+extern\s+(\w+)\s+(pf_[a-z0-9_]+)\s*\((.*?)\)\s*;", src, flags=re.S):
            name, params = m.group(2), _split_params(" ".join(m.group(3).split()))
            assert name in fns, "%s: %s is not declared in include/paraformer_hip.h" % (os.path.basename(path), name)
            got = [_cs_class(p) for p in params]
            assert got == fns[name], "%s: %s%s in C#, %s in the header" % (os.path.basename(path), name, got, fns[name])
            seen += 1
    assert seen >= 45
    # what a C# caller of the reference's classes needs is declared; parity / test entry points need not be
    have = set()
    for path in glob.glob(os.path.join(ROOT, "csharp", "*.cs")):
        have |= set(re.findall(r"static\s+extern\s+\w+\s+(pf_[a-z0-9_]+)", open(path, encoding="utf-8-sig").read()))
    optional = {"pf_engine_set_hotwords", "pf_last_flops", "pf_run_staged", "pf_stage_audio", "pf_sync", "pf_online_encoder",
                "pf_online_decoder", "pf_online_stream_tokens", "pf_group_engine", "pf_online_recognizer_engine", "pf_recognizer_engine"}
    missing = {f for f in fns if not f.startswith(("pf_op_", "pf_host_", "pf_decoded_"))} - have - optional
    assert not missing, missing


def _struct_fields_c(src, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(.*?)(\w+)(\[(\d+)\])?$", decl)
        cls = _c_class(m.group(1) + " x")
        out += [(m.group(2), cls)] * (int(m.group(4)) if m.group(4) else 1)
    return out


def _struct_fields_cs(name):
    src = _strip_comments(re.sub(r"///[^\n]*", "", open(os.path.join(ROOT, "csharp", "ParaformerHip.cs"), encoding="utf-8-sig").read()))
    body = re.search(r"struct %s\s*\{(.*?)\}" % name, src, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"public (\w+) (.*)$", decl)
        cls = {"int": "i32", "long": "i64", "float": "f32", "IntPtr": "ptr"}[m.group(1)]
        out += [(n.strip(), cls) for n in m.group(2).split(",")]
    return out


def test_csharp_structs_match_the_header():
    src, _ = _header_functions()
    for c_name, cs_name in (("pf_engine_config", "PfEngineConfig"), ("pf_batch_out", "PfBatchOut")):
        c, cs = _struct_fields_c(src, c_name), _struct_fields_cs(cs_name)
        assert [k for _, k in c] == [k for _, k in cs], (c_name, c, cs)
        for (cn, _), (sn, _) in zip(c, cs):
            assert sn == cn or sn.rstrip("0123456789") == cn, (c_name, cn, sn)


def test_status_codes_match():
    src, _ = _header_functions()
    c = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"(PF_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", src))
    cs_src = open(os.path.join(ROOT, "csharp", "ParaformerHip.cs"), encoding="utf-8-sig").read()
    cs = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"(PF_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", cs_src))
    assert c and c == cs
    from aliparaformerasr_amd import _native as N
    for k, v in c.items():
        assert getattr(N, k) == v, k
    abi = int(re.search(r"#define PF_ABI_VERSION (\d+)", src).group(1))
    assert ("PF_ABI_VERSION = %d" % abi) in cs_src or ("AbiVersion = %d" % abi) in cs_src or re.search(r"\b%d\b" % abi, cs_src)


def test_library_exports_every_declared_function():
    _, fns = _header_functions()
    from aliparaformerasr_amd import _native as N
    lib = N.load()
    for name in fns:
        assert hasattr(lib, name), name
    assert lib.pf_version() == int(re.search(r"#define PF_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "paraformer_hip.h")).read()).group(1))


def _ct_class(t):
    if t in (C.c_int32, C.c_int, C.c_uint32):
        return "i32"
    if t in (C.c_int64,):
        return "i64"
    if t is C.c_float:
        return "f32"
    if t is C.c_double:
        return "f64"
    return "ptr"                                    # POINTER(...), c_void_p, c_char_p


def test_ctypes_signatures_match_the_header():
    """aliparaformerasr_amd/_native.py SIGNATURES (what every Python test and the bench call through) against the header:
    same functions, same parameter count, same type class per parameter (a c_int32 where the ABI takes int64_t
    truncates silently on x86-64)."""
    _, fns = _header_functions()
    from aliparaformerasr_amd import _native as N
    assert set(N.SIGNATURES) == set(fns), (set(N.SIGNATURES) ^ set(fns))
    for name, (_res, args) in N.SIGNATURES.items():
        got = [_ct_class(a) for a in args]
        assert got == fns[name], "%s: ctypes %s, header %s" % (name, got, fns[name])
    for cname, cls in (("pf_engine_config", N.PfEngineConfig), ("pf_batch_out", N.PfBatchOut)):
        src, _ = _header_functions()
        want = _struct_fields_c(src, cname)
        got = []
        for fname, ftype in cls._fields_:
            n = getattr(ftype, "_length_", None)
            base = ftype._type_ if n and not hasattr(ftype, "contents") else ftype
            got += [(fname, _ct_class(base))] * (n if n and not hasattr(ftype, "contents") else 1)
        assert [k for _, k in got] == [k for _, k in want], (cname, got, want)


def _cs_public_members(cls):
    """public constructors (parameter counts), methods and properties of `class cls` in csharp/*.cs"""
    for path in glob.glob(os.path.join(ROOT, "csharp", "*.cs")):
        src = _strip_comments(re.sub(r"///[^\n]*", "", open(path, encoding="utf-8-sig").read()))
        m = re.search(r"public\s+(?:sealed\s+)?class\s+%s\b[^{]*\{" % cls, src)
        if not m:
            continue
        depth, i = 1, m.end()
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        body = src[m.end():i]
        out = {"constructors": [], "methods": set(), "properties": set()}
        for c in re.finditer(r"^\s*public\s+%s\s*\(([^)]*)\)" % cls, body, flags=re.M):
            out["constructors"].append(len([p for p in _split_params(c.group(1)) if p.strip()]))
        for c in re.finditer(r"^\s*public\s+(?:static\s+|virtual\s+|override\s+)*([\w<>\[\]\?,\.]+(?:\s*<[^>]*>)?\??)\s+(\w+)\s*(\(|\{|=>|;|\s*$)",
                             body, flags=re.M):
            (out["methods"] if c.group(3) == "(" else out["properties"]).add(c.group(2))
        return out
    raise AssertionError("class %s not found under csharp/" % cls)


def test_public_surface_of_the_mirrored_classes_covers_the_reference():
    """"identical public signatures"  checked mechanically — the public member names of the
    reference's OfflineRecognizer / OfflineStream (tests/golden/reference_public_api.json, written from the reference's
    sources by tests/golden/make_public_api.py) must all exist on the C# drop-in classes AND on the Python mirror; the
    constructors must accept the reference's argument counts (the drop-in's extra `device` argument is optional)."""
    import json
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_public_api.json")))
    from aliparaformerasr_amd import offline_recognizer as O, online_recognizer as ON
    py = {"OfflineStream": O.OfflineStream, "OfflineRecognizer": O.OfflineRecognizer,
          "OnlineStream": ON.OnlineStream, "OnlineRecognizer": ON.OnlineRecognizer}
    # the streaming classes (a "next" row): members only OnlineRecognizer.Forward itself touches — the
    # chunk queue and the carried CIF / FSMN state live behind the C ABI (csrc/online.cpp) and are not exposed
    internal_only = {"OnlineStream": {"GetDecodeChunk", "InputSpeech", "IsFinished", "RemoveChunk", "CifAlpha", "CifHidden", "Hyp",
                                      "OnlineInputEntity", "States", "Timestamps", "Tokens"}}
    for cls, want in api.items():
        got = _cs_public_members(cls)
        skip = internal_only.get(cls, set())
        for name in set(want["methods"]) | set(want["properties"]):
            if name in skip:
                continue
            assert name in got["methods"] | got["properties"], "csharp: %s.%s is public in the reference and missing here" % (cls, name)
            assert hasattr(py[cls], name), "python: %s.%s missing" % (cls, name)
        if cls.startswith("Offline"):
            assert not skip
            for n in want["constructors"]:
                assert any(n <= k <= n + 1 for k in got["constructors"]), (cls, n, got["constructors"])
