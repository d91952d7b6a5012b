"""Writes tests/golden/reference_public_api.json: the PUBLIC member names (constructors, methods, properties) of the
reference classes the drop-in mirrors, taken from the reference's own sources.  Names only — data, not source text.
Run in the build container (needs /root/reference); the GPU box and the tests read the JSON.

    python tests/golden/make_public_api.py
"""
import json
import os
import re

REF = "/root/reference/AliParaformerAsr"
CLASSES = {
    "OfflineStream": "OfflineStream.cs",
    "OfflineRecognizer": "OfflineRecognizer.cs",
    "OnlineStream": "OnlineStream.cs",
    "OnlineRecognizer": "OnlineRecognizer.cs",
}


def public_members(path, cls):
    src = open(path, encoding="utf-8-sig").read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {"constructors": [], "methods": [], "properties": []}
    for m in re.finditer(r"^\s*public\s+%s\s*\(([^)]*)\)" % cls, src, flags=re.M):
        out["constructors"].append(len([p for p in m.group(1).split(",") if p.strip()]))
    for m in re.finditer(r"^\s*public\s+(?!class\b)(?:static\s+|virtual\s+|override\s+)*([\w<>\[\]\?,\.]+(?:\s*<[^>]*>)?\??)\s+(\w+)\s*(\(|\{|=>|;)", src, flags=re.M):
        name, paren = m.group(2), m.group(3)
        (out["methods"] if paren == "(" else out["properties"]).append(name)
    for k in out:
        out[k] = sorted(set(out[k])) if k != "constructors" else sorted(out[k])
    return out


if __name__ == "__main__":
    api = {cls: public_members(os.path.join(REF, f), cls) for cls, f in CLASSES.items()}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_public_api.json")
    json.dump(api, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(api, indent=1, sort_keys=True))
