"""Average per-launch value of arbitrary rocprofv3 PMC counters per kernel.
usage: pmc_counters.py counter_collection.csv [min_launches] > table.json"""
import collections, csv, json, re, sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
minl = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = {}
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    if n < minl:
        continue
    out[k] = {"launches": n, **{c: sum(v) / len(v) for c, v in cs.items()}}
print(json.dumps(out, indent=1))
