"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced stream (guide, HBM section): doubled.  WRITE_SIZE was calibrated against a known output size in round 1
(exact).  usage: pmc_summary.py fetch.csv write.csv > pmc.json"""
import collections, csv, json, re, sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        acc[(name, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"notes": __doc__.strip().split("usage")[0].strip(), "kernels": []}
for key in sorted(fetch, key=lambda k: -sum(fetch[k])):
    f, w = fetch[key], write.get(key, [])
    if len(f) < 5:
        continue
    rd = 2.0 * 1024 * sum(f) / len(f)
    wr = 1024 * sum(w) / len(w) if w else 0.0
    out["kernels"].append({"kernel": key[0], "grid": key[1], "launches": len(f), "hbm_read_bytes_per_launch_corrected": rd,
                           "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr})
print(json.dumps(out, indent=1))
