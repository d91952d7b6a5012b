"""GPU idle time inside one steady step of bench.py (one step in flight), from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python bench.py --no-cpu-baseline --no-via-recognizer --steps 6 --warmup 2 --in-flight 1
    python tools/step_gaps.py /tmp/p/.../*_kernel_trace.csv
A step is found as the span between consecutive fbank_kernel launches; prints wall span, summed kernel time, idle time and
the largest gaps with the kernels on either side."""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
starts = [i for i, r in enumerate(rows) if "fbank_kernel" in r[2]]
for a, b in list(zip(starts[:-1], starts[1:]))[-4:-1]:
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = sorted(((seg[i + 1][0] - seg[i][1], seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1)), reverse=True)
    print("step: %d launches, span %.3f ms, kernels %.3f ms, idle %.3f ms (%.1f %%); median gap %.2f us" %
          (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3))
    for g in gaps[:6]:
        print("   gap %.1f us between %s -> %s" % (g[0] / 1e3, g[1], g[2]))
