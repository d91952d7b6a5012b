"""Per-kernel roofline table for BASELINE.json configs[1] (32 x 30 s) from a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py ...`) and the event-timed GEMM /
attention classes of `bench.py` (`class_ms_per_step`).  Algorithmic bytes / FLOPs are the config-2 figures of DESIGN.md §4.

usage: roofline_table.py <kernel_trace.csv> <bench_breakdown.json> > profiles/<name>.md"""
import collections, csv, json, sys

HBM_PEAK, HBM_ACH, MFMA_PEAK = 8.0, 6.3, 2500.0        # TB/s spec, TB/s achievable (guide), TFLOP/s dense f16
B, T, L, D, F, V = 32, 500, None, 512, 2048, 8404

def main():
    trace, bench = sys.argv[1], json.load(open(sys.argv[2]))
    L = bench["config"]["L"]
    M, Md = B * T, B * L
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        groups[(r["Kernel_Name"], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    def find(sub, grid=None):
        for (name, g), v in groups.items():
            if sub in name and (grid is None or g == grid):
                return sum(v) / len(v), len(v)
        return None, 0
    mem = [   # label, kernel substring, grid (threads), algorithmic MB per launch
        ("layernorm 512, encoder rows (fp32 in, f16 out)", "layernorm_kernelILi2ELb0", (M // 4) * 256, (M * D * 4 + M * D * 2) / 1e6),
        ("layernorm 512, decoder rows", "layernorm_kernelILi2ELb0", ((Md + 3) // 4) * 256, (Md * D * 4 + Md * D * 3) / 1e6),
        ("layernorm 2048, decoder FFN", "layernorm_kernelILi8ELb0", ((Md + 3) // 4) * 256, (Md * F * 4 + Md * F * 2) / 1e6),
        ("posenc + layernorm 560 (layer 0)", "layernorm_kernelILi3ELb1", None, (M * 560 * 4 + M * 576 * 2) / 1e6),
        ("FSMN encoder (f16 V in, fp32 out)", "fsmn_enc_kernel", None, (M * D * 2 + M * D * 4) / 1e6),
        ("FSMN decoder (fp32, read-modify-write)", "fsmn_dec_kernel", None, (Md * D * 4 * 3) / 1e6),
        ("fbank (audio in, 80 mel out)", "fbank_kernel", None, (B * 480000 * 4 + B * 3000 * 80 * 4) / 1e6),
        ("LFR + CMVN + pad", "lfr_cmvn_pad_kernel", None, (B * 3000 * 80 * 4 + M * 560 * 4) / 1e6),
        ("arg-max over the vocabulary", "argmax_kernel", None, (Md * V * 4) / 1e6),
        ("CIF im2col", "cif_im2col_kernel", None, (M * D * 2 + M * 3 * D * 2) / 1e6),
        ("CIF weighted gather", "cif_gather_kernel", None, (M * D * 4 + Md * D * 4) / 1e6),
    ]
    tag = next((a for a in sys.argv[3:] if not a.isdigit()), "round6")
    print("# Per-kernel roofline, paraformer-large 32 x 30 s on one MI355X (%s)\n" % tag)
    print("Source: `%s` (rocprofv3 --kernel-trace, average kernel duration) and the HIP-event class times of "
          "`bench.py` (`class_ms_per_step`).  L = %d.  Peaks: HBM 8 TB/s spec (6.3 TB/s achievable), dense f16 MFMA 2.5 PFLOP/s.\n" % (trace.split("/")[-1], L))
    print("## HBM-bound kernels\n\n| kernel | avg µs | launches/step | algorithmic MB | achieved TB/s | of 8 TB/s | of 6.3 TB/s |\n|---|---|---|---|---|---|---|")
    steps = next((int(a) for a in sys.argv[3:] if a.isdigit()), 11)     # forward passes in the traced command: warmup 2 + 1 profiling + 5 timed + 3 host-audio calls
    for label, sub, grid, mb in mem:
        us, n = find(sub, grid)
        if us is None:
            continue
        tbs = mb / us                        # MB/us = TB/s
        print("| %s | %.1f | %d | %.1f | %.2f | %.0f %% | %.0f %% |" % (label, us, round(n / steps), mb, tbs, 100 * tbs / HBM_PEAK, 100 * tbs / HBM_ACH))
    us, n = find("cif_scan_kernel")
    if us:
        print("| CIF integrate-and-fire scan (sequential, latency bound) | %.1f | %d | – | – | – | – |" % (us, round(n / steps)))
    print("\n## MFMA-bound kernel classes (HIP events around each launch; include ~3-4 µs of event overhead per launch)\n")
    print("| class | launches/step | ms/step | TFLOP/s | of 2.5 PFLOP/s |\n|---|---|---|---|---|")
    for cls, v in bench.get("class_ms_per_step", bench.get("kernel_breakdown_ms_per_step", {})).items():
        if v.get("tflops"):
            print("| %s | %d | %.3f | %.0f | %.0f %% |" % (cls, v["launches"], v["ms"], v["tflops"], 100 * v["tflops"] / MFMA_PEAK))
    r = bench["roofline"]
    print("\nDominant kernel (`bench.py` roofline object, timed inside the timed steps): %s — bound: %s, %.0f %s = %.3f of peak "
          "(%.3f of the MFMA peak, %.3f of the HBM peak; %.0f FLOP/B against a machine balance of %.0f); PMC traffic %s bytes/launch vs %d algorithmic.\n"
          % (r["kernel"], r["bound"], r["achieved"], r["unit"], r["frac"], r.get("frac_of_mfma_peak", 0), r.get("frac_of_hbm_peak", 0),
             r.get("intensity_flop_per_byte", 0), r.get("machine_balance_flop_per_byte", 0), r.get("traffic"), r.get("algorithmic_bytes_per_launch", 0)))
    print("Whole path: %.2f ms/step, RTFx %.0f, %.0f TFLOP/s algorithmic over the wall time.\n" % (bench["ms_per_step"], bench["value"], bench["whole_path_tflops_per_gpu"]))

main()
