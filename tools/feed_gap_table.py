"""Builds profiles/<TAG>_feed_gap.md from the counter tables of tools/feed_gap.sh (profiles/<TAG>_pmc_{sq1,sq2,tcp,tcc,grbm}.json) and
the kernel-trace averages (profiles/<TAG>_kernel_stats.csv).  usage: python tools/feed_gap_table.py round5 > profiles/round5_feed_gap.md
Derivations (per launch): SQ_* wave counters are in quad-cycles and relate to SQ_WAVE_CYCLES; SQ_VALU_MFMA_BUSY_CYCLES is in cycles
(32 per 32x32x16 MFMA) and relates to kernel cycles x 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = / 8);
TCP_PENDING_STALL_CYCLES relates to kernel cycles x 256 CUs."""
import csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "round5"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
T = {g: json.load(open(os.path.join(root, "%s_pmc_%s.json" % (tag, g)))) for g in ("sq1", "sq2", "tcp", "tcc", "grbm")}
avg = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(root, "%s_kernel_stats.csv" % tag)))}
ROWS = [
    ("ffn_fused_kernel<8, 0, 2, 1, 1, 0>", "encoder layer tail, ONE launch: out-projection + FSMN + norm2 + FFN + next norm1 + next Q|K|V (k_ffn.hip)"),
    ("attn_kernel<8, 2>", "self-attention (k_attn.hip, 8 waves x 32 queries, Q|K blocked)"),
    ("ffn_fused_kernel<8, 0, 2, 2, 0, 3>", "decoder: previous layer's out-projection + norm1 + FFN block, split form, 3 shares per 64-row tile (k_ffn.hip)"),
    ("dec_mid_kernel", "decoder (round 6): finishing pass of the split FFN + norm2 + FSMN memory + residual + norm3 + q-projection, one launch (k_decmid.hip)"),
    ("ffn_dec_finish_kernel", "decoder: shares summed, hidden LayerNorm applied from row statistics, norm2 (final block only since round 6)"),
    ("attn_kernel<4, 2>", "cross-attention (decoder, 4 waves x 32 queries)"),
    ("gemm_f16_pp3<1, 2>", "K / V projection of all 16 decoder layers (one GEMM, N = 16384)"),
]


def find(table, key):
    for k in table:
        if key in k:
            return table[k]
    return None


print("# Where the shipped kernels wait — hardware counters at HEAD (%s)\n" % tag)
print("`tools/feed_gap.sh`: one `rocprofv3 --kernel-trace --pmc <group>` pass per counter group over `bench.py --steps 1 --warmup 1 --in-flight 1` (never")
print("combined with tracing domains), averaged per launch by `tools/pmc_counters.py`; this table: `tools/feed_gap_table.py %s`.  Raw tables:" % tag)
print("`profiles/%s_pmc_{sq1,sq2,tcp,tcc,grbm}.json`.\n" % tag)
print("| kernel | avg µs (trace) | issuing % | parked on s_waitcnt / barrier % | stalled at issue % | of which LDS issue % | MFMA pipe busy % of kernel cycles | L2 hit % | TCP waiting on L2 % of kernel cycles x CUs | LDS bank conflicts |")
print("|---|---|---|---|---|---|---|---|---|---|")
for key, what in ROWS:
    a, b, c, d, g = (find(T[x], key) for x in ("sq1", "sq2", "tcp", "tcc", "grbm"))
    if not a or not g:
        continue
    us = next((v for k, v in avg.items() if key in k), float("nan"))
    wc = a["SQ_WAVE_CYCLES"]
    cyc = g["GRBM_GUI_ACTIVE"] / 8.0
    print("| `%s` — %s | %.1f | %.0f | %.0f | %.0f | %.1f | %.0f | %.0f | %.0f | %d |" % (
        key, what, us, 100 * a["SQ_ACTIVE_INST_ANY"] / wc, 100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_WAIT_INST_LDS"] / wc,
        100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 100 * d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1),
        100 * c["TCP_PENDING_STALL_CYCLES_sum"] / (cyc * 256), round(b["SQ_LDS_BANK_CONFLICT"])))
