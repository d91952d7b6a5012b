"""CPU: the host-side helpers of bench.py that shape the numbers the driver reads — the quota-aware core count behind
`cpu_baseline.cores`, the roofline object's choice of the binding roof, and `golden_check` for a caller that has ids only
(the OfflineRecognizer API does not hand out token_num)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_usable_cores_is_capped_by_affinity_and_quota():
    n = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            assert n <= max(1, int(float(q) / float(per) + 0.5))
    except OSError:
        pass


def test_roofline_bound_follows_the_kernels_own_intensity():
    # round 4's row-complete FFN-down: 33.55 GFLOP over 149.6 MB = 224 FLOP/B < 312 -> HBM-bound by its own numbers
    r = bench.roofline_object("gemm_rc_kernel<0, 0>", "gemm_ffn2", 16000, 512, 2048, "x", 33.55e9, 149.6e6, 52.7e-6, 50, False, None)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - 149.6e6 / 52.7e-6 / 8e12) < 1e-9
    assert abs(r["frac_of_mfma_peak"] - 33.55e9 / 52.7e-6 / 2.5e15) < 1e-9 and r["frac"] == r["frac_of_hbm_peak"]
    # round 5's fused launch: 100.7 GFLOP over 153.5 MB = 656 FLOP/B -> MFMA-bound
    r = bench.roofline_object("ffn_fused_kernel<8, 0, 2, 1, 1>", "gemm_outffn", 16000, 2048, 512, "x", 100.7e9, 153.5e6, 126e-6, 50, False, 204.5e6)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - 100.7e9 / 126e-6 / 2.5e15) < 1e-9
    assert r["traffic"] == 204.5e6 and r["intensity_flop_per_byte"] > r["machine_balance_flop_per_byte"]


def test_golden_check_with_ids_only_skips_floor_near_ties():
    g = np.load(os.path.join(ROOT, "tests", "golden", "bench_paraformer.npz"))
    ids = g["ids"].copy()
    full = bench.golden_check("paraformer", ids, g["token_num"])
    assert full["ok"] and full["agree_all_positions"] == 1.0 and full["token_num_near_ties_resolved_differently"] == 0
    only = bench.golden_check("paraformer", ids)
    assert only["ok"] and only["near_tie_rows_skipped_without_token_num"] > 0
    # a wrong id on a decisive position of a NON-near-tie utterance must fail both forms
    frac = g["alpha_sum"] - np.floor(g["alpha_sum"])
    far = np.nonzero(np.minimum(frac, 1 - frac) >= bench.ALPHA_NEAR)[0]
    b = int(far[0])
    pos = int(np.argmax((g["margin"][b] > 0.5) & (np.arange(ids.shape[1]) < g["token_num"][b])))
    bad = ids.copy()
    bad[b, pos] = (bad[b, pos] + 1) % 8404
    assert not bench.golden_check("paraformer", bad)["ok"] and not bench.golden_check("paraformer", bad, g["token_num"])["ok"]
    # int8 goldens exist for the benchmark batch and carry what the check needs
    for tag in ("paraformer_int8", "paraformer_int8q", "seaco_int8"):
        gi = np.load(os.path.join(ROOT, "tests", "golden", "bench_%s.npz" % tag))
        assert gi["ids"].shape[0] == 32 and "alpha_sum" in gi.files and "margin" in gi.files
