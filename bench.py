#!/usr/bin/env python
"""bench.py — headline benchmark of the offline Paraformer path on MI355X.

Metric (BASELINE.json): RTFx = audio-seconds / wall-seconds (the inverse of the reference's
`rtf = elapsed_ms / total_audio_ms`, AliParaformerAsr.Examples/OfflineAliParaformerAsrRecognizer.cs:244-249;
both are printed) + utterances/s, paraformer-large, batch 32 x 30 s synthetic 16 kHz per GPU.

One "step" = one pass of the whole hot path (fbank -> LFR/CMVN -> SAN-M encoder x50 -> CIF ->
SAN-M decoder x16 -> vocab GEMM -> last-index arg-max, + the gather of hypotheses when N > 1)
over one batch per GPU, audio already resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU; utterances are independent, so each rank recognises its own
shard (weak scaling, no data-path collective); RCCL is used for the one-off weight broadcast
(rank 0 -> all) and the per-step gather of hypotheses.

The JSON line also carries
  roofline     — the dominant kernel class (FFN up-projection GEMM) timed live with HIP events
                 on the engine stream during the timed steps: algorithmic FLOPs / avg duration
                 against the 2.5 PFLOP/s dense f16 MFMA peak;
  cpu_baseline — the CPU oracle (a port, NOT onnxruntime: neither ORT nor a model file exists
                 in the image) timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH_PER_GPU = 32
SECONDS = 30
SAMPLES = SECONDS * 16000
LCAP = 512
DOMINANT = "gemm_ffn1"
PEAK_F16_TFLOPS = 2500.0      # MI355X dense f16/bf16 MFMA (MI355X_MICROARCH.md)
PMC_FILE = os.path.join(ROOT, "profiles", "round2_pmc.json")


def ids_checksum(ids) -> str:
    """Order-sensitive checksum of the [B, L] arg-max ids of a step (tests/test_gpu_baseline_sizes.py pins it to
    the oracle for a depth-reduced 32 x 30 s run)."""
    return hashlib.sha1(np.ascontiguousarray(ids, dtype=np.int64).tobytes()).hexdigest()


def respawn_under_torchrun(n: int) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves, one process per GPU,
    exactly as the documented launch line does (RCCL rendezvous on 127.0.0.1), and relay rank 0's JSON line."""
    import torch
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible\n" % (n, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PF_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dominant_kernel_name(rows, cus):
    """Which kernel launch_gemm picks for the FFN up-projection [rows x 512] x [512 x 2048] (csrc/k_gemm.hip: the
    persistent 256 x 256-tile kernel when its schedule has fewer idle rounds, else the 256 x 128 persistent kernel)."""
    cd = lambda a, b: (a + b - 1) // b
    t_big, t_pp3 = cd(rows, 256) * 8, cd(rows, 256) * 16
    big = os.environ.get("PF_BIGP", "1") != "0" and t_big >= cus and 1.9 * cd(t_big, cus) <= cd(t_pp3, cus)
    return "gemm_bigp_kernel" if big else "gemm_f16_pp3<3, 2>"


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of the headline command
    (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes; tools/pmc_summary.py, notes in the file).  bench.py cannot
    run a profiler itself."""
    try:
        with open(PMC_FILE) as f:
            for k in json.load(f)["kernels"]:
                if kernel.replace(" ", "") in k["kernel"].replace(" ", "") and k["launches"] >= 50:
                    return float(k["traffic_bytes_per_launch"])
    except Exception:
        pass
    return None


def ort_reference_baseline(cmvn_unused=None):
    """The reference's own CPU path when it can be run on this box: `onnxruntime` importable AND a real model
    directory in $PF_MODEL_DIR (model.onnx or model.int8.onnx + am.mvn).  Session options mirror
    AliParaformerAsr/OfflineModel.cs:41-57: CPU EP, ORT_ENABLE_ALL, inter-op threads = threadsNum (CLI default 2,
    AliParaformerAsr.Examples/Program.cs:98-101), intra-op left at the ORT default (all cores), memory pattern on.
    Returns None when either piece is missing (the expected case: neither exists in the build image)."""
    probe = {"onnxruntime": False, "dotnet": False, "model_dir": os.environ.get("PF_MODEL_DIR") or None}
    try:
        probe["dotnet"] = subprocess.call(["dotnet", "--version"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0
    except OSError:
        pass
    try:
        import onnxruntime as ort
        probe["onnxruntime"] = True
    except Exception:
        return None, probe
    d = probe["model_dir"]
    if not d:
        return None, probe
    path = next((os.path.join(d, n) for n in ("model.int8.onnx", "model.onnx", "model_quant.onnx")
                 if os.path.exists(os.path.join(d, n))), None)
    mvn = os.path.join(d, "am.mvn")
    if not path or not os.path.exists(mvn):
        return None, probe
    from oracle import frontend as fe
    from aliparaformerasr_amd import weights as W
    so = ort.SessionOptions()
    so.graph_optimization_level = ort.GraphOptimizationLevel.ORT_ENABLE_ALL
    so.inter_op_num_threads = 2
    so.enable_mem_pattern = True
    sess = ort.InferenceSession(path, so, providers=["CPUExecutionProvider"])
    shift, scale = fe.parse_mvn_text(open(mvn, encoding="utf-8").read())
    conf = fe.FrontendConf(dither=0.0)
    n_utts = 8
    audio = [W.synth_audio(SAMPLES, u) for u in range(n_utts)]
    t0 = time.perf_counter()
    feats = [fe.wav_frontend(a, conf, shift, scale) for a in audio]
    speech = fe.pad_sequence(feats).reshape(n_utts, -1, 560)
    lens = np.full((n_utts,), speech.shape[1], np.int32)            # quirk Q2: speech_lengths = Tmax for every row
    out = sess.run(None, {sess.get_inputs()[0].name: speech, sess.get_inputs()[1].name: lens})
    np.argmax(out[0], -1)
    dt = time.perf_counter() - t0
    return {"value": n_utts * SECONDS / dt, "unit": "audio-sec/wall-sec", "cores": os.cpu_count(), "kind": "reference",
            "sample": "%d x %d s synthetic utterances through onnxruntime %s CPU EP on %s (OfflineModel.cs:41-57 options), %.1f s wall"
                      % (n_utts, SECONDS, ort.__version__, os.path.basename(path), dt),
            "rtf": dt / (n_utts * SECONDS)}, probe


def cpu_baseline(cfg, weights, cmvn):
    """CPU baseline beside the GPU number: the reference's onnxruntime path when it exists on this box (kind
    "reference"), else this repository's fp32 torch-CPU port of the same graph with fused kernels (kind "port":
    a stand-in, NOT onnxruntime), on a bounded sample of the same workload."""
    ref, probe = ort_reference_baseline()
    if ref is not None:
        ref["probe"] = probe
        return ref
    import torch
    from oracle import frontend as fe, model as om
    from aliparaformerasr_amd import weights as W
    threads = torch.get_num_threads()
    conf = fe.FrontendConf(dither=0.0)
    orc = om.Oracle(om.ModelConfig(**cfg), weights, quant="fp32", fast=True)
    warm = [W.synth_audio(16000, 999)]
    sp = fe.pad_sequence([fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in warm]).reshape(1, -1, 560)
    with torch.inference_mode():
        orc.paraformer(sp)
    def run(n):
        audio = [W.synth_audio(SAMPLES, u) for u in range(n)]
        t0 = time.perf_counter()
        feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
        speech = fe.pad_sequence(feats).reshape(n, -1, 560)
        with torch.inference_mode():
            out = orc.paraformer(speech)
        om.argmax_last(out["logits"])
        return time.perf_counter() - t0
    t1 = run(1)                                   # sizes the bounded sample (~15 s of CPU work)
    n_utts = int(min(16, max(1, round(15.0 / max(t1, 1e-3)))))
    dt = run(n_utts) if n_utts > 1 else t1
    return {"value": n_utts * SECONDS / dt, "unit": "audio-sec/wall-sec", "cores": threads, "kind": "port",
            "sample": "%d x %d s utterances of the same synthetic workload, fp32 torch-CPU port with fused LayerNorm / "
                      "attention kernels (stand-in, not onnxruntime), %.1f s wall" % (n_utts, SECONDS, dt),
            "rtf": dt / (n_utts * SECONDS), "probe": probe,
            "reference_published": "rtf 0.0371 (RTFx 27) on an i7-10750H, settings unstated (README.EN.md:134-136)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0,
                    help="utterances per GPU (default: 32 = BASELINE.json configs[1]; 64 for --model sensevoice = configs[2]; "
                         "128 at --gpus 8 = the per-GPU shard of configs[3], 1024 x 30 s over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="extra untimed step with per-class kernel times")
    ap.add_argument("--model", choices=("paraformer", "sensevoice", "seaco"), default="paraformer",
                    help="sensevoice = BASELINE.json configs[2] (sensevoice-small, 64 x 10 s, use_itn on); seaco = configs[4] "
                         "(SeACo bias decoder + 20 hotwords + BiCIF timestamps); neither is the headline config")
    ap.add_argument("--seconds", type=int, default=0, help="utterance length (default 30; 10 for --model sensevoice)")
    ap.add_argument("--timestamp-head", action="store_true",
                    help="BASELINE.json configs[4]-style variant: adds the BiCIF timestamp head (not the headline config)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args.gpus))

    import torch
    import torch.distributed as dist
    from aliparaformerasr_amd import weights as W
    from aliparaformerasr_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d "
                         "(or plain `python bench.py --gpus %d`, which spawns the ranks itself)" % (args.gpus, world, args.gpus, args.gpus))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)

    # ---- weights: rank 0 builds the synthetic paraformer-large image, RCCL-broadcasts it
    from aliparaformerasr_amd import shard as sh
    sv = args.model == "sensevoice"
    seconds = args.seconds or (10 if sv else SECONDS)
    samples = seconds * 16000
    if args.batch <= 0:
        args.batch = 64 if sv else (128 if (world == 8 and args.model == "paraformer") else BATCH_PER_GPU)
    if sv:
        cfg = W.sensevoice_small_config(use_itn=True)
    elif args.model == "seaco":
        cfg = W.seaco_paraformer_config()
        args.timestamp_head = True
    else:
        cfg = W.paraformer_large_config(timestamp_head=bool(args.timestamp_head))
    cmvn = W.synth_cmvn()
    weights = None
    blob = b""
    if rank == 0:
        weights = W.synth_weights(cfg, 42)
        blob = W.pack_pfw(cfg, weights)
    if world > 1:
        wdev = sh.broadcast_bytes(blob, dist, dev)            # RCCL broadcast rank 0 -> all
    else:
        wdev = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    del blob
    n = wdev.numel()
    torch.cuda.synchronize()
    eng = Engine(weights_device_ptr=wdev.data_ptr(), weights_bytes=n, cmvn=cmvn, device=local)

    # ---- workload: this rank's shard of the utterance list, staged to HBM before timing
    B = args.batch
    audio = [W.synth_audio(samples, rank * B + u) for u in range(B)]
    eng.stage_audio(audio)
    if args.model == "seaco":                                 # N = 20 hotwords of 2-4 ids + the [1] terminator
        hrng = np.random.default_rng(99)
        hws = [list(hrng.integers(3, 8000, size=int(hrng.integers(2, 5)))) for _ in range(20)] + [[1]]
        eng.set_hotwords(np.asarray([h[:10] + [0] * (10 - len(h)) for h in hws], np.int32))
    gathered = {}

    def step():
        eng.run_staged()
        if world > 1:                                         # gather of hypotheses over RCCL
            r = eng.fetch()
            gathered["ids"] = sh.gather_hypotheses(r.token_ids, world * B, LCAP, dist, dev)

    for _ in range(args.warmup):
        step()
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    eng.profile_reset()
    eng.profile_select(DOMINANT)
    eng.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        if i == 0:
            eng.profile(False)        # the dominant class is event-timed on the FIRST timed step only (50 launches):
                                      # an event pair costs ~4 us of device time, 50 pairs per step would tax `value` by ~1.4 %
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    eng.profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    res = eng.fetch()
    ms_dom, n_dom, fpl_dom = eng.profile_get(DOMINANT)
    # the timed steps must have produced a real transcript-shaped result: every utterance decoded, ids in range
    assert res.L > 0 and res.token_ids.shape == (B, res.L), (res.L, res.token_ids.shape)
    assert (res.token_ids >= 0).all() and (res.token_ids < eng.vocab).all()
    assert (res.token_num > 0).all()
    if world > 1:
        g = gathered["ids"]
        assert g.shape == (world * B, LCAP) and (g[rank * B:(rank + 1) * B, :res.L] == res.token_ids).all()

    # PCIe-inclusive rate (never `value`): host float32 audio in, ids back on the host, per batch
    host_ms = None
    if rank == 0:
        eng.recognize(audio)
        t1 = time.perf_counter()
        for _ in range(2):
            eng.recognize(audio)
        host_ms = (time.perf_counter() - t1) / 2 * 1e3

    breakdown = None
    if args.breakdown and rank == 0:
        eng.profile_reset()
        eng.profile_select("")
        eng.profile(True)
        eng.run_staged()
        eng.sync()
        eng.profile(False)
        breakdown = {}
        for cls in ("fbank", "lfr_cmvn_pad", "layernorm", "gemm_qkv", "fsmn", "attn_self", "gemm_out", "gemm_ffn1",
                    "gemm_ffn2", "gemm_cif", "cif_misc", "gemm_dec_kv", "gemm_dec_ffn1", "gemm_dec_ffn2",
                    "gemm_dec_q", "attn_cross", "gemm_dec_out", "gemm_vocab", "argmax", "gemm_ts", "lstm", "ts_misc",
                    "seaco_embed", "gemm_seaco", "attn_seaco", "seaco_merge"):
            ms, cnt, fpl = eng.profile_get(cls)
            breakdown[cls] = {"ms": round(ms, 4), "launches": cnt,
                              "tflops": round(fpl * cnt / (ms * 1e-3) / 1e12, 1) if ms > 0 and fpl > 0 else None}

    if rank == 0:
        audio_s = world * B * seconds * args.steps
        value = audio_s / dt
        flops_step = eng.last_flops()
        ach = fpl_dom / ((ms_dom / max(n_dom, 1)) * 1e-3) / 1e12 if n_dom else 0.0
        dom_rows = B * int(res.L if sv else eng.num_frames(samples))
        dom_kernel = dominant_kernel_name(dom_rows, torch.cuda.get_device_properties(0).multi_processor_count)
        out = {
            "metric": "RTFx (audio-sec/wall-sec), %s offline, batch %dx%ds per GPU"
                      % ("sensevoice-small" if sv else "paraformer-large", B, seconds),
            "value": value, "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "%s offline%s, batch %dx%d s synthetic 16 kHz per GPU "
                                   "(BASELINE.json configs[%d]), seeded synthetic weights"
                                   % ("sensevoice-small (use_itn on)" if sv else ("SeACo-paraformer, 21 hotwords" if args.model == "seaco" else "paraformer-large-zh"),
                                      " + BiCIF timestamp head" if args.timestamp_head else "", B, seconds,
                                      2 if sv else (4 if args.timestamp_head else (3 if world * B == 1024 else 1))),
                       "global_batch": world * B, "samples_per_utt": samples, "T_lfr": eng.num_frames(samples), "L": int(res.L),
                       "parallelism": "dp%d (utterance shards, no data-path collective)" % world},
            "rtf": dt / audio_s, "utt_per_s": world * B * args.steps / dt,
            "rccl_ranks": world if world > 1 else 0,
            "ids_sha1": ids_checksum(res.token_ids),   # rank 0's [B, L] ids of the last timed step
            "token_num_sum": int(res.token_num.sum()),
            "host_audio_ms_per_batch": host_ms,     # one GPU's batch incl. H2D of the audio and D2H of the ids
            "algorithmic_tflop_per_step_per_gpu": flops_step / 1e12,
            "whole_path_tflops_per_gpu": flops_step * args.steps / dt / 1e12,
            "roofline": {"bound": "mfma", "kernel": "%s (class %s: [%d x 512] x [512 x 2048] + bias + ReLU)"
                         % (dom_kernel, DOMINANT, dom_rows), "achieved": ach, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_F16_TFLOPS,
                         "traffic": pmc_traffic(dom_kernel) if (not sv and args.model == "paraformer" and not args.timestamp_head and B == BATCH_PER_GPU and seconds == SECONDS) else None,
                         "traffic_unit": "bytes/launch (PMC, profiles/round2_pmc.json)",
                         "algorithmic_bytes_per_launch": int(fpl_dom / (2 * 512 * 2048)) * (512 + 2048) * 2 + 2048 * 512 * 2,
                         "launches_timed": int(n_dom), "avg_us": ms_dom / max(n_dom, 1) * 1e3,
                         "flops_per_launch": fpl_dom},
        }
        if breakdown is not None:
            out["kernel_breakdown_ms_per_step"] = breakdown
        if world == 1 and not args.no_cpu_baseline and not sv and seconds == SECONDS:
            out["cpu_baseline"] = cpu_baseline(cfg, weights, cmvn)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
