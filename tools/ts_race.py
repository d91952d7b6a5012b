import os, sys, subprocess, numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    from aliparaformerasr_amd import weights as W
    from aliparaformerasr_amd.engine import Engine
    cfg = W.seaco_paraformer_config() if sys.argv[1] == "seaco" else W.paraformer_large_config(timestamp_head=True)
    eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 42)), cmvn=W.synth_cmvn(), device=0)
    audio = [W.synth_audio(30 * 16000, u) for u in range(32)]
    hw = None
    if sys.argv[1] == "seaco":
        hrng = np.random.default_rng(99)
        hws = [list(hrng.integers(3, 8000, size=int(hrng.integers(2, 5)))) for _ in range(20)] + [[1]]
        hw = np.asarray([h[:10] + [0] * (10 - len(h)) for h in hws], np.int32)
    outs = []
    for i in range(3):
        r = eng.recognize(audio, hotwords=hw) if hw is not None else eng.recognize(audio)
        outs.append((r.token_ids.copy(), r.cif_peak.copy() if r.cif_peak is not None else None))
    np.savez(sys.argv[2], **{"ids%d" % i: o[0] for i, o in enumerate(outs)}, **{"pk%d" % i: o[1] for i, o in enumerate(outs) if o[1] is not None})
    sys.exit(0)
VARIANTS = [("1", {}), ("1", {"PF_LSTM_VAR": "0"})]
for model in ("ts", "seaco"):
  base = "/tmp/tsr_base.npz"
  subprocess.check_call([sys.executable, __file__, model, base], env=dict(os.environ, PF_TS_STREAM="0"))
  a = np.load(base)
  for v, extra in VARIANTS:
    f = "/tmp/tsr_%s_%s.npz" % (model, v)
    subprocess.check_call([sys.executable, __file__, model, f], env=dict(os.environ, PF_TS_STREAM=v, **extra))
    b = np.load(f)
    print("== PF_TS_STREAM=%s %r" % (v, extra))
    for i in range(3):
        d = a["ids0"] != b["ids%d" % i]
        print("  ", model, "run", i, "ids differ at", int(d.sum()), "positions; rows", np.unique(np.nonzero(d)[0])[:12], "cols", np.unique(np.nonzero(d)[1])[:12])
        if "pk0" in a.files:
            print("      peaks max diff", float(np.abs(a["pk0"] - b["pk%d" % i]).max()))
