"""Where does the fused encoder launch (k_ffn.hip, 64-row tiles) start to pay?  Whole-path time for batches whose encoder
row count M = B x T lies between the short-input path (M <= 512) and the benchmark (M = 16 000), with the fused forms forced
on from 513 rows (PF_FFN_MIN=513) and off (PF_FFN_FUSED=0).  Run once per setting:
    PF_FFN_MIN=513 python tools/mid_rows.py ; PF_FFN_FUSED=0 PF_ATTN_FFN=0 python tools/mid_rows.py ; python tools/mid_rows.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aliparaformerasr_amd import weights as W
from aliparaformerasr_amd.engine import Engine
cfg = W.paraformer_large_config()
eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 42)), cmvn=W.synth_cmvn(), device=0)
for _ in range(10):
    eng.recognize([W.synth_audio(5 * 16000, 0)])
for (B, secs) in ((2, 30), (4, 30), (8, 30), (16, 30), (8, 10), (16, 10)):
    audio = [W.synth_audio(secs * 16000, u) for u in range(B)]
    eng.stage_audio(audio)
    for _ in range(3):
        eng.run_staged(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.run_staged(); eng.sync()
    dt = (time.perf_counter() - t0) / 10
    print("B=%2d x %2ds (M = %5d rows): %.2f ms" % (B, secs, B * eng.num_frames(secs * 16000), dt * 1e3), flush=True)
eng.close()
