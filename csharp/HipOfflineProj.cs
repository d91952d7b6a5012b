// HipOfflineProj.cs — IOfflineProj (AliParaformerAsr/IOfflineProj.cs:6-40) over libparaformer_hip.so.
//
// The seam is UNCHANGED: ModelProj returns the same ModelOutputEntity the ONNX-backed projections return
// (model_out = log-probs [B, L, V], model_out_lens = token_num, cif_peak_tensor = us_cif_peak [B, 3T]), so
// OfflineRecognizer.Forward (OfflineRecognizer.cs:118-198) — its arg-max loop, time_stamp_lfr6_onnx and DecodeMulti —
// runs untouched on top of it.  Select it in the constructor switch (OfflineRecognizer.cs:39-53):
//
//     case "paraformer": _offlineProj = new HipOfflineProj(_offlineModel, modelFilePath /* .pfw */, mvnFilePath, _confEntity); break;
//
// Moving the [B, L, V] log-probs to the host costs PCIe time (161 MB for 32 x 150 x 8404); callers that only need
// the transcript should use OfflineRecognizerHip (OfflineRecognizerHip.cs), which keeps the arg-max on the device and
// returns ids — the ids are identical (the device scans the same log-probs, ties to the larger index).
using System;
using System.Collections.Generic;
using System.Linq;
using System.Runtime.InteropServices;
using AliParaformerAsr.Model;
using AliParaformerAsr.Native;
using Microsoft.ML.OnnxRuntime;
using Microsoft.ML.OnnxRuntime.Tensors;

namespace AliParaformerAsr
{
    internal sealed class HipOfflineProj : IOfflineProj, IDisposable
    {
        private IntPtr _engine;
        private readonly List<int[]>? _defaultHotwords;
        private readonly bool _seaco;

        // IOfflineProj members that only make sense for an ORT session: kept for the interface, unused by Forward
        public InferenceSession ModelSession { get => null!; set { } }
        public int Blank_id { get; set; } = 0;
        public int Sos_eos_id { get; set; } = 1;
        public int Unk_id { get; set; } = 2;
        public int SampleRate { get; set; } = 16000;
        public int FeatureDim { get; set; } = 80;

        public HipOfflineProj(OfflineModel offlineModel, string pfwPath, string mvnPath, ConfEntity conf, int device = 0)
        {
            _defaultHotwords = offlineModel.Hotwords;                       // OfflineProjOfSeacoParaformer.cs:57-60
            var cfg = new PfEngineConfig { struct_size = Marshal.SizeOf<PfEngineConfig>(), device = device };
            IntPtr w = Marshal.StringToCoTaskMemUTF8(pfwPath), m = Marshal.StringToCoTaskMemUTF8(mvnPath),
                   win = Marshal.StringToCoTaskMemUTF8(conf.frontend_conf.window);
            try
            {
                cfg.weights_path = w; cfg.mvn_path = m; cfg.window = win;
                cfg.fs = conf.frontend_conf.fs; cfg.n_mels = conf.frontend_conf.n_mels;
                cfg.lfr_m = conf.frontend_conf.lfr_m; cfg.lfr_n = conf.frontend_conf.lfr_n;
                cfg.snip_edges = conf.frontend_conf.snip_edges ? 1 : 0;
                cfg.dither = conf.frontend_conf.dither;
                cfg.frame_length_ms = conf.frontend_conf.frame_length; cfg.frame_shift_ms = conf.frontend_conf.frame_shift;
                cfg.use_itn = conf.use_itn ? 1 : 0;
                ParaformerHip.Check(ParaformerHip.pf_engine_create(ref cfg, out _engine));
            }
            finally { Marshal.FreeCoTaskMem(w); Marshal.FreeCoTaskMem(m); Marshal.FreeCoTaskMem(win); }
            ParaformerHip.Check(ParaformerHip.pf_engine_info(_engine, out int kind, out _, out int feat, out _));
            _seaco = kind == 2;
            FeatureDim = feat;
        }

        internal IntPtr Engine => _engine;

        public ModelOutputEntity ModelProj(List<OfflineInputEntity> modelInputs)
        {
            int B = modelInputs.Count;
            var pins = new GCHandle[B];
            var ptrs = new IntPtr[B];
            var lens = new int[B];
            var result = new ModelOutputEntity();
            try
            {
                for (int i = 0; i < B; i++)
                {   // PadHelper.PadSequence (+ the -754511.06 sentinel) happens on the device
                    pins[i] = GCHandle.Alloc(modelInputs[i].Speech, GCHandleType.Pinned);
                    ptrs[i] = pins[i].AddrOfPinnedObject();
                    lens[i] = modelInputs[i].SpeechLength;
                }
                int[]? hot = null; int nHot = 0;
                if (_seaco)
                {   // OfflineProjOfSeacoParaformer.cs:52-60 + EmbedSeacoModel.PadList(…, 0, 10): ids [N, 10]
                    List<int[]> hw = modelInputs.Where(x => x.Hotwords != null).SelectMany(x => x.Hotwords!).ToList();
                    if (hw.Count == 0 && _defaultHotwords != null) hw = _defaultHotwords;
                    nHot = hw.Count;
                    hot = new int[nHot * 10];
                    for (int n = 0; n < nHot; n++)
                        for (int j = 0; j < 10 && j < hw[n].Length; j++) hot[n * 10 + j] = hw[n][j];
                }
                // call 1 learns L, V and the peak length; the log-probs must be requested here (1-float dummy)
                var probe = new float[1];
                var hProbe = GCHandle.Alloc(probe, GCHandleType.Pinned);
                var o = new PfBatchOut { struct_size = Marshal.SizeOf<PfBatchOut>(), logits = hProbe.AddrOfPinnedObject(), logits_cap = 1 };
                int rc;
                try { rc = ParaformerHip.pf_model_proj(_engine, ptrs, lens, B, hot, nHot, ref o); }
                finally { hProbe.Free(); }
                if (rc != ParaformerHip.PF_ERR_CAPACITY) ParaformerHip.Check(rc);     // "capacity" on the dummy is expected
                int L = o.L, V = o.V, P = o.cif_peak_len;
                var logits = new float[(long)B * L * V];
                var ids = new long[B * Math.Max(L, 1)];
                var tokenNum = new int[B];
                var peak = new float[B * Math.Max(P, 1)];
                var hL = GCHandle.Alloc(logits, GCHandleType.Pinned); var hI = GCHandle.Alloc(ids, GCHandleType.Pinned);
                var hT = GCHandle.Alloc(tokenNum, GCHandleType.Pinned); var hP = GCHandle.Alloc(peak, GCHandleType.Pinned);
                try
                {
                    o.logits = hL.AddrOfPinnedObject(); o.logits_cap = logits.LongLength;
                    o.token_ids = hI.AddrOfPinnedObject(); o.l_cap = Math.Max(L, 1);
                    o.token_num = hT.AddrOfPinnedObject();
                    if (P > 0) { o.cif_peak = hP.AddrOfPinnedObject(); o.cif_peak_cap = peak.LongLength; }
                    ParaformerHip.Check(ParaformerHip.pf_fetch(_engine, ref o));
                }
                finally { hL.Free(); hI.Free(); hT.Free(); hP.Free(); }
                result.model_out = new DenseTensor<float>(logits, new[] { B, L, V });          // = out[0] (OfflineProjOfParaformer.cs:72)
                result.model_out_lens = tokenNum;                                              // = out[1] (:73)
                if (P > 0) result.cif_peak_tensor = new DenseTensor<float>(peak, new[] { B, P });   // = out[3] (:76-79)
            }
            catch (Exception ex)
            {
                throw new Exception("ModelProj failed", ex);                                  // OfflineProjOfParaformer.cs:82-85
            }
            finally
            {
                foreach (var p in pins) if (p.IsAllocated) p.Free();
            }
            return result;
        }

        public void Dispose()
        {
            IntPtr e = _engine;
            _engine = IntPtr.Zero;
            if (e != IntPtr.Zero) ParaformerHip.pf_engine_destroy(e);       // idempotent on the native side as well
        }
        void IOfflineProj.Dispose() => Dispose();
        ~HipOfflineProj() { Dispose(); }
    }
}
