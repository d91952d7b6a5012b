// ParaformerHip.cs — P/Invoke declarations of libparaformer_hip.so (include/paraformer_hip.h, PF_ABI_VERSION 6).
// Drop into the reference project (AliParaformerAsr/Native/) — see csharp/README.md.  Not compiled in the build
// image of this repository (no .NET toolchain); the same entry points are exercised through the Python ctypes
// binding aliparaformerasr_amd/_native.py by tests/.
using System;
using System.Runtime.InteropServices;

namespace AliParaformerAsr.Native
{
    /// <summary>pf_engine_config (paraformer_hip.h): replaces OfflineModel.initModel's SessionOptions
    /// (OfflineModel.cs:35-70) and FrontendConfEntity (Model/FrontendConfEntity.cs:7-15).</summary>
    [StructLayout(LayoutKind.Sequential)]
    internal struct PfEngineConfig
    {
        public int struct_size, device;
        public IntPtr weights_path, weights_host, weights_device;
        public long weights_bytes;
        public IntPtr mvn_path, cmvn_shift, cmvn_scale;
        public int cmvn_dim;
        public int fs, n_mels, lfr_m, lfr_n, snip_edges;
        public float dither;
        public IntPtr window;
        public int use_itn;
        public int frame_length_ms, frame_shift_ms, dither_seed;
        /// <summary>0 = f16 MFMA (model.onnx semantics, default), 1 = fp32 MFMA parity mode, 2 = dynamic int8 as the
        /// reference's default model.int8.onnx computes (Examples/Program.cs:98-101): Linear layers on the int8 MFMA,
        /// 3 = "exact" at matrix-core speed (ABI 5): the fp32 graph with every large Linear as three f16 MFMA products of
        /// (hi, lo) operand pairs — token-identical to fp32 on the benchmark batches.</summary>
        public int math_mode;
        public int reserved0, reserved1, reserved2;
    }

    /// <summary>pf_batch_out: capacities in, L / V / cif_peak_len and the filled buffers out.</summary>
    [StructLayout(LayoutKind.Sequential)]
    internal struct PfBatchOut
    {
        public int struct_size, l_cap;
        public long logits_cap, cif_peak_cap;
        public IntPtr token_ids, token_num, logits, cif_peak;
        public int L, V, cif_peak_len, reserved;
    }

    internal static class ParaformerHip
    {
        private const string Lib = "paraformer_hip";   // libparaformer_hip.so next to the assembly / on LD_LIBRARY_PATH

        internal const int PF_OK = 0, PF_ERR_INVALID_ARG = -1, PF_ERR_DEVICE = -2, PF_ERR_IO = -3, PF_ERR_FORMAT = -4,
                           PF_ERR_CAPACITY = -5, PF_ERR_UNSUPPORTED = -6, PF_ERR_DISPOSED = -7, PF_ERR_TOKENS = -8,
                           PF_ERR_NULL_SAMPLES = -9, PF_ERR_RECOGNITION = -10;

        [DllImport(Lib)] internal static extern int pf_version();
        [DllImport(Lib)] internal static extern IntPtr pf_last_error();

        // ---- engine (replaces InferenceSession + OnlineFbank) -------------------------------------------------
        [DllImport(Lib)] internal static extern int pf_engine_create(ref PfEngineConfig cfg, out IntPtr engine);
        [DllImport(Lib)] internal static extern void pf_engine_destroy(IntPtr engine);
        [DllImport(Lib)] internal static extern int pf_engine_info(IntPtr e, out int kind, out int vocab, out int featDim, out int hasTs);
        [DllImport(Lib)] internal static extern int pf_frontend_num_frames(IntPtr e, long nSamples, out int tLfr);
        [DllImport(Lib)] internal static extern int pf_frontend(IntPtr e, float[] samples, long n, [Out] float[] feats, long featsCap, out int tLfr);
        [DllImport(Lib)] internal static extern int pf_fbank(IntPtr e, float[] samples, long n, [Out] float[] fbank, long cap, out int t80);
        [DllImport(Lib)] internal static extern int pf_model_proj(IntPtr e, IntPtr[] speech, int[] speechLenFloats, int B,
                                                                 int[]? hotwords, int nHotwords, ref PfBatchOut o);
        [DllImport(Lib)] internal static extern int pf_forward_feats(IntPtr e, float[] speech, int B, int tMax,
                                                                    int[]? hotwords, int nHotwords, ref PfBatchOut o);
        [DllImport(Lib)] internal static extern int pf_recognize(IntPtr e, IntPtr[] samples, long[] nSamples, int B,
                                                                int[]? hotwords, int nHotwords, ref PfBatchOut o);
        [DllImport(Lib)] internal static extern int pf_fetch(IntPtr e, ref PfBatchOut o);
        [DllImport(Lib)] internal static extern int pf_fetch_ids_device(IntPtr e, IntPtr idsDev, int lCap, out int L);

        // ---- several GPUs in one process (paraformer_hip.h section 4b) ------------------------------------------
        [DllImport(Lib)] internal static extern int pf_group_create(ref PfEngineConfig cfg, int[] devices, int nDevices, out IntPtr group);
        [DllImport(Lib)] internal static extern void pf_group_destroy(IntPtr group);
        [DllImport(Lib)] internal static extern int pf_group_info(IntPtr g, out int nEngines, out int usesRccl);
        [DllImport(Lib)] internal static extern int pf_group_recognize(IntPtr g, IntPtr[] samples, long[] nSamples, int B,
                                                                      int[]? hotwords, int nHotwords, ref PfBatchOut o);
        [DllImport(Lib)] internal static extern int pf_group_fetch(IntPtr g, ref PfBatchOut o);
        /// <summary>Host-only rehearsal of pf_group_recognize's shard plan / rendez-vous / merge (no GPU): see paraformer_hip.h.</summary>
        [DllImport(Lib)] internal static extern int pf_host_group_sim(int G, int B, int[] fireCount, int hasCif, int fixedL, int collective,
                                                                     int failShard, int failStage, [Out] long[] idsOut, int lCap,
                                                                     [Out] int[] tokenNumOut, out int L);

        // ---- profiling (bench harness) ---------------------------------------------------------------------------
        [DllImport(Lib)] internal static extern int pf_profile_enable(IntPtr e, int on);
        [DllImport(Lib)] internal static extern int pf_profile_reset(IntPtr e);
        [DllImport(Lib, CharSet = CharSet.Ansi)] internal static extern int pf_profile_select(IntPtr e, string? className);
        [DllImport(Lib, CharSet = CharSet.Ansi)] internal static extern int pf_profile_get(IntPtr e, string className, out double totalMs, out long launches, out double flopsPerLaunch);
        [DllImport(Lib, CharSet = CharSet.Ansi)] internal static extern int pf_profile_kernel(IntPtr e, string className, [Out] byte[] nameOut, int cap);

        // ---- whole-class mirror (OfflineRecognizer / OfflineStream) ---------------------------------------------
        [DllImport(Lib, CharSet = CharSet.Ansi)]
        internal static extern int pf_recognizer_create([MarshalAs(UnmanagedType.LPUTF8Str)] string model,
            [MarshalAs(UnmanagedType.LPUTF8Str)] string config, [MarshalAs(UnmanagedType.LPUTF8Str)] string mvn,
            [MarshalAs(UnmanagedType.LPUTF8Str)] string tokens, [MarshalAs(UnmanagedType.LPUTF8Str)] string modeleb,
            [MarshalAs(UnmanagedType.LPUTF8Str)] string hotword, int batchSize, int threadsNum, int device, out IntPtr recognizer);
        [DllImport(Lib)] internal static extern void pf_recognizer_dispose(IntPtr r);
        [DllImport(Lib)] internal static extern void pf_recognizer_free(IntPtr r);
        [DllImport(Lib)] internal static extern int pf_recognizer_num_engines(IntPtr r);   // engines of the pool ($PF_RECOGNIZER_ENGINES)
        [DllImport(Lib)] internal static extern int pf_recognizer_create_stream(IntPtr r, out IntPtr stream);
        [DllImport(Lib)] internal static extern int pf_stream_add_samples(IntPtr s, float[]? samples, long n);
        [DllImport(Lib)] internal static extern int pf_stream_set_hotwords(IntPtr s, int[]? ids, int[]? lens, int nHotwords);
        [DllImport(Lib)] internal static extern int pf_stream_get_hotwords(IntPtr s, [Out] int[] ids, int idsCap, [Out] int[] lens, int lensCap, out int nHotwords);
        [DllImport(Lib)] internal static extern int pf_stream_num_feature_floats(IntPtr s, out int n);
        [DllImport(Lib)] internal static extern int pf_stream_tokens(IntPtr s, out IntPtr ids, out int n);
        // ABI 6: the rest of OfflineStream's public surface (OfflineStream.cs:20-34)
        [DllImport(Lib)] internal static extern int pf_stream_create([MarshalAs(UnmanagedType.LPUTF8Str)] string mvnPath, int fs, int nMels, int lfrM, int lfrN,
                                                                    int snipEdges, float dither, [MarshalAs(UnmanagedType.LPUTF8Str)] string window, out IntPtr stream);
        [DllImport(Lib)] internal static extern int pf_stream_set_tokens(IntPtr s, long[]? ids, int n);
        [DllImport(Lib)] internal static extern int pf_stream_num_timestamps(IntPtr s, out int n);
        [DllImport(Lib)] internal static extern int pf_stream_timestamp(IntPtr s, int j, out IntPtr ints, out int nInts);
        [DllImport(Lib)] internal static extern int pf_stream_set_timestamps(IntPtr s, int[]? ints, int[]? lens, int n);
        [DllImport(Lib)] internal static extern int pf_stream_get_speech(IntPtr s, [Out] float[]? speech, long cap, out int nFloats);
        [DllImport(Lib)] internal static extern int pf_stream_set_speech(IntPtr s, float[]? speech, int nFloats, int speechLength);
        [DllImport(Lib)] internal static extern void pf_stream_dispose(IntPtr s);
        [DllImport(Lib)] internal static extern void pf_stream_free(IntPtr s);
        [DllImport(Lib)] internal static extern int pf_recognizer_get_results(IntPtr r, IntPtr[] streams, int nStreams);
        [DllImport(Lib)] internal static extern int pf_result_text(IntPtr r, int i, out IntPtr utf8, out int textLenUtf16);
        [DllImport(Lib)] internal static extern int pf_result_num_tokens(IntPtr r, int i, out int n);
        [DllImport(Lib)] internal static extern int pf_result_token(IntPtr r, int i, int j, out IntPtr utf8);
        [DllImport(Lib)] internal static extern int pf_result_num_timestamps(IntPtr r, int i, out int n);
        [DllImport(Lib)] internal static extern int pf_result_timestamp(IntPtr r, int i, int j, out IntPtr ints, out int nInts);

        /// <summary>pf_status -> the exception the reference throws at the same place (INTEGRATION.md section 1).</summary>
        internal static void Check(int rc)
        {
            if (rc >= 0) return;
            string msg = Marshal.PtrToStringUTF8(pf_last_error()) ?? "";
            switch (rc)
            {
                case PF_ERR_TOKENS: throw new Exception("tokens invalid");                        // OfflineRecognizer.cs:32
                case PF_ERR_DISPOSED: throw new ObjectDisposedException(msg.Length > 0 ? msg : "OfflineRecognizer");   // :96
                case PF_ERR_NULL_SAMPLES: throw new ArgumentNullException("source");              // WavFrontend.cs:34
                case PF_ERR_RECOGNITION: throw new Exception("Offline recognition failed", new Exception(msg));   // :194-197
                default: throw new Exception(msg);
            }
        }
    }
}
