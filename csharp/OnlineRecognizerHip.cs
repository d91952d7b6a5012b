// OnlineRecognizerHip.cs — drop-in classes with the public signatures of the streaming recogniser
// (AliParaformerAsr/OnlineRecognizer.cs:22,27,32,41,533) and OnlineStream (OnlineStream.cs), backed by the native
// mirror pf_online_* (include/paraformer_hip.h section 7): chunking, feature caches, DynamicMask, the carried-integrator
// CIF and DecodeMulti run behind the C ABI exactly where the reference runs them; only samples go in and text comes out.
// Not compiled in this repository's build image (no .NET toolchain); the same entry points are exercised through the
// Python binding aliparaformerasr_amd/online_recognizer.py by tests/test_gpu_online.py.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using AliParaformerAsr.Model;
using AliParaformerAsr.Native;

namespace AliParaformerAsr.Hip
{
    internal static class OnlineNative
    {
        const string Lib = "paraformer_hip";
        [DllImport(Lib)] internal static extern int pf_online_recognizer_create(string encoderPath, string decoderPath, string configPath,
            string mvnPath, string tokensPath, int threadsNum, int device, out IntPtr recognizer);
        [DllImport(Lib)] internal static extern void pf_online_recognizer_dispose(IntPtr r);
        [DllImport(Lib)] internal static extern void pf_online_recognizer_free(IntPtr r);
        [DllImport(Lib)] internal static extern int pf_online_create_stream(IntPtr r, out IntPtr stream);
        [DllImport(Lib)] internal static extern int pf_online_stream_add_samples(IntPtr s, float[] samples, long n);
        [DllImport(Lib)] internal static extern int pf_online_get_results(IntPtr r, IntPtr[] streams, int nStreams);
        [DllImport(Lib)] internal static extern int pf_online_result_text(IntPtr r, int i, out IntPtr utf8);
        [DllImport(Lib)] internal static extern void pf_online_stream_dispose(IntPtr s);
        [DllImport(Lib)] internal static extern void pf_online_stream_free(IntPtr s);
    }

    public sealed class OnlineStream : IDisposable
    {
        internal IntPtr Handle;
        internal OnlineStream(IntPtr h) { Handle = h; }

        // OnlineStream.AddSamples: appends audio; the library cuts it into 60-frame chunks and keeps the caches
        public void AddSamples(float[] samples)
            => ParaformerHip.Check(OnlineNative.pf_online_stream_add_samples(Handle, samples, samples == null ? 0 : samples.LongLength));

        public void Dispose()
        {
            if (Handle == IntPtr.Zero) return;
            OnlineNative.pf_online_stream_dispose(Handle);      // idempotent; later calls answer PF_ERR_DISPOSED
            OnlineNative.pf_online_stream_free(Handle);
            Handle = IntPtr.Zero;
            GC.SuppressFinalize(this);
        }
        ~OnlineStream() { Dispose(); }
    }

    public sealed class OnlineRecognizer : IDisposable
    {
        IntPtr _h;

        // OnlineRecognizer.cs:22 — encoderFilePath names the .pfw container that holds both graphs' tensors
        // (python -m aliparaformerasr_amd.convert); decoderFilePath is accepted and unused; `device` is the extra argument
        public OnlineRecognizer(string encoderFilePath, string decoderFilePath, string configFilePath, string mvnFilePath,
                                string tokensFilePath, int threadsNum = 1, int device = 0)
            => ParaformerHip.Check(OnlineNative.pf_online_recognizer_create(encoderFilePath, decoderFilePath, configFilePath,
                                                                          mvnFilePath, tokensFilePath, threadsNum, device, out _h));

        public OnlineStream CreateOnlineStream()                                                   // :27
        {
            ParaformerHip.Check(OnlineNative.pf_online_create_stream(_h, out IntPtr s));
            return new OnlineStream(s);
        }

        public OnlineRecognizerResultEntity GetResult(OnlineStream stream)                          // :32
            => GetResults(new List<OnlineStream> { stream })[0];

        public List<OnlineRecognizerResultEntity> GetResults(List<OnlineStream> streams)           // :41
        {
            var hs = new IntPtr[streams.Count];
            for (int i = 0; i < hs.Length; i++) hs[i] = streams[i].Handle;
            ParaformerHip.Check(OnlineNative.pf_online_get_results(_h, hs, hs.Length));
            var res = new List<OnlineRecognizerResultEntity>(hs.Length);
            for (int i = 0; i < hs.Length; i++)
            {
                ParaformerHip.Check(OnlineNative.pf_online_result_text(_h, i, out IntPtr p));
                string text = Marshal.PtrToStringUTF8(p) ?? string.Empty;
                res.Add(new OnlineRecognizerResultEntity { Text = text, TextLen = text.Length });
            }
            return res;
        }

        public void Dispose()                                                                      // :533
        {
            if (_h == IntPtr.Zero) return;
            OnlineNative.pf_online_recognizer_dispose(_h);
            OnlineNative.pf_online_recognizer_free(_h);
            _h = IntPtr.Zero;
            GC.SuppressFinalize(this);
        }
        ~OnlineRecognizer() { Dispose(); }
    }
}
