// OfflineRecognizerHip.cs — drop-in classes with the public signatures of OfflineRecognizer
// (AliParaformerAsr/OfflineRecognizer.cs:23,92,102,110,441,468) and OfflineStream (OfflineStream.cs:20,30-34,36,58,69,115), backed
// by the native mirror pf_recognizer_* / pf_stream_*: front-end, model, arg-max, time_stamp_lfr6_onnx and
// DecodeMulti all run behind the C ABI; only ids, timestamps and text cross it.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using AliParaformerAsr.Model;
using AliParaformerAsr.Native;

namespace AliParaformerAsr.Hip
{
    public class OfflineStream : IDisposable
    {
        internal IntPtr Handle;
        internal OfflineStream(IntPtr h) { Handle = h; }

        /// <summary>OfflineStream.cs:20-28.  A stream that belongs to no recognizer yet: its AddSamples calls are kept and
        /// replayed by the first GetResults that receives it (the front-end runs on that recognizer's GPU); the am.mvn values
        /// and frontend_conf must be that recognizer's.</summary>
        public OfflineStream(string mvnFilePath, ConfEntity confEntity)
        {
            FrontendConfEntity f = confEntity.frontend_conf;
            ParaformerHip.Check(ParaformerHip.pf_stream_create(mvnFilePath ?? "", f.fs, f.n_mels, f.lfr_m, f.lfr_n, f.snip_edges ? 1 : 0,
                                                               f.dither, f.window ?? "", out Handle));
        }

        public void AddSamples(float[] samples)
            => ParaformerHip.Check(ParaformerHip.pf_stream_add_samples(Handle, samples, samples == null ? 0 : samples.LongLength));

        /// <summary>OfflineStream.cs:58-68: the entity Forward reads.  A snapshot: features that live on the device are
        /// computed and read back for it.</summary>
        public OfflineInputEntity GetDecodeChunk() => OfflineInputEntity;

        /// <summary>OfflineStream.cs:69-79.</summary>
        public void RemoveChunk()
        {
            if (Tokens.Count > 2) ParaformerHip.Check(ParaformerHip.pf_stream_set_speech(Handle, null, -1, 0));
        }

        /// <summary>OfflineStream.cs:30.  get: a snapshot {Speech, SpeechLength, Hotwords}; set: written through.</summary>
        public OfflineInputEntity OfflineInputEntity
        {
            get
            {
                var e = new OfflineInputEntity();
                ParaformerHip.Check(ParaformerHip.pf_stream_num_feature_floats(Handle, out int len));
                int rc = ParaformerHip.pf_stream_get_speech(Handle, null, 0, out int n);
                if (rc == ParaformerHip.PF_ERR_CAPACITY)
                {
                    var a = new float[n];
                    ParaformerHip.Check(ParaformerHip.pf_stream_get_speech(Handle, a, a.LongLength, out n));
                    e.Speech = a;
                }
                else
                {
                    ParaformerHip.Check(rc);
                    e.Speech = n < 0 ? null : new float[0];
                }
                e.SpeechLength = len;
                e.Hotwords = Hotwords;
                return e;
            }
            set
            {
                ParaformerHip.Check(ParaformerHip.pf_stream_set_speech(Handle, value?.Speech, value?.Speech == null ? -1 : value.Speech.Length,
                                                                       value?.SpeechLength ?? 0));
                Hotwords = value?.Hotwords;
            }
        }

        /// <summary>OfflineStream.cs:31.  {blank, blank} from the constructor on; nothing in the reference reads it afterwards
        /// (Forward works on Tokens), so it stays a managed field.</summary>
        public Int64[] Hyp { get; set; } = new Int64[] { 0, 0 };

        public List<int[]>? Hotwords
        {
            get
            {
                var ids = new int[4096]; var lens = new int[1024];
                ParaformerHip.Check(ParaformerHip.pf_stream_get_hotwords(Handle, ids, ids.Length, lens, lens.Length, out int n));
                if (n < 0) return null;
                var r = new List<int[]>(); int off = 0;
                for (int i = 0; i < n; i++) { r.Add(ids[off..(off + lens[i])]); off += lens[i]; }
                return r;
            }
            set
            {
                if (value == null) { ParaformerHip.Check(ParaformerHip.pf_stream_set_hotwords(Handle, null, null, -1)); return; }
                var flat = new List<int>(); var lens = new int[Math.Max(value.Count, 1)];
                for (int i = 0; i < value.Count; i++) { flat.AddRange(value[i]); lens[i] = value[i].Length; }
                ParaformerHip.Check(ParaformerHip.pf_stream_set_hotwords(Handle, flat.ToArray(), lens, value.Count));
            }
        }

        public List<Int64> Tokens
        {
            get
            {
                ParaformerHip.Check(ParaformerHip.pf_stream_tokens(Handle, out IntPtr p, out int n));
                var a = new long[n];
                if (n > 0) Marshal.Copy(p, a, 0, n);
                return new List<Int64>(a);
            }
            set
            {
                long[] a = value == null ? new long[0] : value.ToArray();
                ParaformerHip.Check(ParaformerHip.pf_stream_set_tokens(Handle, a, a.Length));
            }
        }

        public List<int[]> Timestamps                                           // OfflineStream.cs:33
        {
            get
            {
                ParaformerHip.Check(ParaformerHip.pf_stream_num_timestamps(Handle, out int n));
                var r = new List<int[]>(n);
                for (int j = 0; j < n; j++)
                {
                    ParaformerHip.Check(ParaformerHip.pf_stream_timestamp(Handle, j, out IntPtr p, out int k));
                    var a = new int[k];
                    if (k > 0) Marshal.Copy(p, a, 0, k);
                    r.Add(a);
                }
                return r;
            }
            set
            {
                var flat = new List<int>(); var lens = new int[Math.Max(value?.Count ?? 0, 1)];
                for (int i = 0; i < (value?.Count ?? 0); i++) { flat.AddRange(value![i]); lens[i] = value[i].Length; }
                ParaformerHip.Check(ParaformerHip.pf_stream_set_timestamps(Handle, flat.ToArray(), lens, value?.Count ?? 0));
            }
        }

        protected virtual void Dispose(bool disposing)                          // OfflineStream.cs:81-113
        {   // later calls answer ObjectDisposedException("OfflineStream"); the finaliser releases the handle
            if (Handle != IntPtr.Zero) ParaformerHip.pf_stream_dispose(Handle);
        }
        public void Dispose() => Dispose(disposing: true);                      // :115 (the finaliser is NOT suppressed: it frees the handle)
        ~OfflineStream() { if (Handle != IntPtr.Zero) { ParaformerHip.pf_stream_free(Handle); Handle = IntPtr.Zero; } }
    }

    public sealed class OfflineRecognizer : IDisposable
    {
        private IntPtr _r;

        public OfflineRecognizer(string modelFilePath, string configFilePath, string mvnFilePath, string tokensFilePath,
                                 string modelebFilePath = "", string hotwordFilePath = "", int batchSize = 1, int threadsNum = 1,
                                 int device = 0)
            => ParaformerHip.Check(ParaformerHip.pf_recognizer_create(modelFilePath, configFilePath, mvnFilePath, tokensFilePath,
                                                                      modelebFilePath ?? "", hotwordFilePath ?? "", batchSize,
                                                                      threadsNum, device, out _r));

        /// <summary>Engines of this recognizer's pool (round 5).  GetResults holds no lock in the reference
        /// (OfflineRecognizer.cs:110-198), so a server calls it from several threads; here every call takes a free engine of the
        /// pool (same device, one copy of the weights) — created on demand up to $PF_RECOGNIZER_ENGINES (default 2).</summary>
        public int NumEngines { get { int n = ParaformerHip.pf_recognizer_num_engines(_r); ParaformerHip.Check(n < 0 ? n : 0); return n; } }

        public OfflineStream CreateOfflineStream()
        {
            ParaformerHip.Check(ParaformerHip.pf_recognizer_create_stream(_r, out IntPtr s));
            return new OfflineStream(s);
        }

        public OfflineRecognizerResultEntity GetResult(OfflineStream stream) => GetResults(new List<OfflineStream> { stream })[0];

        public List<OfflineRecognizerResultEntity> GetResults(List<OfflineStream> streams)
        {
            var hs = new IntPtr[Math.Max(streams.Count, 1)];
            for (int i = 0; i < streams.Count; i++) hs[i] = streams[i].Handle;
            ParaformerHip.Check(ParaformerHip.pf_recognizer_get_results(_r, hs, streams.Count));
            var res = new List<OfflineRecognizerResultEntity>();
            for (int i = 0; i < streams.Count; i++)
            {
                var e = new OfflineRecognizerResultEntity();
                ParaformerHip.Check(ParaformerHip.pf_result_text(_r, i, out IntPtr txt, out int len16));
                e.Text = Marshal.PtrToStringUTF8(txt);
                e.TextLen = len16;
                ParaformerHip.Check(ParaformerHip.pf_result_num_tokens(_r, i, out int nt));
                for (int j = 0; j < nt; j++)
                {
                    ParaformerHip.Check(ParaformerHip.pf_result_token(_r, i, j, out IntPtr t));
                    e.Tokens.Add(Marshal.PtrToStringUTF8(t) ?? "");
                }
                ParaformerHip.Check(ParaformerHip.pf_result_num_timestamps(_r, i, out int nts));
                for (int j = 0; j < nts; j++)
                {
                    ParaformerHip.Check(ParaformerHip.pf_result_timestamp(_r, i, j, out IntPtr p, out int k));
                    var a = new int[k];
                    if (k > 0) Marshal.Copy(p, a, 0, k);
                    e.Timestamps.Add(a);
                }
                res.Add(e);
            }
            return res;
        }

        public void DisposeOfflineStream(OfflineStream offlineStream) => offlineStream?.Dispose();

        public void Dispose()
        {
            // frees the engine(s) now; later calls answer ObjectDisposedException.  The finaliser is NOT suppressed: it
            // releases the handle shell (pf_recognizer_free), exactly as OfflineStream does above.
            if (_r != IntPtr.Zero) ParaformerHip.pf_recognizer_dispose(_r);
        }
        ~OfflineRecognizer() { if (_r != IntPtr.Zero) { ParaformerHip.pf_recognizer_free(_r); _r = IntPtr.Zero; } }
    }
}
