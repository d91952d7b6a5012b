// HipWavFrontend.cs — WavFrontend (AliParaformerAsr/WavFrontend.cs:18-153) over libparaformer_hip.so: GetFbank
// (external kaldi fbank, :31-37) and LfrCmvn (:39-51) become ONE device call; am.mvn is parsed once per engine by the
// native side (the reference re-parses it per stream, WavFrontend.cs:28).  OfflineStream.AddSamples
// (OfflineStream.cs:36-57) keeps its shape:
//
//     float[] features = _wavFrontend.GetFeatures(samples);     // instead of GetFbank + LfrCmvn
//     int oLen = _offlineInputEntity.SpeechLength; ... (unchanged)
using System;
using AliParaformerAsr.Native;

namespace AliParaformerAsr
{
    internal sealed class HipWavFrontend
    {
        private readonly IntPtr _engine;      // owned by HipOfflineProj
        private readonly int _featDim;

        public HipWavFrontend(IntPtr engine, int featDim = 560) { _engine = engine; _featDim = featDim; }

        public float[] GetFeatures(float[] samples)
        {
            if (samples == null) throw new ArgumentNullException("source");       // LINQ Select on null, WavFrontend.cs:34
            ParaformerHip.Check(ParaformerHip.pf_frontend_num_frames(_engine, samples.LongLength, out int t));
            var feats = new float[Math.Max(t, 1) * _featDim];
            ParaformerHip.Check(ParaformerHip.pf_frontend(_engine, samples, samples.LongLength, feats, feats.LongLength, out t));
            Array.Resize(ref feats, t * _featDim);
            return feats;
        }

        /// <summary>OnlineFbank.GetFbank alone ([T80, 80]), for callers that keep the managed LFR / CMVN.</summary>
        public float[] GetFbank(float[] samples)
        {
            if (samples == null) throw new ArgumentNullException("source");
            long cap = (samples.LongLength / 160 + 2) * 80;
            var fb = new float[cap];
            ParaformerHip.Check(ParaformerHip.pf_fbank(_engine, samples, samples.LongLength, fb, cap, out int t80));
            Array.Resize(ref fb, t80 * 80);
            return fb;
        }
    }
}
