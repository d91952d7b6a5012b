// online.h — C++ mirror of the reference's STREAMING classes above the device engine:
//   OnlineRecognizer  AliParaformerAsr/OnlineRecognizer.cs:14-542   (Forward :336-403, PredictorProj :126-231 =
//                     the C# CIF with the carried integrator, DecodeMulti :405-437)
//   OnlineStream      AliParaformerAsr/OnlineStream.cs:7-358        (AddSamples :79-105, InputSpeech :106-160,
//                     GetDecodeChunk :162-208: splice cache, LFR, CMVN, x sqrt(512), position encoding, 10-frame cache)
//   OnlineWavFrontend AliParaformerAsr/OnlineWavFrontend.cs:63-80 (LFR without left context), :152-188 (PE)
//   OnlineModel       AliParaformerAsr/OnlineModel.cs:141-165 (DynamicMask), :199-247 (stack / unstack of the FSMN caches)
// The two ONNX sessions (encoder, decoder) are replaced by Engine::online_encoder / online_decoder.
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "engine.h"
#include "hostutil.h"

namespace pf {

// ---- pure host pieces, exposed for parity tests ---------------------------------------------------------------
// OnlineWavFrontend.ApplyLfr (:63-80): windows of lfr_m frames every lfr_n, NO left context;
// t_lfr = t/lfr_n - 1 when t % lfr_n < lfr_m - lfr_n, else t/lfr_n
std::vector<float> online_apply_lfr(const std::vector<float>& fbank, int n_mels, int lfr_m, int lfr_n);
// OnlineWavFrontend.SinusoidalPositionEncoder (:152-188): adds rows start_idx .. of the table whose row p-1 is
// [sin(inv_i * p) | cos(inv_i * p)], inv_i = exp(-(i + 1) * ln(1e4) / (dim/2 - 1))   (note i + 1: differs from offline)
void online_position_encode(std::vector<float>& x, int timesteps, int dim, int start_idx);
// OnlineModel.DynamicMask (:141-165): alphas[0:5] = 0 and alphas[15:] = 0 (chunk_size 5, lfr 10)
void online_dynamic_mask(std::vector<float>& alphas, int chunk_size = 5, int lfr = 10);
// one stream's share of PredictorProj (:146-197): integrate-and-fire over [carried pair ; new frames].
// hiddens [n][D], alphas [n]; returns fired frames; carry_alpha / carry_hidden = the new cache
void online_cif(const std::vector<std::vector<float>>& hiddens, const std::vector<float>& alphas, float threshold,
                std::vector<std::vector<float>>& fired, float& carry_alpha, std::vector<float>& carry_hidden);
// OnlineRecognizer.DecodeMulti (:405-437) for one stream
std::string online_decode_text(const std::vector<std::string>& tokens, const std::vector<int64_t>& ids);

class OnlineRecognizerM;

class OnlineStreamM {
 public:
  explicit OnlineStreamM(std::shared_ptr<OnlineRecognizerM> r);
  void AddSamples(const float* samples, int64_t n);           // OnlineStream.cs:79-105
  bool GetDecodeChunk(std::vector<float>& chunk);             // :162-208 (chunk = [10 cached ; 10 new] x 560), false = not enough frames
  std::vector<int64_t> Tokens{0, 0};                          // :52
  std::vector<std::vector<float>> States;                     // 16 x [512*10]
  std::vector<std::vector<float>> CifHidden;                  // carried hidden(s)
  std::vector<float> CifAlpha;
  std::vector<float> Speech;                                  // OnlineInputEntity.Speech (fbank frames, 80-dim)
  bool disposed = false;
  std::shared_ptr<OnlineRecognizerM> owner;
  // The reference serialises AddSamples / InputSpeech / GetDecodeChunk with lock(obj) (OnlineStream.cs:30,86,116,174):
  // a capture thread may add samples while a decode thread runs GetResults on the same stream.  Here the lock is per
  // stream; it is never held while the engine mutex is taken (AddSamples releases it around the fbank call), so
  // Forward (engine mutex, then stream state) cannot dead-lock against it.
  std::mutex mu;

 private:
  void InputSpeech(const std::vector<float>& samples);        // :106-160
  std::vector<float> cache_samples_, cache_feats_, cache_lfr_splice_;
  bool first_input_ = true;                                   // _cacheInput.Length == 0
  int start_idx_ = 0;
};

class OnlineRecognizerM : public std::enable_shared_from_this<OnlineRecognizerM> {
 public:
  // (encoderFilePath, decoderFilePath, configFilePath, mvnFilePath, tokensFilePath, threadsNum) — OnlineRecognizer.cs:22;
  // here ONE .pfw container holds both graphs' tensors (model_path); decoder_path is accepted and unused
  OnlineRecognizerM(const std::string& model, const std::string& decoder_unused, const std::string& config,
                    const std::string& mvn, const std::string& tokens, int threads_num, int device);
  std::shared_ptr<OnlineStreamM> CreateOnlineStream();
  std::vector<std::string> GetResults(const std::vector<OnlineStreamM*>& streams);   // Forward + DecodeMulti
  void Dispose();
  bool disposed() const { return disposed_.load(); }
  std::shared_ptr<Engine> engine() { std::lock_guard<std::mutex> lk(mu_); return engine_; }
  const ConfEntity& conf() const { return conf_; }
  int chunk_length() const { return 10 * 5 + 10; }            // OnlineModel.cs:29: lfr * chunkSize + 10 = 60 fbank frames
  const std::vector<float>& shift() const { return shift_; }
  const std::vector<float>& scale() const { return scale_; }
  const std::vector<std::string>& tokens() const { return tokens_; }

 private:
  void Forward(const std::vector<OnlineStreamM*>& streams);   // :336-403
  std::mutex mu_;
  std::shared_ptr<Engine> engine_;
  ConfEntity conf_;
  std::vector<float> shift_, scale_;
  std::vector<std::string> tokens_;
  std::atomic<bool> disposed_{false};
};

}  // namespace pf
