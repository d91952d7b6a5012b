// recognizer.h — C++ mirror of the reference's public classes OfflineRecognizer
// (AliParaformerAsr/OfflineRecognizer.cs:13-477) and OfflineStream
// (AliParaformerAsr/OfflineStream.cs:7-121) above the device engine.
#pragma once
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"
#include "hostutil.h"

namespace pf {

// OfflineRecognizerResultEntity (Model/OfflineRecognizerResultEntity.cs:9-29)
struct ResultEntity {
  std::string Text;
  int TextLen = 0;                              // UTF-16 length, as C# string.Length
  std::vector<std::string> Tokens;
  std::vector<std::vector<int32_t>> Timestamps;
};

// DecodeMulti for one stream (OfflineRecognizer.cs:304-418)
ResultEntity decode_multi_one(const std::vector<std::string>& tokens, const std::vector<int64_t>& ids,
                              const std::vector<std::vector<int32_t>>& timestamps);
// time_stamp_lfr6_onnx (OfflineRecognizer.cs:200-302); throws PF_ERR_RECOGNITION where the C#
// would throw inside Forward's try block.
std::vector<std::vector<int32_t>> time_stamp_lfr6(const float* us_cif_peak, int n, std::vector<int64_t> tokens);
// GetHotwords (OfflineRecognizer.cs:72-90) over in-memory lines; appends [sos_eos_id]
std::vector<std::vector<int32_t>> hotword_ids(const std::vector<std::string>& tokens,
                                              const std::vector<std::string>& lines, int sos_eos_id);

class Recognizer;

// Ownership: a stream shares ownership of its recognizer OBJECT (so a stream handle that outlives
// pf_recognizer_free still reaches valid memory and answers PF_ERR_DISPOSED), the recognizer does not track
// its streams (the reference's CreateOfflineStream does not either, OfflineRecognizer.cs:92-100), and the
// device engine is released by Dispose() — never by the last stream going away.
class Stream {
 public:
  explicit Stream(std::shared_ptr<Recognizer> r);
  void AddSamples(const float* samples, int64_t n);          // OfflineStream.cs:36-57
  void Dispose();                                             // OfflineStream.cs:81-121: drops the buffers
  std::vector<float> Speech;                                  // OfflineInputEntity.Speech
  bool has_speech = false;                                    // Speech != null
  int SpeechLength = 0;                                       // float count
  bool hotwords_null = false;
  std::vector<std::vector<int32_t>> Hotwords;
  std::vector<int64_t> Tokens{0, 0};                          // OfflineStream.cs:26
  std::vector<std::vector<int32_t>> Timestamps;
  void RemoveChunk();                                         // OfflineStream.cs:69-79
  bool disposed = false;
  std::shared_ptr<Recognizer> owner;
};

class Recognizer : public std::enable_shared_from_this<Recognizer> {
 public:
  Recognizer(const std::string& model, const std::string& config, const std::string& mvn,
             const std::string& tokens, const std::string& modeleb, const std::string& hotword, int batch_size,
             int threads_num, int device);
  ~Recognizer();
  std::shared_ptr<Stream> CreateOfflineStream();              // OfflineRecognizer.cs:92-100
  // GetResults (:110-116).  The result list belongs to the CALLING THREAD until its next GetResults on this
  // recognizer (the reference returns a fresh List per call; concurrent callers must not share one).
  void GetResults(const std::vector<Stream*>& streams);
  const std::vector<ResultEntity>& results_of_this_thread();
  void Dispose();                                             // waits for calls in flight, then frees the engine
  bool disposed() const { return disposed_.load(); }
  // the engine for the duration of one call (nullptr once disposed); callers lock engine->mutex() themselves
  std::shared_ptr<Engine> engine() { std::lock_guard<std::mutex> lk(mu_); return engine_; }
  const std::vector<std::string>& tokens() const { return tokens_; }

 private:
  void Forward(const std::vector<Stream*>& streams);          // :118-198
  std::mutex mu_;                                             // guards engine_
  std::shared_ptr<Engine> engine_;
  std::vector<std::string> tokens_;
  ConfEntity conf_;
  std::vector<std::vector<int32_t>> hotwords_;
  uint64_t uid_ = 0;                                          // key of this recognizer in the per-thread result store
  std::atomic<bool> disposed_{false};
  friend class Stream;
};

}  // namespace pf
