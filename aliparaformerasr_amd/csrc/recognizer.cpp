// recognizer.cpp — see recognizer.h.  Host-side logic of the reference's public API, restated
// in C++ (the reference is compiled C#; no .NET toolchain exists in the build image).
#include "recognizer.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstring>
#include <set>

namespace pf {

static const std::string kBar = "\xE2\x96\x81";   // U+2581 '▁'

static void replace_all(std::string& s, const std::string& from, const std::string& to) {
  if (from.empty()) return;
  size_t pos = 0;
  while ((pos = s.find(from, pos)) != std::string::npos) {
    s.replace(pos, from.size(), to);
    pos += to.size();
  }
}
static std::string replaced(std::string s, const std::string& from, const std::string& to) {
  replace_all(s, from, to);
  return s;
}
static int count_sub(const std::string& s, const std::string& sub) {
  int n = 0;
  size_t pos = 0;
  while ((pos = s.find(sub, pos)) != std::string::npos) { ++n; pos += sub.size(); }
  return n;
}
// IsChinese(str, allMatch: true): ^[一-龥]+$  (OfflineRecognizer.cs:428-439)
static bool is_chinese_all(const std::string& s) {
  // (the decoding of utf8_decode, hostutil.cpp, without the vector: this runs once per token of every result)
  const size_t n = s.size();
  if (n == 0) return false;
  for (size_t i = 0; i < n;) {
    const unsigned char c = (unsigned char)s[i];
    uint32_t cp;
    int extra;
    if (c < 0x80) { cp = c; extra = 0; }
    else if ((c >> 5) == 0x6) { cp = c & 0x1F; extra = 1; }
    else if ((c >> 4) == 0xE) { cp = c & 0x0F; extra = 2; }
    else if ((c >> 3) == 0x1E) { cp = c & 0x07; extra = 3; }
    else { cp = 0xFFFD; extra = 0; }
    ++i;
    for (int k = 0; k < extra && i < n; ++k, ++i) cp = (cp << 6) | ((unsigned char)s[i] & 0x3F);
    if (cp < 0x4e00 || cp > 0x9fa5) return false;
  }
  return true;
}
// C# IndexOf(x) > 0 : found and not at position 0
static bool index_gt0(const std::string& s, const std::string& sub) {
  const size_t p = s.find(sub);
  return p != std::string::npos && p > 0;
}
static bool index_lt0(const std::string& s, const std::string& sub) { return s.find(sub) == std::string::npos; }

template <class T>
static void remove_first_equal_to_last(std::vector<T>& v) {
  // List<T>.Remove(list.Last()): removes the FIRST element equal to the last one (quirk Q10)
  if (v.empty()) throw Error(PF_ERR_RECOGNITION, "Sequence contains no elements");
  const T last = v.back();
  auto it = std::find(v.begin(), v.end(), last);
  v.erase(it);
}

// kind of a vocabulary entry for DecodeMulti: 2 = one of the four markers it skips, 1 = all-Chinese, 0 = anything else
static int token_kind(const std::string& cur) {
  if (cur == "</s>" || cur == "<s>" || cur == "<blank>" || cur == "<unk>") return 2;
  return is_chinese_all(cur) ? 1 : 0;
}
static std::string token_text(const std::string& line) {       // tokens.txt lines may carry "\t<id>": the text is what precedes the tab
  const size_t tab = line.find('\t');
  return tab == std::string::npos ? line : line.substr(0, tab);
}

TokenTable::TokenTable(const std::vector<std::string>& tokens) {
  cur.reserve(tokens.size());
  kind.reserve(tokens.size());
  for (const std::string& t : tokens) {
    cur.push_back(token_text(t));
    kind.push_back((uint8_t)token_kind(cur.back()));
  }
}

// DecodeMulti for one stream; `lookup(token, kind)` returns the entry's text and its kind (computed per token by the plain form,
// read from the recognizer's table by the other: 32 x 167 tokens per GetResults)
template <class Lookup>
static ResultEntity decode_multi_core(size_t vocab, const std::vector<int64_t>& ids, const TsList& timestamps, Lookup&& lookup) {
  ResultEntity r;
  r.Tokens.reserve(ids.size());
  r.Timestamps.reserve(ids.size());
  std::string text, lastToken;
  text.reserve(ids.size() * 4);
  bool haveLastTs = false;
  TsVec lastTs;
  const size_t n = std::min(ids.size(), timestamps.size());    // Zip
  for (size_t i = 0; i < n; ++i) {
    const int64_t token = ids[i];
    const TsVec& ts = timestamps[i];
    if (token == 2) break;
    if (token < 0 || token >= (int64_t)vocab) throw Error(PF_ERR_RECOGNITION, "token id out of range");
    int kind = 0;
    const std::string& cur = lookup((size_t)token, kind);
    if (kind == 2) continue;
    if (kind == 1) {
      text += cur;
      r.Tokens.push_back(cur);
      r.Timestamps.push_back(ts);
      continue;
    }
    text += kBar + cur + kBar;
    const std::string comb = lastToken + kBar + cur + kBar;
    auto merged_ts = [&]() {
      if (!haveLastTs) return ts;
      TsVec t = lastTs;
      t.append(ts.begin(), ts.end());
      return t;
    };
    if (index_gt0(comb, "@@" + kBar + kBar)) {
      const std::string curToken = replaced(comb, "@@" + kBar + kBar, "");
      const TsVec curTs = merged_ts();
      remove_first_equal_to_last(r.Tokens);
      r.Tokens.push_back(replaced(curToken, kBar, ""));
      if (r.Timestamps.empty()) throw Error(PF_ERR_RECOGNITION, "Sequence contains no elements");
      r.Timestamps.pop_back();                 // arrays compare by reference in C#: the last one
      r.Timestamps.push_back(curTs);
      lastToken = curToken; lastTs = curTs; haveLastTs = true;
    } else if ((count_sub(comb, kBar) == 3 || count_sub(comb, kBar) == 5) && index_lt0(comb, kBar + kBar + kBar)) {
      const std::string curToken = replaced(comb, kBar + kBar, "");
      const TsVec curTs = merged_ts();
      if (!r.Tokens.empty()) remove_first_equal_to_last(r.Tokens);
      r.Tokens.push_back(replaced(curToken, kBar, ""));
      if (!r.Timestamps.empty()) r.Timestamps.pop_back();
      r.Timestamps.push_back(curTs);
      lastToken = curToken; lastTs = curTs; haveLastTs = true;
    } else {
      r.Tokens.push_back(replaced(cur, kBar, ""));
      r.Timestamps.push_back(ts);
      lastToken = kBar + cur + kBar; lastTs = ts; haveLastTs = true;
    }
  }
  if (index_gt0(text, "@@" + kBar + kBar) || index_lt0(text, kBar + kBar + kBar)) {
    replace_all(text, "@@" + kBar + kBar, "");
    replace_all(text, kBar + kBar, " ");
    replace_all(text, "@@", " ");
    replace_all(text, kBar, " ");
  } else {
    replace_all(text, kBar + kBar + kBar, " ");
    replace_all(text, kBar + kBar, "");
    replace_all(text, kBar, "");
  }
  r.Text = text;
  r.TextLen = utf16_length(text);
  return r;
}

ResultEntity decode_multi_one(const std::vector<std::string>& tokens, const std::vector<int64_t>& ids,
                              const TsList& timestamps) {
  std::string scratch;
  return decode_multi_core(tokens.size(), ids, timestamps, [&](size_t t, int& kind) -> const std::string& {
    scratch = token_text(tokens[t]);
    kind = token_kind(scratch);
    return scratch;
  });
}

ResultEntity decode_multi_one(const TokenTable& table, const std::vector<int64_t>& ids, const TsList& timestamps) {
  return decode_multi_core(table.cur.size(), ids, timestamps, [&](size_t t, int& kind) -> const std::string& {
    kind = table.kind[t];
    return table.cur[t];
  });
}

TsList time_stamp_lfr6(const float* us_cif_peak, int n, const std::vector<int64_t>& tokens_in) {
  // float32 arithmetic throughout, (int)(t*1000) truncation (quirk Q10)
  const int START_END_THRESHOLD = 5, MAX_TOKEN_DURATION = 30;
  volatile float tr0 = 10.0f * 6;
  volatile float tr1 = tr0 / 1000;
  const float TIME_RATE = tr1 / 3;
  const float total_offset = -1.5f;
  const int num_frames = n;
  if (tokens_in.empty()) throw Error(PF_ERR_RECOGNITION, "Sequence contains no elements");   // tokens.Last()
  std::vector<int64_t> tokens(tokens_in);
  if (tokens.back() == 2) tokens.pop_back();
  std::vector<float> fire;
  for (int i = 0; i < n; ++i)
    if ((double)us_cif_peak[i] > (double)1.0f - 1e-4) fire.push_back((float)i + total_offset);
  if (fire.empty()) throw Error(PF_ERR_RECOGNITION, "Index was out of range (no CIF fire)");   // fire_place[0]
  std::vector<std::array<float, 2>> tl;
  std::vector<bool> nc;
  if (fire[0] > (float)START_END_THRESHOLD) {
    tl.push_back({0.0f, fire[0] * TIME_RATE});
    nc.push_back(false);
  }
  const int nf = (int)fire.size();
  for (int i = 0; i < nf - 1; ++i) {
    if (i >= (int)tokens.size()) throw Error(PF_ERR_RECOGNITION, "Index was outside the bounds of the array");
    nc.push_back(tokens[i] != 1);
    if (i == nf - 2 || MAX_TOKEN_DURATION < 0 || fire[i + 1] - fire[i] < (float)MAX_TOKEN_DURATION) {
      tl.push_back({fire[i] * TIME_RATE, fire[i + 1] * TIME_RATE});
    } else {
      const float split = fire[i] + (float)MAX_TOKEN_DURATION;
      tl.push_back({fire[i] * TIME_RATE, split * TIME_RATE});
      tl.push_back({split * TIME_RATE, fire[i + 1] * TIME_RATE});
      nc.push_back(false);
    }
  }
  if ((float)num_frames - fire.back() > (float)START_END_THRESHOLD) {
    const float end = ((float)num_frames + fire.back()) / 2;
    if (tl.empty()) throw Error(PF_ERR_RECOGNITION, "Sequence contains no elements");
    tl.back()[1] = end * TIME_RATE;
    tl.push_back({end * TIME_RATE, (float)num_frames * TIME_RATE});
    nc.push_back(false);
  } else {
    if (tl.empty()) throw Error(PF_ERR_RECOGNITION, "Sequence contains no elements");
    tl.back()[1] = (float)num_frames * TIME_RATE;
  }
  nc.push_back(true);
  TsList out;
  const size_t m = std::min(nc.size(), tl.size());
  for (size_t i = 0; i < m; ++i) {
    if (!nc[i]) continue;
    volatile float a = tl[i][0] * 1000, b = tl[i][1] * 1000;
    out.push_back({(int32_t)a, (int32_t)b});
  }
  return out;
}

std::vector<std::vector<int32_t>> hotword_ids(const std::vector<std::string>& tokens,
                                              const std::vector<std::string>& lines, int sos_eos_id) {
  std::vector<std::vector<int32_t>> hw;
  for (const std::string& sentence : lines) {
    std::vector<int32_t> ids;
    for (uint32_t cp : utf8_decode(sentence)) {
      if (cp >= 0x10000) continue;               // ToCharArray yields surrogate halves: never a token
      const std::string ch = utf8_encode(cp);
      int idx = -1;
      for (size_t t = 0; t < tokens.size(); ++t)
        if (tokens[t] == ch) { idx = (int)t; break; }    // Array.IndexOf: whole-line equality
      if (idx != -1) ids.push_back(idx);
    }
    hw.push_back(ids);
  }
  hw.push_back({sos_eos_id});
  return hw;
}

// Results live in thread-local storage keyed by the recognizer's uid: they die with the calling thread (no
// growth under thread-pool churn, no stale hit when the OS re-uses a thread id) and entries of recognizers that
// no longer exist are purged on the next call.
static std::mutex g_live_mu;
static std::set<uint64_t> g_live;
static uint64_t g_next_uid = 1;
static thread_local std::map<uint64_t, std::vector<ResultEntity>> t_results;

static uint64_t register_recognizer() {
  std::lock_guard<std::mutex> lk(g_live_mu);
  const uint64_t id = g_next_uid++;
  g_live.insert(id);
  return id;
}

// ------------------------------------------------------------------ Stream ----------------
Stream::Stream(std::shared_ptr<Recognizer> r) : owner(std::move(r)) {}
Stream::Stream(const std::string& mvn_path, const ConfEntity& conf) : uconf(conf) {
  if (!mvn_path.empty()) parse_mvn_text(read_text_file(mvn_path.c_str()), ushift, uscale);   // WavFrontend.cs:19 LoadCmvn
}
Stream::~Stream() {
  drop_device_audio();
  if (dev_ev) hipEventDestroy(dev_ev);
}

// frames GetFbank + LfrCmvn return for one AddSamples call of n samples (WavFrontend.cs:31-111; kaldi frame count)
static int conf_lfr_frames(const ConfEntity& c, int64_t n) {
  const int t80 = c.snip_edges ? (n < 400 ? 0 : (int)(1 + (n - 400) / 160)) : (int)((n + 80) / 160);
  return (c.lfr_m == 1 && c.lfr_n == 1) ? t80 : t80 / c.lfr_n;
}

void Stream::wait_device_audio() {
  if (dev_ev_pending) { hipEventSynchronize(dev_ev); dev_ev_pending = false; }
}

void Stream::drop_device_audio() {
  wait_device_audio();                                  // the buffer goes back to the cache: no DMA may still be writing it
  if (dev_audio && owner) owner->audio_free(dev_audio, dev_bytes);
  dev_audio = nullptr; dev_bytes = 0; dev_n = 0;
  device_form = false;
}

// device form -> host form: the features of the stored samples, computed by an engine of the pool and read back
void Stream::materialize() {
  if (!device_form) return;
  Recognizer::Lease e = owner->acquire();
  std::vector<float> feats;
  int t = 0;
  wait_device_audio();
  if (dev_n > 0) e->frontend_from_device(dev_audio, dev_n, feats, t);
  e.release();
  drop_device_audio();
  Speech.swap(feats);
  has_speech = true;
  SpeechLength = (int)Speech.size();
}

void Stream::AddSamples(const float* samples, int64_t n) {
  if (disposed) throw Error(PF_ERR_DISPOSED, "OfflineStream");
  if (!samples) throw Error(PF_ERR_NULL_SAMPLES, "source");       // ArgumentNullException("source")
  PF_CHECK(n >= 0, PF_ERR_INVALID_ARG, "negative sample count");
  if (!owner) {                                                    // public-constructor stream: replayed at adoption
    pending.emplace_back(samples, samples + n);
    const int m = (uconf.lfr_m != 1 || uconf.lfr_n != 1) ? uconf.lfr_m : 1;
    has_speech = true;
    SpeechLength += conf_lfr_frames(uconf, n) * m * uconf.n_mels;
    return;
  }
  if (owner->disposed()) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");
  if (!has_speech && !device_form && owner->device_streams()) {
    // first call: the samples go to the device and stay there; SpeechLength is what GetFbank + LfrCmvn would return
    // (OfflineStream.cs:40-41) — a function of the sample count
    // Device memory for a backlog of streams can run out where the reference never fails: then this stream simply takes the
    // host form below (features computed now and kept on the host, as in rounds 1-4) — ADVICE r5
    size_t got = 0;
    float* d = nullptr;
    bool on_device = true;
    try {
      if (n > 0) { d = owner->audio_alloc((size_t)n * 4, &got); owner->upload(d, samples, (size_t)n * 4, &dev_ev, &dev_ev_pending); }
    } catch (const Error& ex) {
      (void)hipGetLastError();                                      // clear the sticky out-of-memory status
      if (d) { (void)hipDeviceSynchronize(); owner->audio_free(d, got); }   // (pieces of a staged upload may be in flight)
      if (ex.code == PF_ERR_DISPOSED) throw;
      on_device = false;
    }
    if (on_device) {
      dev_audio = d; dev_bytes = got; dev_n = n;
      device_form = true;
      has_speech = true;
      SpeechLength = owner->feature_floats(n);
      return;
    }
  }
  materialize();                                                   // a second call appends to the FEATURES (:43-54)
  Recognizer::Lease e = owner->acquire();
  std::vector<float> feats;
  int t = 0;
  e->frontend_host(samples, n, feats, t);
  e.release();
  Speech.insert(Speech.end(), feats.begin(), feats.end());
  has_speech = true;
  SpeechLength = (int)Speech.size();
}

void Stream::Dispose() {
  disposed = true;
  drop_device_audio();
  std::vector<float>().swap(Speech);
  std::vector<int64_t>().swap(Tokens);
  TsList().swap(Timestamps);
  std::vector<std::vector<int32_t>>().swap(Hotwords);
  has_speech = false;
  SpeechLength = 0;
}

void Stream::RemoveChunk() {
  if (Tokens.size() > 2) {
    drop_device_audio();
    Speech.clear();
    has_speech = false;
    SpeechLength = 0;
  }
}

// ------------------------------------------------------------------ Recognizer ------------
Recognizer::Recognizer(const std::string& model, const std::string& config, const std::string& mvn,
                       const std::string& tokens, const std::string& modeleb, const std::string& hotword,
                       int /*batch_size (unused, quirk Q14)*/, int /*threads_num*/, int device) {
  (void)modeleb;
  conf_ = load_conf(config);                                      // OfflineRecognizer.cs:26
  // :29-33  ReadTokens; null/empty -> "tokens invalid" (checked before any device work so that
  // the contract test does not need a GPU)
  if (!tokens.empty()) {
    if (!file_exists(tokens)) throw Error(PF_ERR_IO, "tokens file not found: " + tokens);
    tokens_ = read_lines(tokens);
    token_table_ = TokenTable(tokens_);
  }
  if (tokens_.empty()) throw Error(PF_ERR_TOKENS, "tokens invalid");
  if (!hotword.empty() && file_exists(hotword))                    // :34-38, :75
    hotwords_ = hotword_ids(tokens_, read_lines(hotword), 1);
  device_ = device;
  model_path_ = model; mvn_path_ = mvn; window_ = conf_.window;
  pf_engine_config& ec = ec_;
  std::memset(&ec, 0, sizeof(ec));
  ec.struct_size = sizeof(ec);
  ec.device = device;
  // am.mvn is read HERE, once (the reference reads its files in the constructor only): engines created later for the pool
  // get the parsed vectors, not the path
  if (!mvn_path_.empty()) {
    parse_mvn_text(read_text_file(mvn_path_.c_str()), cmvn_shift_, cmvn_scale_);
    ec.cmvn_shift = cmvn_shift_.data(); ec.cmvn_scale = cmvn_scale_.data(); ec.cmvn_dim = (int32_t)cmvn_shift_.size();
  }
  ec.fs = conf_.fs; ec.n_mels = conf_.n_mels; ec.lfr_m = conf_.lfr_m; ec.lfr_n = conf_.lfr_n;
  ec.snip_edges = conf_.snip_edges ? 1 : 0;
  ec.dither = conf_.dither;
  // frame_length / frame_shift of the yaml are NOT forwarded: WavFrontend.cs:21-27 builds OnlineFbank from dither,
  // snip_edges, window, fs and n_mels only, so whatever the file says the reference frames at the kaldi defaults
  // (25 ms / 10 ms) — a conf with other values must run here exactly as it does there
  ec.frame_length_ms = 0;
  ec.frame_shift_ms = 0;
  ec.window = window_.c_str();
  ec.use_itn = conf_.use_itn ? 1 : 0;
  // `-accuracy int8` (the reference CLI's default, Examples/Program.cs:98-101) selects the arithmetic through the FILE
  // NAME there (model.int8.onnx vs model.onnx, Examples/OfflineAliParaformerAsrRecognizer.cs:17-22); the same
  // convention here: a container named *.int8.* runs with dynamically quantised Linear layers (math_mode 2)
  {
    const size_t sl = model.find_last_of("/\\");
    const std::string fname = sl == std::string::npos ? model : model.substr(sl + 1);
    if (fname.find(".int8.") != std::string::npos) ec.math_mode = 2;
  }
  { const char* e = getenv("PF_RECOGNIZER_ENGINES"); if (e && e[0]) max_engines_ = std::max(1, std::min(8, atoi(e))); }
  { const char* e = getenv("PF_RECOGNIZER_DEVICE_STREAMS"); if (e && e[0]) device_streams_ = e[0] != '0'; }
  // the weight file -> ONE device image that every engine of the pool adopts in place (pf_engine_config.weights_device)
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw Error(PF_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    PF_CHECK(device >= 0 && device < ndev, PF_ERR_DEVICE, "device ordinal out of range");
    std::vector<char> file;
    read_binary_file(model_path_.c_str(), file);
    PF_CHECK(file.size() >= 16, PF_ERR_FORMAT, "weights: image too small");
    PF_HIP(hipSetDevice(device_));
    void* img = nullptr;
    PF_HIP(hipMalloc(&img, file.size()));
    const int dev = device_;
    image_ = std::shared_ptr<void>(img, [dev](void* p) { hipSetDevice(dev); hipFree(p); });
    image_bytes_ = (int64_t)file.size();
    if (hipMemcpy(img, file.data(), file.size(), hipMemcpyHostToDevice) != hipSuccess)
      throw Error(PF_ERR_DEVICE, "weights: upload failed");       // (image_ frees itself)
  }
  ec.weights_device = image_.get();
  ec.weights_bytes = image_bytes_;
  { const char* e = getenv("PF_RECOGNIZER_AUDIO_CACHE_MB"); if (e && e[0]) audio_cache_cap_ = (size_t)std::max(0, atoi(e)) << 20; }
  { const char* e = getenv("PF_RECOGNIZER_STAGING_MB"); if (e && e[0]) staging_bytes_ = (size_t)std::min(1024, std::max(0, atoi(e))) << 20; }
  { const char* e = getenv("PF_RECOGNIZER_STAGGER"); if (e && e[0]) stagger_frac_ = std::max(0.f, std::min(0.9f, (float)atof(e))); }
  { const char* e = getenv("PF_RECOGNIZER_STAGING_POLICY"); staging_always_ = e && std::string(e) == "always"; }
  { const char* e = getenv("PF_RECOGNIZER_COPY_THREADS"); if (e && e[0]) crew_threads_ = std::min(15, std::max(0, atoi(e))); }
  crew_threads_ = std::min<int>(crew_threads_, std::max(0, (int)std::thread::hardware_concurrency() - 1));
  { const char* e = getenv("PF_RECOGNIZER_STAGING_PIECE_KB"); if (e && e[0]) staging_piece_ = (size_t)std::max(64, atoi(e)) << 10; }
  engines_.push_back(make_engine());
  busy_.push_back(0);
  engine_kind_ = engines_[0]->model().kind;
  sv_device_prompt_ = engines_[0]->has_device_prompt();
  if (engine_kind_ == "sensevoicesmall") { sv_embed_ = engines_[0]->embed_table(); sv_use_itn_ = engines_[0]->model().use_itn; }
  feat_m_ = (conf_.lfr_m != 1 || conf_.lfr_n != 1) ? conf_.lfr_m : 1;
  // the device form of a stream needs the batched front-end to compute exactly what AddSamples + PadSequence would
  device_streams_ = device_streams_ && engines_[0]->staged_frontend_matches_host();
  uid_ = register_recognizer();
}

std::shared_ptr<Engine> Recognizer::make_engine() {
  std::shared_ptr<void> img = image_;                // the engine points into the image: it must outlive the engine
  return std::shared_ptr<Engine>(new Engine(ec_), [img](Engine* e) { delete e; });
}

int Recognizer::feature_floats(int64_t n) {
  std::shared_ptr<Engine> e = engine();
  if (!e) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");
  return e->num_lfr_frames(n) * feat_m_ * conf_.n_mels;
}

Recognizer::Lease& Recognizer::Lease::operator=(Lease&& o) noexcept {
  if (this != &o) {
    release();
    r_ = o.r_; idx_ = o.idx_; e_ = std::move(o.e_); lk_ = std::move(o.lk_);
    o.r_ = nullptr; o.idx_ = -1;
  }
  return *this;
}
void Recognizer::Lease::release() {
  if (!r_) return;
  if (lk_.owns_lock()) lk_.unlock();
  {
    std::lock_guard<std::mutex> g(r_->mu_);
    if (idx_ >= 0 && idx_ < (int)r_->busy_.size()) r_->busy_[(size_t)idx_] = 0;
  }
  r_->cv_.notify_all();
  e_.reset();
  r_ = nullptr; idx_ = -1;
}

Recognizer::Lease Recognizer::acquire() {
  std::unique_lock<std::mutex> lk(mu_);
  for (;;) {
    if (disposed_ || engines_.empty()) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");
    // a free engine whose mutex is free too: users of pf_recognizer_engine lock engine 0 without a lease, and a call must
    // not queue behind them while another engine idles or the pool may still grow
    int idx = -1, held = -1;
    std::unique_lock<std::mutex> elk;
    for (size_t i = 0; i < engines_.size(); ++i) {
      if (busy_[i]) continue;
      std::unique_lock<std::mutex> t(engines_[i]->mutex(), std::try_to_lock);
      if (t.owns_lock()) { idx = (int)i; elk = std::move(t); break; }
      if (held < 0) held = (int)i;
    }
    const bool may_grow = (int)engines_.size() + creating_ < max_engines_;
    if (idx < 0 && held >= 0 && !may_grow) idx = held;          // nothing else to take: queue on that engine's mutex below
    if (idx < 0 && may_grow) {
      // every engine is busy and the pool may grow: build one outside the lock (weight conversion takes ~1 s)
      ++creating_;
      lk.unlock();
      std::shared_ptr<Engine> ne;
      try {
        PF_HIP(hipSetDevice(device_));
        ne = make_engine();
      } catch (...) {
        lk.lock();
        --creating_;
        cv_.notify_all();
        throw;
      }
      lk.lock();
      --creating_;
      cv_.notify_all();                   // Dispose() waits for creating_ == 0 (this branch used to leave it asleep)
      if (disposed_) { lk.unlock(); ne.reset(); lk.lock(); continue; }
      engines_.push_back(ne);
      busy_.push_back(0);
      idx = (int)engines_.size() - 1;
    }
    if (idx >= 0) {
      busy_[(size_t)idx] = 1;
      Lease l;
      l.r_ = this; l.idx_ = idx; l.e_ = engines_[(size_t)idx];
      lk.unlock();
      if (elk.owns_lock()) l.lk_ = std::move(elk);
      else l.lk_ = std::unique_lock<std::mutex>(l.e_->mutex());     // also serialises with users of pf_recognizer_engine
      return l;
    }
    cv_.wait(lk);
  }
}

float* Recognizer::audio_alloc(size_t bytes, size_t* got) {
  const size_t cls = (size_t)round_up((int64_t)std::max<size_t>(bytes, 4), (int64_t)(256 << 10));   // 256 KiB classes
  *got = cls;
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (disposed_) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");
    auto it = audio_cache_.find(cls);
    if (it != audio_cache_.end() && !it->second.empty()) {
      float* p = it->second.back();
      it->second.pop_back();
      audio_cached_bytes_ -= cls;
      return p;
    }
  }
  void* p = nullptr;
  PF_HIP(hipSetDevice(device_));
  PF_HIP(hipMalloc(&p, cls));
  return (float*)p;
}

void Recognizer::audio_free(float* p, size_t bytes) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (!disposed_ && audio_cached_bytes_ + bytes <= audio_cache_cap_) {          // keep up to 1 GiB (PF_RECOGNIZER_AUDIO_CACHE_MB) for re-use
      audio_cache_[bytes].push_back(p);
      audio_cached_bytes_ += bytes;
      return;
    }
  }
  hipSetDevice(device_);
  hipFree(p);
}

static std::atomic<long long> g_up_ns{0}, g_up_copy_ns{0}, g_up_wait_ns{0}, g_up_calls{0};
static const int kUpTiming = env_int("PF_UPLOAD_TIMING", 0);
struct UpTimer {                                         // PF_UPLOAD_TIMING=1: where an upload's host time goes (tools/r6_staging_ab.sh)
  std::atomic<long long>& acc; std::chrono::steady_clock::time_point t0;
  explicit UpTimer(std::atomic<long long>& a) : acc(a) { if (kUpTiming) t0 = std::chrono::steady_clock::now(); }
  ~UpTimer() { if (kUpTiming) acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

bool Recognizer::seen_before(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(seen_mu_);
  if (seen_.size() >= 16384) seen_.clear();
  auto it = seen_.find(p);
  if (it != seen_.end() && it->second == bytes) return true;
  seen_[p] = bytes;
  return false;
}

void Recognizer::upload(float* dst, const float* src, size_t bytes, hipEvent_t* ev, bool* pending) {
  UpTimer whole(g_up_ns);
  if (kUpTiming && (++g_up_calls % 256) == 0)
    fprintf(stderr, "[upload] %lld calls: %.1f us per call, host copy %.1f, ring waits %.1f\n", (long long)g_up_calls, g_up_ns / 1e3 / g_up_calls,
            g_up_copy_ns / 1e3 / g_up_calls, g_up_wait_ns / 1e3 / g_up_calls);
  // a lane nobody is copying through right now, else the next in turn
  const unsigned first = next_lane_.fetch_add(1);
  std::unique_lock<std::mutex> lk;
  CopyLane* lnp = nullptr;
  for (unsigned i = 0; i < lanes_.size() && !lnp; ++i) {
    CopyLane& c = lanes_[(first + i) % lanes_.size()];
    std::unique_lock<std::mutex> t(c.mu, std::try_to_lock);
    if (t.owns_lock()) { lk = std::move(t); lnp = &c; }
  }
  if (!lnp) { lnp = &lanes_[first % lanes_.size()]; lk = std::unique_lock<std::mutex>(lnp->mu); }
  CopyLane& ln = *lnp;
  if (disposed_) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");     // (Dispose destroys the lanes under this lock)
  PF_HIP(hipSetDevice(device_));
  if (!ln.s) PF_HIP(hipStreamCreateWithFlags(&ln.s, hipStreamNonBlocking));
  const bool staged = ev && staging_bytes_ > 0 && (staging_always_ || !seen_before(src, bytes));
  if (staged && !ln.tried) {
    ln.tried = true;
    void* p = nullptr;
    if (hipHostMalloc(&p, staging_bytes_, hipHostMallocDefault) == hipSuccess) { ln.pin = (char*)p; ln.cap = staging_bytes_; }
    else (void)hipGetLastError();                                       // no pinned memory: the runtime's pageable copy below
  }
  if (!staged || !ln.pin) {
    PF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ln.s));
    PF_HIP(hipStreamSynchronize(ln.s));                // the samples are on the device when AddSamples returns
    return;
  }
  if (!ln.sp) {
    // The DMAs of a staged upload are copy KERNELS on this part: on a plain stream they queue behind the other callers' compute
    // kernels and land 0.3 ms late each (the ring fills, AddSamples waits: 10.6 - 11.0 ms per batch with 4 callers against 9.3 - 9.6).
    // The staged path therefore has a stream of its own at the device's highest priority; the runtime's own path keeps the plain
    // one (its synchronous copies were 3 us slower there).  PF_RECOGNIZER_LANE_PRIORITY=0: a plain stream for both.
    static const int prio = env_int("PF_RECOGNIZER_LANE_PRIORITY", 1);
    int lo = 0, hi = 0;
    if (prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) {
      if (hipStreamCreateWithPriority(&ln.sp, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); ln.sp = nullptr; }
    }
    if (!ln.sp) PF_HIP(hipStreamCreateWithFlags(&ln.sp, hipStreamNonBlocking));
  }
  if (crew_threads_ > 0) std::call_once(crew_once_, [&] { crew_.reset(new CopyCrew(crew_threads_)); });
  // staged: piece by piece through the pinned ring; the DMA of piece i runs under the host copy of piece i + 1
  const size_t piece = std::min(staging_piece_, ln.cap);
  for (size_t done = 0; done < bytes;) {
    const size_t nb = std::min(piece, bytes - done);
    if (ln.head + nb > ln.cap) ln.head = 0;
    const size_t off = ln.head;
    // pieces still in flight over [off, off + nb): the ring is FIFO on one stream, so everything up to the last overlap is done
    // once that one is
    int last = -1;
    for (int i = 0; i < (int)ln.inflight.size(); ++i) {
      const CopyLane::Piece& q = ln.inflight[i];
      if (q.off < off + nb && off < q.off + q.bytes) last = i;
    }
    if (last >= 0) {
      UpTimer w(g_up_wait_ns);
      PF_HIP(hipEventSynchronize(ln.inflight[last].ev));
      for (int i = 0; i <= last; ++i) { ln.spare.push_back(ln.inflight.front().ev); ln.inflight.pop_front(); }
    }
    {
      UpTimer c(g_up_copy_ns);
      if (crew_) crew_->copy(ln.pin + off, (const char*)src + done, nb);
      else std::memcpy(ln.pin + off, (const char*)src + done, nb);
    }
    PF_HIP(hipMemcpyAsync((char*)dst + done, ln.pin + off, nb, hipMemcpyHostToDevice, ln.sp));
    hipEvent_t pe = nullptr;
    if (!ln.spare.empty()) { pe = ln.spare.back(); ln.spare.pop_back(); }
    else PF_HIP(hipEventCreateWithFlags(&pe, hipEventDisableTiming));
    if (hipEventRecord(pe, ln.sp) != hipSuccess) { ln.spare.push_back(pe); PF_HIP(hipStreamSynchronize(ln.sp)); }
    else ln.inflight.push_back({off, nb, pe});
    ln.head = off + nb;
    done += nb;
  }
  if (!*ev) PF_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  PF_HIP(hipEventRecord(*ev, ln.sp));
  *pending = true;
}

std::shared_ptr<Stream> Recognizer::CreateOfflineStream() {
  if (disposed_) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");   // ObjectDisposedException
  return std::make_shared<Stream>(shared_from_this());
}

void Recognizer::adopt(Stream* s) {
  if (s->owner) return;
  const ConfEntity& a = s->uconf; const ConfEntity& b = conf_;
  const bool same = a.fs == b.fs && a.n_mels == b.n_mels && a.lfr_m == b.lfr_m && a.lfr_n == b.lfr_n &&
                    a.snip_edges == b.snip_edges && a.dither == b.dither && a.window == b.window &&
                    s->ushift == cmvn_shift_ && s->uscale == cmvn_scale_;
  PF_CHECK(same, PF_ERR_UNSUPPORTED,
           "OfflineStream(mvnFilePath, confEntity): the stream's front-end differs from the recognizer's (am.mvn values or "
           "frontend_conf); create it with CreateOfflineStream or with the recognizer's own files");
  s->owner = shared_from_this();
  std::vector<std::vector<float>> calls;
  calls.swap(s->pending);
  s->has_speech = false;
  s->SpeechLength = 0;
  static const float kNone = 0.f;
  for (auto& c : calls) s->AddSamples(c.empty() ? &kNone : c.data(), (int64_t)c.size());
}

void Recognizer::Dispose() {
  std::vector<std::shared_ptr<Engine>> es;
  std::map<size_t, std::vector<float*>> cache;
  {
    std::unique_lock<std::mutex> lk(mu_);
    if (disposed_.exchange(true)) return;
    cv_.notify_all();
    // calls in flight hold a lease: wait for them (and for engines still being built), then take the pool
    cv_.wait(lk, [&] {
      if (creating_ > 0) return false;
      for (char b : busy_) if (b) return false;
      return true;
    });
    es.swap(engines_);
    busy_.clear();
    cache.swap(audio_cache_);
    audio_cached_bytes_ = 0;
  }
  // a user of pf_recognizer_engine may still hold engine 0's mutex: wait for it so that Dispose returns with the device
  // idle, then drop ours (the last owner frees the device memory)
  for (auto& e : es) { std::lock_guard<std::mutex> lk(e->mutex()); }
  es.clear();                                      // tokens_ stays: a concurrent GetResults may still be decoding
  hipSetDevice(device_);
  for (auto& kv : cache)
    for (float* p : kv.second) hipFree(p);
  free_device_side();
}

void Recognizer::free_device_side() {
  hipSetDevice(device_);
  for (auto& kv : audio_cache_)
    for (float* p : kv.second) hipFree(p);
  audio_cache_.clear();
  audio_cached_bytes_ = 0;
  for (CopyLane& ln : lanes_) {
    std::lock_guard<std::mutex> lk(ln.mu);
    if (ln.s) { hipStreamSynchronize(ln.s); hipStreamDestroy(ln.s); ln.s = nullptr; }
    if (ln.sp) { hipStreamSynchronize(ln.sp); hipStreamDestroy(ln.sp); ln.sp = nullptr; }
    for (auto& q : ln.inflight) hipEventDestroy(q.ev);
    for (hipEvent_t e : ln.spare) hipEventDestroy(e);
    ln.inflight.clear(); ln.spare.clear();
    if (ln.pin) { hipHostFree(ln.pin); ln.pin = nullptr; ln.cap = 0; }
  }
  crew_.reset();                                   // (every lane has been through its lock above: no upload is using the helpers)
  image_.reset();                                  // freed when the last engine that adopted it is gone
}

// PF_RECOGNIZER_TIMING=1: where a GetResults call spends its time on the host (printed every 32 calls)
static const int kFwdTiming = env_int("PF_RECOGNIZER_TIMING", 0);
static std::atomic<long long> g_fwd_ns[8];
static std::atomic<long long> g_fwd_calls{0};
struct FwdClock {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(int i) {
    if (!kFwdTiming) return;
    const auto n = std::chrono::steady_clock::now();
    g_fwd_ns[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count();
    t = n;
  }
};
static void fwd_report() {
  if (!kFwdTiming) return;
  const long long c = ++g_fwd_calls;
  if (c % 32) return;
  static const char* names[8] = {"acquire", "hotwords+waits", "stage_audio", "enqueue", "first fetch (sync)", "second fetch", "tokens+timestamps+RemoveChunk", "text"};
  fprintf(stderr, "[GetResults] %lld calls:", c);
  for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f us |", names[i], g_fwd_ns[i] / 1e3 / c);
  fprintf(stderr, "\n");
}

// Two steps that start together on two engines finish together, and their callers then upload together — with the GPU idle — and
// start together again: a convoy that cost two callers on two engines 1.5 - 2 ms per cycle (10.0 - 10.4 ms per batch where four callers
// reached 8.6).  Any offset between the engines persists once it exists, and the best one is about half a step (the other caller's
// uploads and text stage then sit in the middle of this caller's kernels): a step does not start within stagger_frac_ x (the recent
// duration of a step) of the previous start on another engine of the pool — the caller sleeps the difference, once.  Same box,
// 32 x 30 s: 2 callers 10.0 -> 8.6 ms per batch, 3 callers 8.56 -> 8.39, 4 callers 8.56 -> 8.50 (PF_RECOGNIZER_STAGGER, default
// 0.35; 0 = off; profiles/round6_staging_ab.txt, run 8).
void Recognizer::stagger_start() {
  if (stagger_frac_ <= 0.f) return;
  using clk = std::chrono::steady_clock;
  long long wait_us = 0;
  {
    std::lock_guard<std::mutex> lk(mu_);
    // (N engines evenly apart are a step / N apart: the fraction is quoted for a pool of two)
    const double n = (double)std::max<size_t>(2, engines_.size());
    const long long gap_us = std::min<long long>(10000, (long long)(stagger_frac_ * 2.0 / n * step_ema_us_));
    const auto now = clk::now();
    const long long since = std::chrono::duration_cast<std::chrono::microseconds>(now - last_start_).count();
    if (gap_us > 0 && since >= 0 && since < gap_us) wait_us = gap_us - since;
    last_start_ = now + std::chrono::microseconds(wait_us);
  }
  if (wait_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(wait_us));
}

void Recognizer::note_step(std::chrono::steady_clock::time_point t0) {
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  std::lock_guard<std::mutex> lk(mu_);
  step_ema_us_ = step_ema_us_ <= 0.0 ? us : 0.75 * step_ema_us_ + 0.25 * us;
}

void Recognizer::Forward(const std::vector<Stream*>& streams) {
  if (streams.empty()) return;                                      // :120-123
  try {
    if (disposed_) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");
    for (Stream* s : streams) adopt(s);
    // the fast form needs every stream in the device form; otherwise every stream is brought to the host form first
    bool all_dev = device_streams_;
    for (Stream* s : streams) all_dev = all_dev && s->device_form;
    const bool sv = engine_kind_ == "sensevoicesmall";
    if (all_dev && sv && !sv_device_prompt_) all_dev = false;
    if (!all_dev)
      for (Stream* s : streams) s->materialize();
    FwdClock fc;
    std::chrono::steady_clock::time_point t_step;
    bool timed_step = false;
    Lease lease = acquire();
    Engine* e = lease.get();
    const ModelCfg& mc = e->model();
    const int W = mc.feat_dim;
    const int B = (int)streams.size();
    fc.lap(0);
    if (!all_dev && sv) {
      // SenseVoice split-embed variant: prepend [emb(lang), emb(1), emb(2), emb(textnorm)] to Speech
      // IN PLACE (OfflineProjOfSenseVoiceSmall.cs:78-106, quirk Q8); effective ids per quirk Q7.
      const std::vector<float>& emb = e->embed_table();
      PF_CHECK(emb.size() >= (size_t)16 * W, PF_ERR_FORMAT, "sensevoice: embed table missing from the container");
      const int languageId = mc.use_itn ? 14 : 15, textnormId = 15;
      const int order[4] = {languageId, 1, 2, textnormId};
      for (Stream* s : streams) {
        if (!s->has_speech) continue;
        std::vector<float> sp((size_t)4 * W + s->Speech.size());
        for (int r = 0; r < 4; ++r) std::memcpy(&sp[(size_t)r * W], &emb[(size_t)order[r] * W], (size_t)W * 4);
        std::memcpy(sp.data() + (size_t)4 * W, s->Speech.data(), s->Speech.size() * 4);
        s->Speech.swap(sp);
        s->SpeechLength = (int)s->Speech.size();
      }
    }
    for (Stream* s : streams)
      // PadSequence dereferences a null Speech -> NullReferenceException inside Forward's try
      if (!s->has_speech || s->SpeechLength == 0) throw Error(PF_ERR_RECOGNITION, "Object reference not set (Speech is null)");
    if (mc.seaco) {
      // OfflineProjOfSeacoParaformer.cs:52-60: hotwords = SelectMany over the streams' Hotwords (a null list
      // throws inside ModelProj -> "Offline recognition failed"); empty -> the constructor's default list
      // (hotword file + [sos_eos_id] terminator, OfflineRecognizer.cs:72-90).  PadList(…, 0, 10) (EmbedSeacoModel.cs).
      std::vector<std::vector<int32_t>> hw;
      for (Stream* s : streams) {
        if (s->hotwords_null) throw Error(PF_ERR_RECOGNITION, "Value cannot be null (Hotwords)");
        for (auto& w : s->Hotwords) hw.push_back(w);
      }
      if (hw.empty()) hw = hotwords_;
      std::vector<int32_t> pad;
      for (auto& w : hw)
        for (int j = 0; j < 10; ++j) pad.push_back(j < (int)w.size() ? w[j] : 0);
      e->set_hotwords(pad.data(), (int)hw.size());
    }
    e->drop_thread_result();              // this call's result is read back under the same lease, not from a slot
    if (all_dev) {
      // the batched front-end over the samples where they are (fbank -> LFR + CMVN + PadSequence [+ the SenseVoice query
      // rows]) and the model, one stream of launches: WavFrontend.cs:31-111 + Utils/PadHelper.cs:25 + ModelProj
      std::vector<const float*> ptrs;
      std::vector<int64_t> ns;
      PF_HIP(hipSetDevice(device_));
      for (Stream* s : streams) {
        ptrs.push_back(s->dev_audio); ns.push_back(s->dev_n);
        // staged uploads may still be landing: the engine's stream waits for them, the host does not
        if (s->dev_ev_pending) PF_HIP(hipStreamWaitEvent(e->stream(), s->dev_ev, 0));
      }
      fc.lap(1);
      stagger_start();
      t_step = std::chrono::steady_clock::now();
      timed_step = true;
      e->stage_device_audio(ptrs.data(), ns.data(), B);
      fc.lap(2);
      e->run_staged(false);
      fc.lap(3);
    } else {
      std::vector<const float*> ptrs;
      std::vector<int32_t> lens;
      for (Stream* s : streams) { ptrs.push_back(s->Speech.data()); lens.push_back((int32_t)s->Speech.size()); }
      e->model_proj_host(ptrs.data(), lens.data(), B, false);
    }
    pf_batch_out out;
    std::memset(&out, 0, sizeof(out));
    out.struct_size = sizeof(out);
    e->fetch(&out);                       // sync; learn L
    if (timed_step) note_step(t_step);
    fc.lap(4);
    const int L = out.L;
    std::vector<int64_t> ids((size_t)B * std::max(L, 1));
    out.token_ids = ids.data();
    out.l_cap = std::max(L, 1);
    const int P = out.cif_peak_len;                                    // 3*Tmax for timestamp models (:172-183)
    std::vector<float> peak((size_t)B * std::max(P, 1));
    if (P > 0) { out.cif_peak = peak.data(); out.cif_peak_cap = (int64_t)peak.size(); }
    e->fetch(&out);
    fc.lap(5);
    lease.release();                      // the device work of this call is over: the text stage below needs no engine
    for (int b = 0; b < B; ++b) {
      Stream* s = streams[b];
      s->Tokens.assign(ids.begin() + (size_t)b * out.l_cap, ids.begin() + (size_t)b * out.l_cap + L);   // :187
      s->Timestamps.reserve(s->Timestamps.size() + (size_t)L);
      if (P > 0) {
        // :172-183: the peak row and ALL L arg-max ids go to time_stamp_lfr6_onnx
        TsList ts = time_stamp_lfr6(peak.data() + (size_t)b * P, P, s->Tokens);
        for (auto& t2 : ts) s->Timestamps.push_back(t2);
      } else {
        for (int l = 0; l < L; ++l) s->Timestamps.push_back({0, 0});                                    // :151,:188
      }
      if (all_dev && sv && s->Tokens.size() <= 2) {
        // quirk Q8 on the device form: the reference has prepended the query rows to Speech IN PLACE, and a stream whose
        // chunk RemoveChunk keeps (at most two ids) carries them into its next call: give it the host form with them
        s->materialize();
        const std::vector<float>& emb = sv_embed_;
        const int order[4] = {sv_use_itn_ ? 14 : 15, 1, 2, 15};
        std::vector<float> sp((size_t)4 * W + s->Speech.size());
        for (int r = 0; r < 4; ++r) std::memcpy(&sp[(size_t)r * W], &emb[(size_t)order[r] * W], (size_t)W * 4);
        std::memcpy(sp.data() + (size_t)4 * W, s->Speech.data(), s->Speech.size() * 4);
        s->Speech.swap(sp);
        s->SpeechLength = (int)s->Speech.size();
      }
      s->RemoveChunk();                                                                                  // :189
    }
    fc.lap(6);
  } catch (const Error& ex) {
    if (ex.code == PF_ERR_RECOGNITION) throw;
    throw Error(PF_ERR_RECOGNITION, std::string("Offline recognition failed: ") + ex.what());           // :194-197
  }
}

Recognizer::~Recognizer() {
  // never disposed (or the constructor threw after the image was uploaded): nobody else can reach this object any more,
  // so the engines, the audio cache, the copy lanes and the image go here (~0.9 GB used to leak)
  if (!disposed_.exchange(true)) { engines_.clear(); busy_.clear(); free_device_side(); }
  std::lock_guard<std::mutex> lk(g_live_mu);
  g_live.erase(uid_);
}

void Recognizer::GetResults(const std::vector<Stream*>& streams) {
  Forward(streams);
  FwdClock fc;
  std::vector<ResultEntity> out;
  for (Stream* s : streams) out.push_back(decode_multi_one(token_table_, s->Tokens, s->Timestamps));
  fc.lap(7);
  fwd_report();
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (auto it = t_results.begin(); it != t_results.end();)
      it = g_live.count(it->first) ? std::next(it) : t_results.erase(it);
  }
  t_results[uid_] = std::move(out);
}

const std::vector<ResultEntity>& Recognizer::results_of_this_thread() {
  static const std::vector<ResultEntity> kEmpty;
  auto it = t_results.find(uid_);
  return it == t_results.end() ? kEmpty : it->second;
}

}  // namespace pf
