// recognizer.cpp — see recognizer.h.  Host-side logic of the reference's public API, restated
// in C++ (the reference is compiled C#; no .NET toolchain exists in the build image).
#include "recognizer.h"

#include <algorithm>
#include <array>
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
  const std::vector<uint32_t> cps = utf8_decode(s);
  if (cps.empty()) return false;
  for (uint32_t c : cps)
    if (c < 0x4e00 || c > 0x9fa5) return false;
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

ResultEntity decode_multi_one(const std::vector<std::string>& tokens, const std::vector<int64_t>& ids,
                              const std::vector<std::vector<int32_t>>& timestamps) {
  ResultEntity r;
  std::string text, lastToken;
  bool haveLastTs = false;
  std::vector<int32_t> lastTs;
  const size_t n = std::min(ids.size(), timestamps.size());    // Zip
  for (size_t i = 0; i < n; ++i) {
    const int64_t token = ids[i];
    const std::vector<int32_t>& ts = timestamps[i];
    if (token == 2) break;
    if (token < 0 || token >= (int64_t)tokens.size()) throw Error(PF_ERR_RECOGNITION, "token id out of range");
    std::string cur = tokens[(size_t)token];
    const size_t tab = cur.find('\t');
    if (tab != std::string::npos) cur = cur.substr(0, tab);
    if (cur == "</s>" || cur == "<s>" || cur == "<blank>" || cur == "<unk>") continue;
    if (is_chinese_all(cur)) {
      text += cur;
      r.Tokens.push_back(cur);
      r.Timestamps.push_back(ts);
      continue;
    }
    text += kBar + cur + kBar;
    const std::string comb = lastToken + kBar + cur + kBar;
    auto merged_ts = [&]() {
      if (!haveLastTs) return ts;
      std::vector<int32_t> t = lastTs;
      t.insert(t.end(), ts.begin(), ts.end());
      return t;
    };
    if (index_gt0(comb, "@@" + kBar + kBar)) {
      const std::string curToken = replaced(comb, "@@" + kBar + kBar, "");
      const std::vector<int32_t> curTs = merged_ts();
      remove_first_equal_to_last(r.Tokens);
      r.Tokens.push_back(replaced(curToken, kBar, ""));
      if (r.Timestamps.empty()) throw Error(PF_ERR_RECOGNITION, "Sequence contains no elements");
      r.Timestamps.pop_back();                 // arrays compare by reference in C#: the last one
      r.Timestamps.push_back(curTs);
      lastToken = curToken; lastTs = curTs; haveLastTs = true;
    } else if ((count_sub(comb, kBar) == 3 || count_sub(comb, kBar) == 5) && index_lt0(comb, kBar + kBar + kBar)) {
      const std::string curToken = replaced(comb, kBar + kBar, "");
      const std::vector<int32_t> curTs = merged_ts();
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

std::vector<std::vector<int32_t>> time_stamp_lfr6(const float* us_cif_peak, int n, std::vector<int64_t> tokens) {
  // float32 arithmetic throughout, (int)(t*1000) truncation (quirk Q10)
  const int START_END_THRESHOLD = 5, MAX_TOKEN_DURATION = 30;
  volatile float tr0 = 10.0f * 6;
  volatile float tr1 = tr0 / 1000;
  const float TIME_RATE = tr1 / 3;
  const float total_offset = -1.5f;
  const int num_frames = n;
  if (tokens.empty()) throw Error(PF_ERR_RECOGNITION, "Sequence contains no elements");   // tokens.Last()
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
  std::vector<std::vector<int32_t>> out;
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

void Stream::AddSamples(const float* samples, int64_t n) {
  if (disposed) throw Error(PF_ERR_DISPOSED, "OfflineStream");
  if (!samples) throw Error(PF_ERR_NULL_SAMPLES, "source");       // ArgumentNullException("source")
  std::shared_ptr<Engine> e = owner->engine();                     // keeps the engine alive across this call
  if (!e) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");
  std::lock_guard<std::mutex> lk(e->mutex());                      // process-wide lock in the reference
  std::vector<float> feats;
  int t = 0;
  e->frontend_host(samples, n, feats, t);
  Speech.insert(Speech.end(), feats.begin(), feats.end());
  has_speech = true;
  SpeechLength = (int)Speech.size();
}

void Stream::Dispose() {
  disposed = true;
  std::vector<float>().swap(Speech);
  std::vector<int64_t>().swap(Tokens);
  std::vector<std::vector<int32_t>>().swap(Timestamps);
  std::vector<std::vector<int32_t>>().swap(Hotwords);
  has_speech = false;
  SpeechLength = 0;
}

void Stream::RemoveChunk() {
  if (Tokens.size() > 2) {
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
  }
  if (tokens_.empty()) throw Error(PF_ERR_TOKENS, "tokens invalid");
  if (!hotword.empty() && file_exists(hotword))                    // :34-38, :75
    hotwords_ = hotword_ids(tokens_, read_lines(hotword), 1);
  pf_engine_config ec;
  std::memset(&ec, 0, sizeof(ec));
  ec.struct_size = sizeof(ec);
  ec.device = device;
  ec.weights_path = model.c_str();
  ec.mvn_path = mvn.c_str();
  ec.fs = conf_.fs; ec.n_mels = conf_.n_mels; ec.lfr_m = conf_.lfr_m; ec.lfr_n = conf_.lfr_n;
  ec.snip_edges = conf_.snip_edges ? 1 : 0;
  ec.dither = conf_.dither;
  // frame_length / frame_shift of the yaml are NOT forwarded: WavFrontend.cs:21-27 builds OnlineFbank from dither,
  // snip_edges, window, fs and n_mels only, so whatever the file says the reference frames at the kaldi defaults
  // (25 ms / 10 ms) — a conf with other values must run here exactly as it does there
  ec.frame_length_ms = 0;
  ec.frame_shift_ms = 0;
  ec.window = conf_.window.c_str();
  ec.use_itn = conf_.use_itn ? 1 : 0;
  // `-accuracy int8` (the reference CLI's default, Examples/Program.cs:98-101) selects the arithmetic through the FILE
  // NAME there (model.int8.onnx vs model.onnx, Examples/OfflineAliParaformerAsrRecognizer.cs:17-22); the same
  // convention here: a container named *.int8.* runs with dynamically quantised Linear layers (math_mode 2)
  {
    const size_t sl = model.find_last_of("/\\");
    const std::string fname = sl == std::string::npos ? model : model.substr(sl + 1);
    if (fname.find(".int8.") != std::string::npos) ec.math_mode = 2;
  }
  engine_ = std::make_shared<Engine>(ec);
  uid_ = register_recognizer();
}

std::shared_ptr<Stream> Recognizer::CreateOfflineStream() {
  if (disposed_) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");   // ObjectDisposedException
  return std::make_shared<Stream>(shared_from_this());
}

void Recognizer::Dispose() {
  std::shared_ptr<Engine> e;
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (disposed_.exchange(true)) return;
    e.swap(engine_);
  }
  // a call that fetched the engine before this point still owns a reference; wait for it on the engine mutex so
  // that Dispose returns with the device idle, then drop ours (the last owner frees the device memory)
  if (e) { std::lock_guard<std::mutex> lk(e->mutex()); }
  e.reset();                                       // tokens_ stays: a concurrent GetResults may still be decoding
}

void Recognizer::Forward(const std::vector<Stream*>& streams) {
  if (streams.empty()) return;                                      // :120-123
  try {
    std::shared_ptr<Engine> eh = engine();
    if (disposed_ || !eh) throw Error(PF_ERR_DISPOSED, "OfflineRecognizer");
    Engine* e = eh.get();
    std::lock_guard<std::mutex> lk(e->mutex());
    const ModelCfg& mc = e->model();
    const int W = mc.feat_dim;
    // SenseVoice split-embed variant: prepend [emb(lang), emb(1), emb(2), emb(textnorm)] to Speech
    // IN PLACE (OfflineProjOfSenseVoiceSmall.cs:78-106, quirk Q8); effective ids per quirk Q7.
    if (mc.kind == "sensevoicesmall") {
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
    std::vector<const float*> ptrs;
    std::vector<int32_t> lens;
    for (Stream* s : streams) {
      // PadSequence dereferences a null Speech -> NullReferenceException inside Forward's try
      if (!s->has_speech || s->Speech.empty()) throw Error(PF_ERR_RECOGNITION, "Object reference not set (Speech is null)");
      ptrs.push_back(s->Speech.data());
      lens.push_back((int32_t)s->Speech.size());
    }
    const int B = (int)streams.size();
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
    e->drop_thread_result();              // this call's result is read back under the same lock, not from a slot
    e->model_proj_host(ptrs.data(), lens.data(), B, false);
    pf_batch_out out;
    std::memset(&out, 0, sizeof(out));
    out.struct_size = sizeof(out);
    e->fetch(&out);                       // sync; learn L
    const int L = out.L;
    std::vector<int64_t> ids((size_t)B * std::max(L, 1));
    out.token_ids = ids.data();
    out.l_cap = std::max(L, 1);
    const int P = out.cif_peak_len;                                    // 3*Tmax for timestamp models (:172-183)
    std::vector<float> peak((size_t)B * std::max(P, 1));
    if (P > 0) { out.cif_peak = peak.data(); out.cif_peak_cap = (int64_t)peak.size(); }
    e->fetch(&out);
    for (int b = 0; b < B; ++b) {
      Stream* s = streams[b];
      s->Tokens.assign(ids.begin() + (size_t)b * out.l_cap, ids.begin() + (size_t)b * out.l_cap + L);   // :187
      if (P > 0) {
        // :172-183: the peak row and ALL L arg-max ids go to time_stamp_lfr6_onnx
        std::vector<std::vector<int32_t>> ts = time_stamp_lfr6(peak.data() + (size_t)b * P, P, s->Tokens);
        for (auto& t2 : ts) s->Timestamps.push_back(t2);
      } else {
        for (int l = 0; l < L; ++l) s->Timestamps.push_back({0, 0});                                    // :151,:188
      }
      s->RemoveChunk();                                                                                  // :189
    }
  } catch (const Error& ex) {
    if (ex.code == PF_ERR_RECOGNITION) throw;
    throw Error(PF_ERR_RECOGNITION, std::string("Offline recognition failed: ") + ex.what());           // :194-197
  }
}

Recognizer::~Recognizer() {
  std::lock_guard<std::mutex> lk(g_live_mu);
  g_live.erase(uid_);
}

void Recognizer::GetResults(const std::vector<Stream*>& streams) {
  Forward(streams);
  std::vector<ResultEntity> out;
  for (Stream* s : streams) out.push_back(decode_multi_one(tokens_, s->Tokens, s->Timestamps));
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
