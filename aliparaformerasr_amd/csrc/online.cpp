// online.cpp — see online.h.  Host-side logic of the reference's streaming API restated in C++ (the reference is
// compiled C#; no .NET toolchain exists in the build image); citations are relative to AliParaformerAsr/.
#include "online.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace pf {

// ------------------------------------------------------------------ pure host pieces -------
std::vector<float> online_apply_lfr(const std::vector<float>& in, int n_mels, int lfr_m, int lfr_n) {
  // OnlineWavFrontend.cs:63-80
  const int t = (int)(in.size() / (size_t)n_mels);
  int t_lfr = 0;
  if (t % lfr_n < lfr_m - lfr_n) t_lfr = t / lfr_n - 1;
  if (t % lfr_n >= lfr_m - lfr_n) t_lfr = t / lfr_n;
  if (t_lfr < 0) throw Error(PF_ERR_RECOGNITION, "Arithmetic operation resulted in an overflow (negative LFR frame count)");
  std::vector<float> out((size_t)t_lfr * lfr_m * n_mels);
  for (int i = 0; i < t_lfr; ++i) {
    const size_t src = (size_t)i * lfr_n * n_mels, len = (size_t)lfr_m * n_mels;
    if (src + len > in.size()) throw Error(PF_ERR_RECOGNITION, "Source array was not long enough (ApplyLfr)");   // Array.Copy would throw
    std::memcpy(&out[(size_t)i * len], &in[src], len * sizeof(float));
  }
  return out;
}

void online_position_encode(std::vector<float>& x, int timesteps, int dim, int start_idx) {
  // OnlineWavFrontend.cs:152-188, float / double mix as written there
  const int half = dim / 2;
  const float inc = (float)std::log((double)10000.0f) / (float)(half - 1);
  std::vector<float> inv((size_t)half);
  for (int i = 0; i < half; ++i) {
    const float v = (float)(i + 1) * (-inc);
    inv[(size_t)i] = (float)std::exp((double)v);
  }
  for (int t = 0; t < timesteps; ++t) {
    const float p = (float)(start_idx + t + 1);
    float* row = &x[(size_t)t * dim];
    for (int i = 0; i < half; ++i) {
      const float arg = inv[(size_t)i] * p;                   // float product, then double sin / cos
      row[i] += (float)std::sin((double)arg);
      row[half + i] += (float)std::cos((double)arg);
    }
  }
}

void online_dynamic_mask(std::vector<float>& a, int chunk_size, int lfr) {
  // OnlineModel.cs:141-165
  const int n = (int)a.size();
  for (int i = 0; i < std::min(chunk_size, n); ++i) a[(size_t)i] = 0.f;
  const int decode_lfr = chunk_size + lfr;
  for (int i = decode_lfr; i < n; ++i) a[(size_t)i] = 0.f;
}

void online_cif(const std::vector<std::vector<float>>& hiddens, const std::vector<float>& alphas, float threshold,
                std::vector<std::vector<float>>& fired, float& carry_alpha, std::vector<float>& carry_hidden) {
  // OnlineRecognizer.cs:152-197, float arithmetic, products and sums rounded separately
  const size_t D = hiddens.empty() ? 0 : hiddens[0].size();
  float integrate = 0.0f;
  std::vector<float> frames(D, 0.f);
  fired.clear();
  const size_t n = std::min(hiddens.size(), alphas.size());
  for (size_t j = 0; j < n; ++j) {
    const float alpha = alphas[j];
    const std::vector<float>& h = hiddens[j];
    if (alpha + integrate < threshold) {
      integrate += alpha;
      for (size_t k = 0; k < D; ++k) { const volatile float prod = alpha * h[k]; frames[k] += prod; }
    } else {
      const float wgt = threshold - integrate;
      for (size_t k = 0; k < D; ++k) { const volatile float prod = wgt * h[k]; frames[k] += prod; }
      fired.push_back(frames);
      integrate += alpha;
      integrate -= threshold;
      for (size_t k = 0; k < D; ++k) frames[k] = integrate * h[k];
    }
  }
  carry_alpha = integrate;
  carry_hidden = frames;
  if (integrate > 0.0f)
    for (size_t k = 0; k < D; ++k) carry_hidden[k] = frames[k] / integrate;
}

static bool is_chinese_all(const std::string& s) {             // ^[一-龥]+$  (OnlineRecognizer.cs:446-458)
  const std::vector<uint32_t> cps = utf8_decode(s);
  if (cps.empty()) return false;
  for (uint32_t c : cps)
    if (c < 0x4e00 || c > 0x9fa5) return false;
  return true;
}
static void replace_all(std::string& s, const std::string& from, const std::string& to) {
  size_t pos = 0;
  while ((pos = s.find(from, pos)) != std::string::npos) { s.replace(pos, from.size(), to); pos += to.size(); }
}

std::string online_decode_text(const std::vector<std::string>& tokens, const std::vector<int64_t>& ids) {
  // OnlineRecognizer.cs:405-437
  static const std::string bar = "\xE2\x96\x81";               // U+2581
  std::string text;
  for (int64_t id : ids) {
    if (id == 2) break;
    if (id < 0 || id >= (int64_t)tokens.size()) throw Error(PF_ERR_RECOGNITION, "Index was outside the bounds of the array (tokens)");
    const std::string& tk = tokens[(size_t)id];
    if (tk != "</s>" && tk != "<s>" && tk != "<blank>" && tk != "<unk>") {
      if (is_chinese_all(tk)) text += tk;
      else text += bar + tk + bar;
    }
  }
  replace_all(text, "@@" + bar + bar, "");
  replace_all(text, "@@" + bar, "");
  replace_all(text, bar + bar, " ");
  replace_all(text, bar, "");
  for (char& c : text)                                         // ToLower(): ASCII letters (the token tables are CJK + lower-case BPE)
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
  return text;
}

// ------------------------------------------------------------------ OnlineStream -----------
OnlineStreamM::OnlineStreamM(std::shared_ptr<OnlineRecognizerM> r) : owner(std::move(r)) {
  std::shared_ptr<Engine> e = owner->engine();
  const int nd = e ? e->dec_layers() : 16, D = 512, lorder = 10;                    // OnlineStream.cs:44-46, 263-273
  States.assign((size_t)nd, std::vector<float>((size_t)D * lorder, 0.f));
  CifHidden.assign(1, std::vector<float>((size_t)D, 0.f));                          // InitHidden :241-250
  CifAlpha.assign(1, 0.f);                                                          // InitAlpha :251-260
  cache_feats_.assign((size_t)10 * 560, 0.f);                                       // InitCacheFeats :274-278
  cache_samples_.assign((size_t)160 * owner->chunk_length(), 0.f);                  // :63: a chunk of SILENCE is queued first
}

void OnlineStreamM::AddSamples(const float* samples, int64_t n) {
  if (disposed) throw Error(PF_ERR_DISPOSED, "OnlineStream");
  if (!samples) throw Error(PF_ERR_NULL_SAMPLES, "source");
  std::vector<float> s;
  {
    std::lock_guard<std::mutex> lk(mu);
    cache_samples_.insert(cache_samples_.end(), samples, samples + n);
    const size_t chunk = (size_t)160 * owner->chunk_length();
    if (cache_samples_.size() > chunk) {                        // ONE chunk per call (:94-102), the rest stays cached
      s.assign(cache_samples_.begin(), cache_samples_.begin() + (long)chunk);
      cache_samples_.erase(cache_samples_.begin(), cache_samples_.begin() + (long)chunk);
    }
  }
  if (!s.empty()) InputSpeech(s);
}

void OnlineStreamM::InputSpeech(const std::vector<float>& samples) {
  std::shared_ptr<Engine> e = owner->engine();
  if (!e) throw Error(PF_ERR_DISPOSED, "OnlineRecognizer");
  std::vector<float> fb;
  int t80 = 0;
  {
    std::lock_guard<std::mutex> lk(e->mutex());
    e->fbank_host(samples.data(), (int64_t)samples.size(), fb, t80);     // x 32768 + kaldi fbank (:129-130)
  }
  std::lock_guard<std::mutex> lk(mu);
  if (first_input_ && t80 > 0) {                                          // :131-143: the first frame is repeated once
    Speech.insert(Speech.end(), fb.begin(), fb.begin() + 80);
    first_input_ = false;
  }
  Speech.insert(Speech.end(), fb.begin(), fb.end());
}

bool OnlineStreamM::GetDecodeChunk(std::vector<float>& chunk) {
  std::lock_guard<std::mutex> lk(mu);
  const int F = 80, CL = owner->chunk_length();
  if ((size_t)CL * F > Speech.size()) return false;
  std::vector<float> pad;
  if (!cache_lfr_splice_.empty()) pad = cache_lfr_splice_;                // :172-178
  else pad.assign(Speech.begin(), Speech.begin() + F);                    // :180-187: first frame once more
  pad.insert(pad.end(), Speech.begin(), Speech.begin() + (long)CL * F);
  cache_lfr_splice_.assign(pad.end() - F, pad.end());                     // :189-190
  const ConfEntity& c = owner->conf();
  std::vector<float> x = pad;
  if (c.lfr_m != 1 || c.lfr_n != 1) x = online_apply_lfr(pad, F, c.lfr_m, c.lfr_n);
  const std::vector<float>& sh = owner->shift();
  const std::vector<float>& sc = owner->scale();
  if (!sh.empty()) {                                                      // ApplyCmvn, OnlineWavFrontend.cs:46-62
    const size_t dim = sh.size();
    for (size_t i = 0; i + dim <= x.size(); i += dim)
      for (size_t k = 0; k < dim; ++k) { const volatile float s = x[i + k] + sh[k]; x[i + k] = s * sc[k]; }
  }
  const double root = std::pow(512.0, 0.5);
  for (float& v : x) v = (float)((double)v * root);                       // :193
  const int W = 560, timesteps = (int)(x.size() / W);
  online_position_encode(x, timesteps, W, start_idx_);                    // :195-196
  chunk = cache_feats_;
  chunk.insert(chunk.end(), x.begin(), x.end());
  start_idx_ += timesteps;
  cache_feats_.assign(chunk.end() - (long)cache_feats_.size(), chunk.end());   // :201
  Speech.erase(Speech.begin(), Speech.begin() + (long)CL * F);            // RemoveChunk :210-224
  return true;
}

// ------------------------------------------------------------------ OnlineRecognizer -------
OnlineRecognizerM::OnlineRecognizerM(const std::string& model, const std::string&, const std::string& config,
                                     const std::string& mvn, const std::string& tokens, int, int device) {
  conf_ = load_conf(config);
  if (!tokens.empty() && file_exists(tokens)) tokens_ = read_lines(tokens);
  if (!mvn.empty()) parse_mvn_text(read_text_file(mvn), shift_, scale_);
  pf_engine_config ec;
  std::memset(&ec, 0, sizeof(ec));
  ec.struct_size = sizeof(ec);
  ec.device = device;
  ec.weights_path = model.c_str();
  ec.mvn_path = mvn.c_str();
  ec.fs = conf_.fs; ec.n_mels = conf_.n_mels; ec.lfr_m = conf_.lfr_m; ec.lfr_n = conf_.lfr_n;
  ec.snip_edges = conf_.snip_edges ? 1 : 0;
  ec.dither = conf_.dither;
  ec.frame_length_ms = 0; ec.frame_shift_ms = 0;     // never forwarded by the reference (OnlineWavFrontend.cs:20-27): kaldi defaults apply
  ec.window = conf_.window.c_str();
  engine_ = std::make_shared<Engine>(ec);
  // the streaming mirror carries the reference's literal geometry (OnlineStream.cs:44-46,263-278: 10 x 560 feature
  // cache, 16 x [512, 10] FSMN caches; OnlineWavFrontend.cs:152-188: sqrt(512)); a container of another geometry
  // would make Forward() index past those buffers, where the reference gets an ORT shape error
  const ModelCfg& m = engine_->model();
  PF_CHECK(m.kind_id() != 1, PF_ERR_UNSUPPORTED, "OnlineRecognizer: a SenseVoice container has no streaming graphs");
  PF_CHECK(m.d_model == 512 && m.feat_dim == 560 && m.kernel == 11, PF_ERR_UNSUPPORTED,
           "OnlineRecognizer: the streaming path is built for d_model 512, feature dim 560, FSMN kernel 11 (got " +
               std::to_string(m.d_model) + ", " + std::to_string(m.feat_dim) + ", " + std::to_string(m.kernel) + ")");
  PF_CHECK(conf_.lfr_m * conf_.n_mels == 560, PF_ERR_UNSUPPORTED, "OnlineRecognizer: lfr_m * n_mels must be 560");
}

std::shared_ptr<OnlineStreamM> OnlineRecognizerM::CreateOnlineStream() {
  if (disposed_) throw Error(PF_ERR_DISPOSED, "OnlineRecognizer");
  return std::make_shared<OnlineStreamM>(shared_from_this());
}

void OnlineRecognizerM::Dispose() {
  std::shared_ptr<Engine> e;
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (disposed_.exchange(true)) return;
    e.swap(engine_);
  }
  if (e) { std::lock_guard<std::mutex> lk(e->mutex()); }
  e.reset();
}

void OnlineRecognizerM::Forward(const std::vector<OnlineStreamM*>& streams) {
  if (streams.empty()) return;
  std::vector<OnlineStreamM*> work;
  std::vector<std::vector<float>> chunks;
  for (OnlineStreamM* s : streams) {                            // :349-363
    std::vector<float> c;
    if (!s->GetDecodeChunk(c)) continue;
    chunks.push_back(std::move(c));
    work.push_back(s);
  }
  if (work.empty()) return;
  try {
    std::shared_ptr<Engine> eh = engine();
    if (disposed_ || !eh) throw Error(PF_ERR_DISPOSED, "OnlineRecognizer");
    Engine* e = eh.get();
    std::lock_guard<std::mutex> lk(e->mutex());
    const int B = (int)work.size(), W = 560, D = 512, nd = e->dec_layers(), CW = 10;
    // PadSequence_unittest (:460-473): equal lengths assumed (stream i copied at i * its own length), then every
    // exact 0 becomes -23.025850929940457F (NOT multiplied by 32768 as the offline sentinel is)
    size_t maxlen = 0;
    for (auto& c : chunks) maxlen = std::max(maxlen, c.size());
    std::vector<float> speech(maxlen * (size_t)B, 0.f);
    for (int i = 0; i < B; ++i) {
      if ((size_t)i * chunks[(size_t)i].size() + chunks[(size_t)i].size() > speech.size())
        throw Error(PF_ERR_RECOGNITION, "Destination array was not long enough (PadSequence)");
      std::memcpy(&speech[(size_t)i * chunks[(size_t)i].size()], chunks[(size_t)i].data(), chunks[(size_t)i].size() * 4);
    }
    for (float& v : speech) v = v == 0.f ? -23.025850929940457f : v;
    const int Tc = (int)(maxlen / W);
    // stack_states (OnlineModel.cs:199-220): EVERY layer's in_cache is built from the streams' LAYER-0 state
    std::vector<float> cin((size_t)nd * B * D * CW);
    for (int l = 0; l < nd; ++l)
      for (int n = 0; n < B; ++n)
        std::memcpy(&cin[((size_t)l * B + n) * D * CW], work[(size_t)n]->States[0].data(), (size_t)D * CW * 4);
    // EncoderProj (:49-124)
    std::vector<float> enc((size_t)B * Tc * D), alphas((size_t)B * Tc);
    e->online_encoder(speech.data(), B, Tc, enc.data(), alphas.data());
    // PredictorProj (:126-231)
    std::vector<std::vector<std::vector<float>>> fired((size_t)B);
    const float thr = e->model().cif_threshold;
    size_t len_time = 0;
    for (int b = 0; b < B; ++b) {
      std::vector<float> a(alphas.begin() + (long)b * Tc, alphas.begin() + (long)(b + 1) * Tc);
      online_dynamic_mask(a);
      OnlineStreamM* s = work[(size_t)b];
      for (int t = 0; t < Tc; ++t)
        s->CifHidden.emplace_back(enc.begin() + ((long)b * Tc + t) * D, enc.begin() + ((long)b * Tc + t + 1) * D);
      s->CifAlpha.insert(s->CifAlpha.end(), a.begin(), a.end());
      if (b == 0) len_time = s->CifAlpha.size();                // :150: stream 0's count is used for every stream
    }
    int max_tok = 0;
    for (int b = 0; b < B; ++b) {
      OnlineStreamM* s = work[(size_t)b];
      if (s->CifAlpha.size() < len_time || s->CifHidden.size() < len_time)
        throw Error(PF_ERR_RECOGNITION, "Index was out of range (CIF cache)");
      std::vector<std::vector<float>> h(s->CifHidden.begin(), s->CifHidden.begin() + (long)len_time);
      std::vector<float> a(s->CifAlpha.begin(), s->CifAlpha.begin() + (long)len_time);
      float ca = 0.f;
      std::vector<float> ch;
      online_cif(h, a, thr, fired[(size_t)b], ca, ch);
      s->CifAlpha.assign(1, ca);                                // :214-220
      s->CifHidden.assign(1, ch);
      max_tok = std::max(max_tok, (int)fired[(size_t)b].size());
    }
    if (max_tok == 0) return;                                   // :380: Acoustic_embeds.Length == 0
    std::vector<float> emb((size_t)B * max_tok * D, 0.f);
    std::vector<int32_t> emb_len((size_t)B);
    for (int b = 0; b < B; ++b) {
      emb_len[(size_t)b] = (int32_t)fired[(size_t)b].size();
      for (size_t l = 0; l < fired[(size_t)b].size(); ++l)
        std::memcpy(&emb[((size_t)b * max_tok + l) * D], fired[(size_t)b][l].data(), (size_t)D * 4);
    }
    // DecoderProj (:233-334) + unstack_states (OnlineModel.cs:221-247)
    std::vector<int64_t> ids((size_t)B * max_tok);
    std::vector<float> cout((size_t)nd * B * D * CW);
    e->online_decoder(enc.data(), B, Tc, emb.data(), max_tok, emb_len.data(), cin.data(), nullptr, ids.data(), cout.data());
    for (int b = 0; b < B; ++b) {
      OnlineStreamM* s = work[(size_t)b];
      // :389: ALL max_tok positions are appended, also for streams that fired fewer tokens
      s->Tokens.insert(s->Tokens.end(), ids.begin() + (long)b * max_tok, ids.begin() + (long)(b + 1) * max_tok);
      for (int l = 0; l < nd; ++l)
        s->States[(size_t)l].assign(cout.begin() + ((long)l * B + b) * D * CW, cout.begin() + ((long)l * B + b + 1) * D * CW);
    }
  } catch (const Error& ex) {
    if (ex.code == PF_ERR_RECOGNITION && std::string(ex.what()).rfind("Online recognition failed", 0) == 0) throw;
    throw Error(PF_ERR_RECOGNITION, std::string("Online recognition failed: ") + ex.what());   // :397-400
  }
}

std::vector<std::string> OnlineRecognizerM::GetResults(const std::vector<OnlineStreamM*>& streams) {
  Forward(streams);
  std::vector<std::string> out;
  for (OnlineStreamM* s : streams) out.push_back(online_decode_text(tokens_, s->Tokens));
  return out;
}

}  // namespace pf
