// c_api.cpp — extern "C" boundary (include/paraformer_hip.h).  Every entry point converts C++
// exceptions into a negative pf_status + thread-local message; no exception crosses the ABI.
#include <algorithm>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>

#include "engine.h"
#include "group.h"
#include "online.h"
#include "recognizer.h"
#include "shards_sim.h"

namespace pf {
static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
}  // namespace pf

using namespace pf;

// Handle shells outlive the objects they point to: a second pf_engine_destroy / pf_recognizer_free / pf_stream_free on
// the same handle (Dispose followed by a finaliser, OfflineRecognizer.cs:448-476) finds an empty shell instead of freed
// memory.  Engine / recognizer / group shells (a handful per process) are never returned to the allocator; STREAM shells
// — one per utterance in a server — go through a quarantine (ShellPool below): a freed shell answers PF_ERR_DISPOSED for
// as long as it sits in the queue and is handed out again only after kQuarantine younger frees, so the memory is bounded
// (a few MB) instead of growing by one shell per stream for the life of the process.  The device / host state a shell
// pointed to IS released at once.
struct pf_engine {
  std::mutex mu;
  std::shared_ptr<Engine> e;
};
struct pf_recognizer {
  std::mutex mu;
  std::shared_ptr<Recognizer> r;
  pf_engine eng;
};
struct pf_stream {
  std::shared_ptr<Stream> s;
  bool freed = false;
};
struct pf_decoded { ResultEntity r; };
struct pf_online_recognizer {
  std::mutex mu;
  std::shared_ptr<OnlineRecognizerM> r;
  pf_engine eng;
};
struct pf_online_stream {
  std::shared_ptr<OnlineStreamM> s;
  bool freed = false;
};
struct pf_group {
  std::mutex mu;
  std::shared_ptr<Group> g;
  std::vector<std::unique_ptr<pf_engine>> views;    // borrowed engine handles (pf_group_engine)
};

// Recycles shells of type T (default-constructible, with `bool freed`) through a FIFO quarantine.
template <class T>
class ShellPool {
 public:
  static constexpr size_t kQuarantine = 1 << 16;
  T* get() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (q_.size() > kQuarantine) {
        T* t = q_.front();
        q_.pop_front();
        *t = T();                                     // a stale handle older than kQuarantine frees now aliases a live stream
        return t;
      }
    }
    return new T();
  }
  void retire(T* t) {                                 // the caller has emptied the shell and set freed
    std::lock_guard<std::mutex> lk(mu_);
    q_.push_back(t);
  }
 private:
  std::mutex mu_;
  std::deque<T*> q_;
};
static ShellPool<pf_stream>& stream_shells() { static ShellPool<pf_stream>* p = new ShellPool<pf_stream>(); return *p; }
static ShellPool<pf_online_stream>& online_stream_shells() { static ShellPool<pf_online_stream>* p = new ShellPool<pf_online_stream>(); return *p; }

#define PF_TRY try {
#define PF_CATCH                                                          \
  }                                                                       \
  catch (const pf::Error& ex) { pf::set_last_error(ex.what()); return ex.code; } \
  catch (const std::bad_alloc&) { pf::set_last_error("out of host memory"); return PF_ERR_DEVICE; } \
  catch (const std::exception& ex) { pf::set_last_error(ex.what()); return PF_ERR_INVALID_ARG; }

#define NEED(p) PF_CHECK((p) != nullptr, PF_ERR_INVALID_ARG, "null argument: " #p)

extern "C" {

int pf_version(void) { return PF_ABI_VERSION; }
const char* pf_last_error(void) { return pf::g_last_error.c_str(); }

int pf_engine_create(const pf_engine_config* cfg, pf_engine** out) {
  PF_TRY
  NEED(cfg); NEED(out);
  PF_CHECK(cfg->struct_size == (int32_t)sizeof(pf_engine_config), PF_ERR_INVALID_ARG, "pf_engine_config.struct_size mismatch");
  *out = nullptr;
  pf_engine* h = new pf_engine();
  try {
    h->e = std::make_shared<Engine>(*cfg);
  } catch (...) {
    delete h;
    throw;
  }
  *out = h;
  return PF_OK;
  PF_CATCH
}

void pf_engine_destroy(pf_engine* h) {
  if (!h) return;
  std::shared_ptr<Engine> e;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    e.swap(h->e);
  }
  if (!e) return;                                   // second destroy: nothing left to do
  try {
    { std::lock_guard<std::mutex> lk(e->mutex()); } // a call in flight on another thread finishes first
    e.reset();
  } catch (...) {}
}

static std::shared_ptr<Engine> E(pf_engine* h) {
  PF_CHECK(h != nullptr, PF_ERR_INVALID_ARG, "null engine");
  std::lock_guard<std::mutex> lk(h->mu);
  PF_CHECK(h->e != nullptr, PF_ERR_DISPOSED, "OfflineRecognizer");
  return h->e;                                      // the caller's copy keeps the engine alive for the call
}

int pf_engine_info(pf_engine* h, int32_t* kind, int32_t* vocab, int32_t* feat_dim, int32_t* has_ts) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  if (kind) *kind = e->model().kind_id();
  if (vocab) *vocab = e->model().vocab;
  if (feat_dim) *feat_dim = e->model().feat_dim;
  if (has_ts) *has_ts = e->model().timestamp_head ? 1 : 0;
  return PF_OK;
  PF_CATCH
}

int pf_frontend_num_frames(pf_engine* h, int64_t n, int32_t* t) {
  PF_TRY
  NEED(t);
  *t = E(h)->num_lfr_frames(n);
  return PF_OK;
  PF_CATCH
}

int pf_frontend(pf_engine* h, const float* samples, int64_t n, float* feats, int64_t cap, int32_t* t_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  if (!samples) throw Error(PF_ERR_NULL_SAMPLES, "source");
  std::lock_guard<std::mutex> lk(e->mutex());
  std::vector<float> f;
  int t = 0;
  e->frontend_host(samples, n, f, t);
  if (t_out) *t_out = t;
  PF_CHECK((int64_t)f.size() <= cap, PF_ERR_CAPACITY, "feats capacity < " + std::to_string(f.size()));
  if (!f.empty()) { NEED(feats); std::memcpy(feats, f.data(), f.size() * 4); }
  return PF_OK;
  PF_CATCH
}

int pf_fbank(pf_engine* h, const float* samples, int64_t n, float* out, int64_t cap, int32_t* t80_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  if (!samples) throw Error(PF_ERR_NULL_SAMPLES, "source");
  std::lock_guard<std::mutex> lk(e->mutex());
  std::vector<float> f;
  int t = 0;
  e->fbank_host(samples, n, f, t);
  if (t80_out) *t80_out = t;
  PF_CHECK((int64_t)f.size() <= cap, PF_ERR_CAPACITY, "fbank capacity < " + std::to_string(f.size()));
  if (!f.empty()) { NEED(out); std::memcpy(out, f.data(), f.size() * 4); }
  return PF_OK;
  PF_CATCH
}

static bool want_logits(const pf_batch_out* o) { return o && o->logits && o->logits_cap > 0; }

int pf_forward_feats(pf_engine* h, const float* speech, int32_t B, int32_t Tmax, const int32_t* hotwords,
                     int32_t n_hotwords, pf_batch_out* out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(out);
  std::lock_guard<std::mutex> lk(e->mutex());
  if (e->model().seaco) e->set_hotwords(hotwords, hotwords ? n_hotwords : 0);
  e->forward_feats_host(speech, B, Tmax, want_logits(out));
  e->publish_thread_result();
  e->fetch(out);
  return PF_OK;
  PF_CATCH
}

int pf_model_proj(pf_engine* h, const float* const* speech, const int32_t* lens, int32_t B,
                  const int32_t* hotwords, int32_t n_hotwords, pf_batch_out* out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(out); NEED(speech); NEED(lens);
  std::lock_guard<std::mutex> lk(e->mutex());
  if (e->model().seaco) e->set_hotwords(hotwords, hotwords ? n_hotwords : 0);
  e->model_proj_host(speech, lens, B, want_logits(out));
  e->publish_thread_result();
  e->fetch(out);
  return PF_OK;
  PF_CATCH
}

int pf_recognize(pf_engine* h, const float* const* samples, const int64_t* n, int32_t B, const int32_t* hotwords,
                 int32_t n_hotwords, pf_batch_out* out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(out); NEED(samples); NEED(n);
  std::lock_guard<std::mutex> lk(e->mutex());
  if (e->model().seaco) e->set_hotwords(hotwords, hotwords ? n_hotwords : 0);
  e->stage_audio(samples, n, B);
  e->run_staged(want_logits(out));
  e->publish_thread_result();
  e->fetch(out);
  return PF_OK;
  PF_CATCH
}

int pf_engine_set_hotwords(pf_engine* h, const int32_t* hotwords, int32_t n_hotwords) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  PF_CHECK(n_hotwords >= 0 && (n_hotwords == 0 || hotwords), PF_ERR_INVALID_ARG, "set_hotwords: bad arguments");
  std::lock_guard<std::mutex> lk(e->mutex());
  if (e->model().seaco) e->set_hotwords(hotwords, n_hotwords);
  return PF_OK;
  PF_CATCH
}

int pf_stage_audio(pf_engine* h, const float* const* samples, const int64_t* n, int32_t B) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(samples); NEED(n);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->stage_audio(samples, n, B);
  return PF_OK;
  PF_CATCH
}

int pf_run_staged(pf_engine* h) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  std::lock_guard<std::mutex> lk(e->mutex());
  e->drop_thread_result();               // pf_fetch after this call reads the engine's staged result
  e->run_staged(false);
  return PF_OK;
  PF_CATCH
}

int pf_sync(pf_engine* h) {
  PF_TRY
  E(h)->sync();
  return PF_OK;
  PF_CATCH
}

int pf_fetch(pf_engine* h, pf_batch_out* out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  std::lock_guard<std::mutex> lk(e->mutex());
  e->fetch(out);
  return PF_OK;
  PF_CATCH
}

int pf_fetch_ids_device(pf_engine* h, int64_t* ids_dev, int32_t l_cap, int32_t* L_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  std::lock_guard<std::mutex> lk(e->mutex());
  e->fetch_ids_device(ids_dev, l_cap, L_out);
  return PF_OK;
  PF_CATCH
}

int pf_profile_enable(pf_engine* h, int32_t on) {
  PF_TRY
  E(h)->profile_enable(on != 0);
  return PF_OK;
  PF_CATCH
}
int pf_profile_reset(pf_engine* h) {
  PF_TRY
  E(h)->profile_reset();
  return PF_OK;
  PF_CATCH
}
int pf_profile_select(pf_engine* h, const char* cls) {
  PF_TRY
  E(h)->profile_select(cls ? cls : "");
  return PF_OK;
  PF_CATCH
}
int pf_profile_get(pf_engine* h, const char* cls, double* ms, int64_t* launches, double* fpl) {
  PF_TRY
  NEED(cls);
  if (!E(h)->profile_get(cls, ms, launches, fpl)) {
    if (ms) *ms = 0;
    if (launches) *launches = 0;
    if (fpl) *fpl = 0;
  }
  return PF_OK;
  PF_CATCH
}
int pf_profile_kernel(pf_engine* h, const char* cls, char* name_out, int32_t cap) {
  PF_TRY
  NEED(cls); NEED(name_out);
  PF_CHECK(cap > 0, PF_ERR_INVALID_ARG, "pf_profile_kernel: cap <= 0");
  const std::string k = E(h)->profile_kernel(cls);
  PF_CHECK((int)k.size() < cap, PF_ERR_CAPACITY, "pf_profile_kernel: name needs " + std::to_string(k.size() + 1) + " bytes");
  std::memcpy(name_out, k.c_str(), k.size() + 1);
  return PF_OK;
  PF_CATCH
}
int pf_last_flops(pf_engine* h, double* f) {
  PF_TRY
  NEED(f);
  *f = E(h)->last_flops();
  return PF_OK;
  PF_CATCH
}

// ---- multi-device group ---------------------------------------------------
int pf_group_create(const pf_engine_config* cfg, const int32_t* devices, int32_t n_devices, pf_group** out) {
  PF_TRY
  NEED(cfg); NEED(out); NEED(devices);
  PF_CHECK(cfg->struct_size == (int32_t)sizeof(pf_engine_config), PF_ERR_INVALID_ARG, "pf_engine_config.struct_size mismatch");
  *out = nullptr;
  std::shared_ptr<Group> g = std::make_shared<Group>(*cfg, devices, n_devices);
  pf_group* h = new pf_group();
  h->g = g;
  for (int i = 0; i < g->size(); ++i) {
    h->views.emplace_back(new pf_engine());
    h->views.back()->e = g->engine(i);
  }
  *out = h;
  return PF_OK;
  PF_CATCH
}

void pf_group_destroy(pf_group* h) {
  if (!h) return;
  std::shared_ptr<Group> g;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    g.swap(h->g);
    for (auto& v : h->views) { std::lock_guard<std::mutex> lk2(v->mu); v->e.reset(); }
  }
  if (!g) return;
  try {
    { std::lock_guard<std::mutex> lk(g->mutex()); }
    g.reset();
  } catch (...) {}
}

static std::shared_ptr<Group> GR(pf_group* h) {
  PF_CHECK(h != nullptr, PF_ERR_INVALID_ARG, "null group");
  std::lock_guard<std::mutex> lk(h->mu);
  PF_CHECK(h->g != nullptr, PF_ERR_DISPOSED, "OfflineRecognizer");
  return h->g;
}

int pf_group_info(pf_group* h, int32_t* n_engines, int32_t* uses_rccl) {
  PF_TRY
  std::shared_ptr<Group> g = GR(h);
  if (n_engines) *n_engines = g->size();
  if (uses_rccl) *uses_rccl = g->uses_rccl() ? 1 : 0;
  return PF_OK;
  PF_CATCH
}

pf_engine* pf_group_engine(pf_group* h, int32_t i) {
  if (!h) return nullptr;
  std::lock_guard<std::mutex> lk(h->mu);
  if (!h->g || i < 0 || i >= (int32_t)h->views.size()) return nullptr;
  return h->views[(size_t)i].get();
}

int pf_group_recognize(pf_group* h, const float* const* samples, const int64_t* n, int32_t B, const int32_t* hotwords,
                       int32_t n_hotwords, pf_batch_out* out) {
  PF_TRY
  std::shared_ptr<Group> g = GR(h);
  NEED(out);
  std::lock_guard<std::mutex> lk(g->mutex());
  g->recognize(samples, n, B, hotwords, n_hotwords, want_logits(out));
  g->fetch(out);
  return PF_OK;
  PF_CATCH
}

// ---- pf_host_group_sim: the shard runner of pf_group with arithmetic stand-ins for the devices (shards_sim.h) ----
int pf_host_group_sim(int32_t G, int32_t B, const int32_t* fire_count, int32_t has_cif, int32_t fixed_L, int32_t collective,
                      int32_t fail_shard, int32_t fail_stage, int64_t* ids_out, int32_t l_cap, int32_t* token_num_out,
                      int32_t* L_out) {
  PF_TRY
  PF_CHECK(G > 0 && G <= 64 && B >= 0 && (B == 0 || fire_count), PF_ERR_INVALID_ARG, "pf_host_group_sim: bad arguments");
  pf::SimBackend be(G);
  be.B = B; be.has_cif = has_cif; be.fixed_L = fixed_L; be.collective = collective; be.fail_shard = fail_shard;
  be.fail_stage = fail_stage; be.fire = fire_count;
  pf::ShardRunner runner(G);
  pf::HostBatchOut m;
  runner.recognize(be, B, /*Tg=*/1, has_cif != 0, /*V=*/1, false, m);
  if (L_out) *L_out = m.L;
  if (ids_out) {
    PF_CHECK(l_cap >= m.L, PF_ERR_CAPACITY, "pf_host_group_sim: l_cap < L");
    for (int b = 0; b < B; ++b) std::memcpy(ids_out + (size_t)b * l_cap, m.ids.data() + (size_t)b * m.L, (size_t)m.L * 8);
  }
  if (token_num_out && B > 0) std::memcpy(token_num_out, m.token_num.data(), (size_t)B * 4);
  return PF_OK;
  PF_CATCH
}

int pf_group_fetch(pf_group* h, pf_batch_out* out) {
  PF_TRY
  std::shared_ptr<Group> g = GR(h);
  std::lock_guard<std::mutex> lk(g->mutex());
  g->fetch(out);
  return PF_OK;
  PF_CATCH
}

// ---- stand-alone ops -------------------------------------------------------
int pf_op_lfr_cmvn_pad(pf_engine* h, const float* const* fbank, const int32_t* t80, int32_t B, int32_t sentinel,
                       float* out, int64_t cap, int32_t* tmax) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(fbank); NEED(t80);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_lfr_cmvn_pad(fbank, t80, B, sentinel, out, cap, tmax);
  return PF_OK;
  PF_CATCH
}
int pf_op_argmax(pf_engine* h, const float* x, int64_t rows, int32_t V, int64_t* ids) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(ids);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_argmax(x, rows, V, ids);
  return PF_OK;
  PF_CATCH
}
int pf_op_gemm(pf_engine* h, const float* A, const float* W, const float* bias, int32_t M, int32_t N, int32_t K,
               int32_t epi, float* C) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(A); NEED(W); NEED(C);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_gemm(A, W, bias, M, N, K, epi, C);
  return PF_OK;
  PF_CATCH
}
int pf_op_qlinear(pf_engine* h, const float* x, const float* W, const float* bias, int32_t M, int32_t N, int32_t K, int32_t relu,
                  int32_t x_is_f16, float* y, uint8_t* xq_out, float* aparams_out, uint8_t* wq_out, float* wscale_out,
                  int32_t* wzp_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(W); NEED(y);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_qlinear(x, W, bias, M, N, K, relu, x_is_f16, y, xq_out, aparams_out, wq_out, wscale_out, wzp_out);
  return PF_OK;
  PF_CATCH
}
int pf_op_gemm_ex(pf_engine* h, const pf_gemm_desc* d, const float* A, const float* W, float* C) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(d); NEED(A); NEED(W); NEED(C);
  PF_CHECK(d->struct_size == (int32_t)sizeof(pf_gemm_desc), PF_ERR_INVALID_ARG, "pf_gemm_desc.struct_size mismatch");
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_gemm_ex(*d, A, W, C);
  return PF_OK;
  PF_CATCH
}
int pf_op_gemm_rc(pf_engine* h, const pf_gemm_rc_desc* d, const float* A, const float* W, float* x_out, float* n16_out,
                  float* n32_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(d); NEED(A); NEED(W);
  PF_CHECK(d->struct_size == (int32_t)sizeof(pf_gemm_rc_desc), PF_ERR_INVALID_ARG, "pf_gemm_rc_desc.struct_size mismatch");
  PF_CHECK(x_out || n16_out || n32_out, PF_ERR_INVALID_ARG, "gemm_rc: no output requested");
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_gemm_rc(*d, A, W, x_out, n16_out, n32_out);
  return PF_OK;
  PF_CATCH
}
int pf_op_ffn(pf_engine* h, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
              const float* resid, int32_t M, int32_t D, int32_t F, float* y) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(w1); NEED(b1); NEED(w2); NEED(b2); NEED(resid); NEED(y);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_ffn(x, w1, b1, w2, b2, resid, M, D, F, y);
  return PF_OK;
  PF_CATCH
}
int pf_op_ffn_fused(pf_engine* h, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                    const float* resid, const float* g, const float* be, int32_t M, float* x_out, float* n16_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(w1); NEED(b1); NEED(w2); NEED(b2);
  PF_CHECK(x_out || n16_out, PF_ERR_INVALID_ARG, "ffn_fused: no output requested");
  PF_CHECK((g != nullptr) == (be != nullptr) && (g || !n16_out), PF_ERR_INVALID_ARG, "ffn_fused: the LayerNorm output needs gamma and beta");
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_ffn_fused(x, w1, b1, w2, b2, resid, g, be, M, x_out, n16_out);
  return PF_OK;
  PF_CATCH
}
int pf_op_dec_ffn_fused(pf_engine* h, const pf_dec_ffn_desc* d, float* t_out, float* n_out, float* x_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(d);
  PF_CHECK(d->struct_size == (int32_t)sizeof(pf_dec_ffn_desc), PF_ERR_INVALID_ARG, "pf_dec_ffn_desc.struct_size mismatch");
  PF_CHECK(d->reserved == 0, PF_ERR_INVALID_ARG, "pf_dec_ffn_desc.reserved must be 0");
  NEED(d->w1); NEED(d->b1); NEED(d->gamma_f); NEED(d->beta_f); NEED(d->w2);
  if (d->ctx) { NEED(d->wo); NEED(d->bo); NEED(d->resid); NEED(d->ln1_gamma); NEED(d->ln1_beta); }
  else { NEED(d->x); PF_CHECK(!x_out, PF_ERR_INVALID_ARG, "dec_ffn_fused: x_out needs the out-projection form (ctx)"); }
  PF_CHECK(t_out || n_out, PF_ERR_INVALID_ARG, "dec_ffn_fused: no output requested");
  PF_CHECK((d->ln_gamma != nullptr) == (d->ln_beta != nullptr) && (d->ln_gamma || !n_out), PF_ERR_INVALID_ARG,
           "dec_ffn_fused: the LayerNorm output needs gamma and beta");
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_dec_ffn_fused(*d, t_out, n_out, x_out);
  return PF_OK;
  PF_CATCH
}
int pf_op_attn_ffn_fused(pf_engine* h, const pf_attn_ffn_desc* d, float* x_out, float* n16_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(d);
  PF_CHECK(d->struct_size == (int32_t)sizeof(pf_attn_ffn_desc), PF_ERR_INVALID_ARG, "pf_attn_ffn_desc.struct_size mismatch");
  NEED(d->ctx); NEED(d->wo); NEED(d->bo); NEED(d->v); NEED(d->fsmn_w); NEED(d->ln2_gamma); NEED(d->ln2_beta);
  NEED(d->w1); NEED(d->b1); NEED(d->w2); NEED(d->b2);
  PF_CHECK(x_out || n16_out, PF_ERR_INVALID_ARG, "attn_ffn_fused: no output requested");
  PF_CHECK((d->ln_gamma != nullptr) == (d->ln_beta != nullptr) && (d->ln_gamma || !n16_out), PF_ERR_INVALID_ARG,
           "attn_ffn_fused: the LayerNorm output needs gamma and beta");
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_ffn_fused(nullptr, d->w1, d->b1, d->w2, d->b2, d->resid, d->ln_gamma, d->ln_beta, d->M, x_out, n16_out, d);
  return PF_OK;
  PF_CATCH
}
int pf_op_fsmn_enc(pf_engine* h, const float* v, const float* w, int32_t B, int32_t T, int32_t D, int32_t k, float* y) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(v); NEED(w); NEED(y);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_fsmn_enc(v, w, B, T, D, k, y);
  return PF_OK;
  PF_CATCH
}
int pf_op_linear32(pf_engine* h, const float* x, const float* W, const float* bias, const float* resid, int32_t M, int32_t N,
                   int32_t K, int32_t relu, float* y) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(W); NEED(y);
  PF_CHECK(M > 0 && N > 0 && K > 0, PF_ERR_INVALID_ARG, "linear32: bad shape");
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_linear32(x, W, bias, resid, M, N, K, relu != 0, y);
  return PF_OK;
  PF_CATCH
}
int pf_op_ffn32(pf_engine* h, const float* x, const float* W1, const float* b1, const float* W2, const float* b2, int32_t M,
                int32_t D, int32_t F, float* y) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(W1); NEED(b1); NEED(W2); NEED(b2); NEED(y);
  PF_CHECK(M > 0 && D > 0 && F > 0, PF_ERR_INVALID_ARG, "ffn32: bad shape");
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_ffn32(x, W1, b1, W2, b2, M, D, F, y);
  return PF_OK;
  PF_CATCH
}
int pf_op_fsmn_dec(pf_engine* h, const float* tn, const float* w, const int32_t* token_num, int32_t B, int32_t L,
                   int32_t D, int32_t k, float* x) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(tn); NEED(w); NEED(token_num); NEED(x);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_fsmn_dec(tn, w, token_num, B, L, D, k, x);
  return PF_OK;
  PF_CATCH
}
int pf_op_logsoftmax_argmax(pf_engine* h, const float* x, int64_t rows, int32_t V, float* y, int64_t* ids) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(ids);
  PF_CHECK(rows >= 0 && V > 0, PF_ERR_INVALID_ARG, "logsoftmax_argmax: bad shape");
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_logsoftmax_argmax(x, rows, V, y, ids);
  return PF_OK;
  PF_CATCH
}
int pf_op_layernorm(pf_engine* h, const float* x, const float* g, const float* b, int64_t rows, int32_t D, float* y) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(g); NEED(b); NEED(y);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_layernorm(x, g, b, rows, D, y);
  return PF_OK;
  PF_CATCH
}
int pf_op_attention(pf_engine* h, const float* q, const float* k, const float* v, int32_t B, int32_t Lq, int32_t Lk,
                    int32_t heads, float* out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(q); NEED(k); NEED(v); NEED(out);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_attention(q, k, v, B, Lq, Lk, heads, out);
  return PF_OK;
  PF_CATCH
}
int pf_op_qkv_attention(pf_engine* h, const float* x, const float* w, const float* bias, int32_t B, int32_t T, int32_t K,
                        float* q_out, float* k_out, float* v_out, float* ctx_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(x); NEED(w);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_qkv_attention(x, w, bias, B, T, K, q_out, k_out, v_out, ctx_out);
  return PF_OK;
  PF_CATCH
}
int pf_op_fsmn(pf_engine* h, const float* v, const float* w, const float* mask, int32_t B, int32_t T, int32_t D,
               int32_t k, float* y) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(v); NEED(w); NEED(y);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_fsmn(v, w, mask, B, T, D, k, y);
  return PF_OK;
  PF_CATCH
}
int pf_op_cif(pf_engine* h, const float* H, const float* alphas, int32_t B, int32_t T, int32_t D, float thr,
              int32_t Lcap, float* E_, int32_t* fire_count, int32_t* token_num, int32_t* L_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  NEED(H); NEED(alphas); NEED(fire_count); NEED(token_num);
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_cif(H, alphas, B, T, D, thr, Lcap, E_, fire_count, token_num, L_out);
  return PF_OK;
  PF_CATCH
}
int pf_op_encoder(pf_engine* h, const float* speech, int32_t B, int32_t T, float* Hout) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  Engine* e = eh_.get();
  std::lock_guard<std::mutex> lk(e->mutex());
  e->op_encoder(speech, B, T, Hout);
  return PF_OK;
  PF_CATCH
}

// ---- recognizer mirror -------------------------------------------------------
int pf_recognizer_create(const char* model, const char* config, const char* mvn, const char* tokens,
                         const char* modeleb, const char* hotword, int32_t batch_size, int32_t threads_num,
                         int32_t device, pf_recognizer** out) {
  PF_TRY
  NEED(out);
  *out = nullptr;
  auto s = [](const char* p) { return std::string(p ? p : ""); };
  std::shared_ptr<Recognizer> r = std::make_shared<Recognizer>(s(model), s(config), s(mvn), s(tokens), s(modeleb),
                                                               s(hotword), batch_size, threads_num, device);
  pf_recognizer* h = new pf_recognizer();
  h->r = r;
  h->eng.e = r->engine();
  *out = h;
  return PF_OK;
  PF_CATCH
}

void pf_recognizer_dispose(pf_recognizer* h) {
  if (!h) return;
  std::shared_ptr<Recognizer> r;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    r = h->r;
  }
  if (!r) return;
  { std::lock_guard<std::mutex> lk(h->eng.mu); h->eng.e.reset(); }
  try { r->Dispose(); } catch (...) {}
}

void pf_recognizer_free(pf_recognizer* h) {
  if (!h) return;
  pf_recognizer_dispose(h);                         // releases the engine even while streams are still alive
  std::lock_guard<std::mutex> lk(h->mu);
  h->r.reset();                                     // the object itself goes with its last stream
}

pf_engine* pf_recognizer_engine(pf_recognizer* h) {
  if (!h) return nullptr;
  std::lock_guard<std::mutex> lk(h->eng.mu);
  return h->eng.e ? &h->eng : nullptr;
}

static std::shared_ptr<Recognizer> R(pf_recognizer* h);
int pf_recognizer_num_engines(pf_recognizer* h) {
  PF_TRY
  std::shared_ptr<Recognizer> r = R(h);
  PF_CHECK(!r->disposed(), PF_ERR_DISPOSED, "OfflineRecognizer");
  return r->engines_created();
  PF_CATCH
}

static std::shared_ptr<Recognizer> R(pf_recognizer* h) {
  PF_CHECK(h != nullptr, PF_ERR_INVALID_ARG, "null recognizer");
  std::lock_guard<std::mutex> lk(h->mu);
  PF_CHECK(h->r != nullptr, PF_ERR_DISPOSED, "OfflineRecognizer");
  return h->r;
}
static Stream* S(pf_stream* h) {
  PF_CHECK(h != nullptr, PF_ERR_INVALID_ARG, "null stream");
  PF_CHECK(!h->freed && h->s != nullptr, PF_ERR_DISPOSED, "OfflineStream");
  PF_CHECK(!h->s->disposed, PF_ERR_DISPOSED, "OfflineStream");         // ObjectDisposedException
  return h->s.get();
}

int pf_recognizer_create_stream(pf_recognizer* h, pf_stream** out) {
  PF_TRY
  NEED(out);
  *out = nullptr;
  std::shared_ptr<Stream> s = R(h)->CreateOfflineStream();
  pf_stream* sh = stream_shells().get();
  sh->s = s;
  *out = sh;
  return PF_OK;
  PF_CATCH
}

int pf_stream_create(const char* mvn_path, int32_t fs, int32_t n_mels, int32_t lfr_m, int32_t lfr_n, int32_t snip_edges,
                     float dither, const char* window, pf_stream** out) {
  PF_TRY
  NEED(out);
  *out = nullptr;
  ConfEntity c;                                     // FrontendConfEntity defaults where the caller passes 0 / NULL
  if (fs > 0) c.fs = fs;
  if (n_mels > 0) c.n_mels = n_mels;
  if (lfr_m > 0) c.lfr_m = lfr_m;
  if (lfr_n > 0) c.lfr_n = lfr_n;
  c.snip_edges = snip_edges != 0;
  c.dither = dither;
  if (window && window[0]) c.window = window;
  std::shared_ptr<Stream> s = std::make_shared<Stream>(std::string(mvn_path ? mvn_path : ""), c);
  pf_stream* sh = stream_shells().get();
  sh->s = s;
  *out = sh;
  return PF_OK;
  PF_CATCH
}

int pf_stream_add_samples(pf_stream* h, const float* samples, int64_t n) {
  PF_TRY
  Stream* s = S(h);
  PF_CHECK(!s->owner || !s->owner->disposed(), PF_ERR_DISPOSED, "OfflineRecognizer");
  s->AddSamples(samples, n);
  return PF_OK;
  PF_CATCH
}

int pf_stream_set_hotwords(pf_stream* h, const int32_t* ids, const int32_t* lens, int32_t n) {
  PF_TRY
  Stream* s = S(h);
  s->Hotwords.clear();
  s->hotwords_null = n < 0;
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    NEED(lens);
    PF_CHECK(lens[i] >= 0, PF_ERR_INVALID_ARG, "negative hotword length");
    if (lens[i] > 0) NEED(ids);
    s->Hotwords.emplace_back(ids + off, ids + off + lens[i]);
    off += (size_t)lens[i];
  }
  return PF_OK;
  PF_CATCH
}

int pf_stream_get_hotwords(pf_stream* h, int32_t* ids, int32_t ids_cap, int32_t* lens, int32_t lens_cap, int32_t* n) {
  PF_TRY
  Stream* s = S(h);
  NEED(n);
  if (s->hotwords_null) { *n = -1; return PF_OK; }
  *n = (int32_t)s->Hotwords.size();
  size_t tot = 0;
  for (auto& w : s->Hotwords) tot += w.size();
  PF_CHECK((int32_t)s->Hotwords.size() <= lens_cap && (int64_t)tot <= ids_cap, PF_ERR_CAPACITY, "hotword buffers too small");
  size_t off = 0;
  for (size_t i = 0; i < s->Hotwords.size(); ++i) {
    lens[i] = (int32_t)s->Hotwords[i].size();
    for (int32_t v : s->Hotwords[i]) ids[off++] = v;
  }
  return PF_OK;
  PF_CATCH
}

int pf_stream_num_feature_floats(pf_stream* h, int32_t* n) {
  PF_TRY
  NEED(n);
  *n = S(h)->SpeechLength;
  return PF_OK;
  PF_CATCH
}

void pf_stream_dispose(pf_stream* h) {
  // DisposeOfflineStream (OfflineRecognizer.cs:441): the buffers go, the handle stays valid and every later
  // call on it answers PF_ERR_DISPOSED (ObjectDisposedException) until pf_stream_free
  if (!h || h->freed || !h->s) return;
  h->s->Dispose();
}

void pf_stream_free(pf_stream* h) {
  if (!h || h->freed) return;
  if (h->s) h->s->Dispose();
  h->s.reset();                                     // drops this stream's share of the recognizer object
  h->freed = true;                                  // the shell stays (double free / use after free -> PF_ERR_DISPOSED) ...
  stream_shells().retire(h);                        // ... in the quarantine
}

int pf_recognizer_get_results(pf_recognizer* h, pf_stream* const* streams, int32_t n) {
  PF_TRY
  std::shared_ptr<Recognizer> r = R(h);
  std::vector<Stream*> ss;
  for (int i = 0; i < n; ++i) { NEED(streams); ss.push_back(S(streams[i])); }
  r->GetResults(ss);
  return PF_OK;
  PF_CATCH
}

static const ResultEntity& RES(pf_recognizer* h, int32_t i) {
  const std::vector<ResultEntity>& res = R(h)->results_of_this_thread();   // thread-local: outlives this call
  PF_CHECK(i >= 0 && i < (int32_t)res.size(), PF_ERR_INVALID_ARG, "result index out of range");
  return res[(size_t)i];
}

int pf_result_text(pf_recognizer* h, int32_t i, const char** utf8, int32_t* len16) {
  PF_TRY
  const ResultEntity& e = RES(h, i);
  if (utf8) *utf8 = e.Text.c_str();
  if (len16) *len16 = e.TextLen;
  return PF_OK;
  PF_CATCH
}
int pf_result_num_tokens(pf_recognizer* h, int32_t i, int32_t* n) {
  PF_TRY
  NEED(n);
  *n = (int32_t)RES(h, i).Tokens.size();
  return PF_OK;
  PF_CATCH
}
int pf_result_token(pf_recognizer* h, int32_t i, int32_t j, const char** utf8) {
  PF_TRY
  const ResultEntity& e = RES(h, i);
  PF_CHECK(j >= 0 && j < (int32_t)e.Tokens.size(), PF_ERR_INVALID_ARG, "token index out of range");
  NEED(utf8);
  *utf8 = e.Tokens[(size_t)j].c_str();
  return PF_OK;
  PF_CATCH
}
int pf_result_num_timestamps(pf_recognizer* h, int32_t i, int32_t* n) {
  PF_TRY
  NEED(n);
  *n = (int32_t)RES(h, i).Timestamps.size();
  return PF_OK;
  PF_CATCH
}
int pf_result_timestamp(pf_recognizer* h, int32_t i, int32_t j, const int32_t** ints, int32_t* n_ints) {
  PF_TRY
  const ResultEntity& e = RES(h, i);
  PF_CHECK(j >= 0 && j < (int32_t)e.Timestamps.size(), PF_ERR_INVALID_ARG, "timestamp index out of range");
  if (ints) *ints = e.Timestamps[(size_t)j].data();
  if (n_ints) *n_ints = (int32_t)e.Timestamps[(size_t)j].size();
  return PF_OK;
  PF_CATCH
}
int pf_stream_tokens(pf_stream* h, const int64_t** ids, int32_t* n) {
  PF_TRY
  Stream* s = S(h);
  if (ids) *ids = s->Tokens.data();
  if (n) *n = (int32_t)s->Tokens.size();
  return PF_OK;
  PF_CATCH
}

int pf_stream_set_tokens(pf_stream* h, const int64_t* ids, int32_t n) {
  PF_TRY
  Stream* s = S(h);
  PF_CHECK(n >= 0, PF_ERR_INVALID_ARG, "negative token count");
  if (n > 0) NEED(ids);
  s->Tokens.assign(ids, ids + n);
  return PF_OK;
  PF_CATCH
}
int pf_stream_num_timestamps(pf_stream* h, int32_t* n) {
  PF_TRY
  NEED(n);
  *n = (int32_t)S(h)->Timestamps.size();
  return PF_OK;
  PF_CATCH
}
int pf_stream_timestamp(pf_stream* h, int32_t j, const int32_t** ints, int32_t* n_ints) {
  PF_TRY
  Stream* s = S(h);
  PF_CHECK(j >= 0 && j < (int32_t)s->Timestamps.size(), PF_ERR_INVALID_ARG, "timestamp index out of range");
  if (ints) *ints = s->Timestamps[(size_t)j].data();
  if (n_ints) *n_ints = (int32_t)s->Timestamps[(size_t)j].size();
  return PF_OK;
  PF_CATCH
}
int pf_stream_set_timestamps(pf_stream* h, const int32_t* ints, const int32_t* lens, int32_t n) {
  PF_TRY
  Stream* s = S(h);
  PF_CHECK(n >= 0, PF_ERR_INVALID_ARG, "negative timestamp count");
  TsList ts;
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    NEED(lens);
    PF_CHECK(lens[i] >= 0, PF_ERR_INVALID_ARG, "negative timestamp length");
    if (lens[i] > 0) NEED(ints);
    ts.emplace_back(ints + off, ints + off + lens[i]);
    off += (size_t)lens[i];
  }
  s->Timestamps.swap(ts);
  return PF_OK;
  PF_CATCH
}
int pf_stream_get_speech(pf_stream* h, float* out, int64_t cap, int32_t* n_floats) {
  PF_TRY
  Stream* s = S(h);
  NEED(n_floats);
  if (!s->has_speech) { *n_floats = -1; return PF_OK; }            // OfflineInputEntity.Speech == null
  PF_CHECK(s->owner != nullptr || s->pending.empty(), PF_ERR_UNSUPPORTED,
           "OfflineInputEntity.Speech of a stream no recognizer has adopted yet: its features are computed at the first GetResults");
  if (s->device_form) {
    PF_CHECK(!s->owner->disposed(), PF_ERR_DISPOSED, "OfflineRecognizer");
    s->materialize();                                               // device form -> host form (features read back)
  }
  *n_floats = (int32_t)s->Speech.size();
  PF_CHECK((int64_t)s->Speech.size() <= cap, PF_ERR_CAPACITY, "speech buffer too small");
  if (!s->Speech.empty()) { NEED(out); std::memcpy(out, s->Speech.data(), s->Speech.size() * 4); }
  return PF_OK;
  PF_CATCH
}
int pf_stream_set_speech(pf_stream* h, const float* speech, int32_t n_floats, int32_t speech_length) {
  PF_TRY
  Stream* s = S(h);
  s->drop_device_audio();
  s->pending.clear();
  if (n_floats < 0) {                                               // Speech = null
    std::vector<float>().swap(s->Speech);
    s->has_speech = false;
  } else {
    if (n_floats > 0) NEED(speech);
    s->Speech.assign(speech, speech + n_floats);
    s->has_speech = true;
  }
  s->SpeechLength = speech_length;
  return PF_OK;
  PF_CATCH
}

// ---- streaming path ---------------------------------------------------------
static thread_local std::vector<std::string> t_online_texts;

int pf_online_recognizer_create(const char* enc, const char* dec, const char* config, const char* mvn, const char* tokens,
                                int32_t threads, int32_t device, pf_online_recognizer** out) {
  PF_TRY
  NEED(out);
  *out = nullptr;
  auto s = [](const char* p) { return std::string(p ? p : ""); };
  std::shared_ptr<OnlineRecognizerM> r = std::make_shared<OnlineRecognizerM>(s(enc), s(dec), s(config), s(mvn), s(tokens), threads, device);
  pf_online_recognizer* h = new pf_online_recognizer();
  h->r = r;
  h->eng.e = r->engine();
  *out = h;
  return PF_OK;
  PF_CATCH
}
void pf_online_recognizer_dispose(pf_online_recognizer* h) {
  if (!h) return;
  std::shared_ptr<OnlineRecognizerM> r;
  { std::lock_guard<std::mutex> lk(h->mu); r = h->r; }
  if (!r) return;
  { std::lock_guard<std::mutex> lk(h->eng.mu); h->eng.e.reset(); }
  try { r->Dispose(); } catch (...) {}
}
void pf_online_recognizer_free(pf_online_recognizer* h) {
  if (!h) return;
  pf_online_recognizer_dispose(h);
  std::lock_guard<std::mutex> lk(h->mu);
  h->r.reset();
}
pf_engine* pf_online_recognizer_engine(pf_online_recognizer* h) {
  if (!h) return nullptr;
  std::lock_guard<std::mutex> lk(h->eng.mu);
  return h->eng.e ? &h->eng : nullptr;
}
static std::shared_ptr<OnlineRecognizerM> OR(pf_online_recognizer* h) {
  PF_CHECK(h != nullptr, PF_ERR_INVALID_ARG, "null recognizer");
  std::lock_guard<std::mutex> lk(h->mu);
  PF_CHECK(h->r != nullptr, PF_ERR_DISPOSED, "OnlineRecognizer");
  return h->r;
}
static OnlineStreamM* OS(pf_online_stream* h) {
  PF_CHECK(h != nullptr, PF_ERR_INVALID_ARG, "null stream");
  PF_CHECK(!h->freed && h->s != nullptr && !h->s->disposed, PF_ERR_DISPOSED, "OnlineStream");
  return h->s.get();
}
int pf_online_create_stream(pf_online_recognizer* h, pf_online_stream** out) {
  PF_TRY
  NEED(out);
  *out = nullptr;
  std::shared_ptr<OnlineStreamM> s = OR(h)->CreateOnlineStream();
  pf_online_stream* sh = online_stream_shells().get();
  sh->s = s;
  *out = sh;
  return PF_OK;
  PF_CATCH
}
int pf_online_stream_add_samples(pf_online_stream* h, const float* samples, int64_t n) {
  PF_TRY
  OnlineStreamM* s = OS(h);
  PF_CHECK(!s->owner->disposed(), PF_ERR_DISPOSED, "OnlineRecognizer");
  PF_CHECK(n >= 0, PF_ERR_INVALID_ARG, "negative sample count");
  s->AddSamples(samples, n);
  return PF_OK;
  PF_CATCH
}
int pf_online_get_results(pf_online_recognizer* h, pf_online_stream* const* streams, int32_t n) {
  PF_TRY
  std::shared_ptr<OnlineRecognizerM> r = OR(h);
  std::vector<OnlineStreamM*> ss;
  for (int i = 0; i < n; ++i) { NEED(streams); ss.push_back(OS(streams[i])); }
  t_online_texts = r->GetResults(ss);
  return PF_OK;
  PF_CATCH
}
int pf_online_result_text(pf_online_recognizer* h, int32_t i, const char** utf8) {
  PF_TRY
  OR(h);
  NEED(utf8);
  PF_CHECK(i >= 0 && i < (int32_t)t_online_texts.size(), PF_ERR_INVALID_ARG, "result index out of range");
  *utf8 = t_online_texts[(size_t)i].c_str();
  return PF_OK;
  PF_CATCH
}
int pf_online_stream_tokens(pf_online_stream* h, const int64_t** ids, int32_t* n) {
  PF_TRY
  OnlineStreamM* s = OS(h);
  if (ids) *ids = s->Tokens.data();
  if (n) *n = (int32_t)s->Tokens.size();
  return PF_OK;
  PF_CATCH
}
void pf_online_stream_dispose(pf_online_stream* h) {
  if (!h || h->freed || !h->s) return;
  h->s->disposed = true;
  std::vector<float>().swap(h->s->Speech);
}
void pf_online_stream_free(pf_online_stream* h) {
  if (!h || h->freed) return;
  if (h->s) h->s->disposed = true;
  h->s.reset();
  h->freed = true;
  online_stream_shells().retire(h);
}
int pf_online_encoder(pf_engine* h, const float* speech, int32_t B, int32_t Tc, float* enc_out, float* alphas_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  std::lock_guard<std::mutex> lk(eh_->mutex());
  eh_->online_encoder(speech, B, Tc, enc_out, alphas_out);
  return PF_OK;
  PF_CATCH
}
int pf_online_decoder(pf_engine* h, const float* enc, int32_t B, int32_t Tc, const float* embeds, int32_t L,
                      const int32_t* embeds_len, const float* caches_in, int32_t n_caches, float* logits_out,
                      int64_t* ids_out, float* caches_out) {
  PF_TRY
  std::shared_ptr<Engine> eh_ = E(h);
  std::lock_guard<std::mutex> lk(eh_->mutex());
  PF_CHECK(n_caches == eh_->dec_layers(), PF_ERR_INVALID_ARG, "online_decoder: one cache per decoder layer expected");
  eh_->online_decoder(enc, B, Tc, embeds, L, embeds_len, caches_in, logits_out, ids_out, caches_out);
  return PF_OK;
  PF_CATCH
}
int pf_host_online_lfr(const float* fbank, int32_t t80, int32_t lfr_m, int32_t lfr_n, float* out, int64_t cap, int32_t* t_lfr) {
  PF_TRY
  NEED(t_lfr);
  PF_CHECK(t80 >= 0 && (t80 == 0 || fbank) && lfr_m >= 1 && lfr_n >= 1, PF_ERR_INVALID_ARG, "online_lfr: bad arguments");
  std::vector<float> in(fbank, fbank + (size_t)t80 * 80);
  std::vector<float> o = online_apply_lfr(in, 80, lfr_m, lfr_n);
  *t_lfr = (int32_t)(o.size() / ((size_t)lfr_m * 80));
  if (out) {
    PF_CHECK(cap >= (int64_t)o.size(), PF_ERR_CAPACITY, "online_lfr: output capacity too small");
    if (!o.empty()) std::memcpy(out, o.data(), o.size() * 4);
  }
  return PF_OK;
  PF_CATCH
}
int pf_host_online_posenc(float* x, int32_t timesteps, int32_t dim, int32_t start_idx) {
  PF_TRY
  NEED(x);
  PF_CHECK(timesteps >= 0 && dim >= 4 && dim % 2 == 0 && start_idx >= 0, PF_ERR_INVALID_ARG, "online_posenc: bad arguments");
  std::vector<float> v(x, x + (size_t)timesteps * dim);
  online_position_encode(v, timesteps, dim, start_idx);
  if (!v.empty()) std::memcpy(x, v.data(), v.size() * 4);
  return PF_OK;
  PF_CATCH
}
int pf_host_online_dynamic_mask(float* alphas, int32_t n) {
  PF_TRY
  NEED(alphas);
  PF_CHECK(n >= 0, PF_ERR_INVALID_ARG, "online_dynamic_mask: negative length");
  std::vector<float> v(alphas, alphas + n);
  online_dynamic_mask(v);
  if (!v.empty()) std::memcpy(alphas, v.data(), v.size() * 4);
  return PF_OK;
  PF_CATCH
}
int pf_host_online_cif(const float* hiddens, const float* alphas, int32_t n, int32_t D, float threshold, float* fired,
                       int32_t fired_cap, int32_t* n_fired, float* carry_alpha, float* carry_hidden) {
  PF_TRY
  NEED(n_fired); NEED(carry_alpha); NEED(carry_hidden);
  PF_CHECK(n >= 0 && D > 0 && fired_cap >= 0, PF_ERR_INVALID_ARG, "online_cif: bad sizes");
  if (n > 0) { NEED(hiddens); NEED(alphas); }
  std::vector<std::vector<float>> h;
  for (int i = 0; i < n; ++i) h.emplace_back(hiddens + (size_t)i * D, hiddens + (size_t)(i + 1) * D);
  std::vector<float> a(alphas, alphas + n), ch;
  std::vector<std::vector<float>> f;
  float ca = 0.f;
  online_cif(h, a, threshold, f, ca, ch);
  *n_fired = (int32_t)f.size();
  *carry_alpha = ca;
  std::memset(carry_hidden, 0, (size_t)D * 4);                 // n = 0: nothing was integrated, the carried frame is empty
  if (!ch.empty()) std::memcpy(carry_hidden, ch.data(), std::min(ch.size(), (size_t)D) * 4);
  PF_CHECK((int32_t)f.size() <= fired_cap, PF_ERR_CAPACITY, "online_cif: fired capacity too small");
  for (size_t l = 0; l < f.size(); ++l) { NEED(fired); std::memcpy(fired + l * D, f[l].data(), (size_t)D * 4); }
  return PF_OK;
  PF_CATCH
}
int pf_host_online_decode(const char* const* tokens, int32_t n_tokens, const int64_t* ids, int32_t n_ids, char* out, int32_t cap) {
  PF_TRY
  NEED(out);
  PF_CHECK(n_tokens >= 0 && n_ids >= 0 && cap > 0, PF_ERR_INVALID_ARG, "online_decode: bad sizes");
  if (n_tokens > 0) NEED(tokens);
  if (n_ids > 0) NEED(ids);
  std::vector<std::string> tk;
  for (int i = 0; i < n_tokens; ++i) { PF_CHECK(tokens[i] != nullptr, PF_ERR_INVALID_ARG, "online_decode: null token"); tk.emplace_back(tokens[i]); }
  std::vector<int64_t> idv(ids, ids + n_ids);
  const std::string t = online_decode_text(tk, idv);
  PF_CHECK((int32_t)t.size() + 1 <= cap, PF_ERR_CAPACITY, "online_decode: output capacity too small");
  std::memcpy(out, t.c_str(), t.size() + 1);
  return PF_OK;
  PF_CATCH
}

// ---- host text stage --------------------------------------------------------
int pf_host_timestamps(const float* peak, int32_t n, const int64_t* tokens, int32_t n_tokens, int32_t* out_pairs,
                       int32_t cap_pairs) {
  PF_TRY
  NEED(peak);
  std::vector<int64_t> tk;
  if (n_tokens > 0) { NEED(tokens); tk.assign(tokens, tokens + n_tokens); }
  auto ts = time_stamp_lfr6(peak, n, tk);
  PF_CHECK((int32_t)ts.size() <= cap_pairs, PF_ERR_CAPACITY, "timestamp capacity too small");
  for (size_t i = 0; i < ts.size(); ++i) { out_pairs[2 * i] = ts[i][0]; out_pairs[2 * i + 1] = ts[i][1]; }
  return (int)ts.size();
  PF_CATCH
}

int pf_host_hotword_ids(const char* const* tokens, int32_t n_tokens, const char* const* lines, int32_t n_lines,
                        int32_t* ids, int32_t ids_cap, int32_t* lens, int32_t lens_cap, int32_t* n_hotwords) {
  PF_TRY
  NEED(n_hotwords);
  std::vector<std::string> tk, ln;
  for (int i = 0; i < n_tokens; ++i) tk.emplace_back(tokens[i]);
  for (int i = 0; i < n_lines; ++i) ln.emplace_back(lines[i]);
  auto hw = hotword_ids(tk, ln, 1);
  size_t tot = 0;
  for (auto& w : hw) tot += w.size();
  PF_CHECK((int32_t)hw.size() <= lens_cap && (int64_t)tot <= ids_cap, PF_ERR_CAPACITY, "hotword buffers too small");
  size_t off = 0;
  for (size_t i = 0; i < hw.size(); ++i) {
    lens[i] = (int32_t)hw[i].size();
    for (int32_t v : hw[i]) ids[off++] = v;
  }
  *n_hotwords = (int32_t)hw.size();
  return PF_OK;
  PF_CATCH
}

int pf_host_decode(const char* const* tokens, int32_t n_tokens, const int64_t* ids, int32_t n_ids,
                   const int32_t* ts_ints, const int32_t* ts_lens, int32_t n_ts, pf_decoded** out) {
  PF_TRY
  NEED(out);
  *out = nullptr;
  std::vector<std::string> tk;
  for (int i = 0; i < n_tokens; ++i) tk.emplace_back(tokens[i]);
  std::vector<int64_t> idv(ids, ids + n_ids);
  TsList ts;
  size_t off = 0;
  for (int i = 0; i < n_ts; ++i) {
    ts.emplace_back(ts_ints + off, ts_ints + off + ts_lens[i]);
    off += (size_t)ts_lens[i];
  }
  pf_decoded* d = new pf_decoded();
  try {
    // both forms of DecodeMulti — per-token derivations on the fly, and out of the table a recognizer builds once — must agree:
    // this entry is what the golden vectors and the fuzz harness drive, the recognizer itself uses the table
    d->r = decode_multi_one(tk, idv, ts);
    const ResultEntity t = decode_multi_one(TokenTable(tk), idv, ts);
    PF_CHECK(t.Text == d->r.Text && t.TextLen == d->r.TextLen && t.Tokens == d->r.Tokens && t.Timestamps == d->r.Timestamps,
             PF_ERR_RECOGNITION, "DecodeMulti: the table form disagrees with the plain form");
  } catch (...) {
    delete d;
    throw;
  }
  *out = d;
  return PF_OK;
  PF_CATCH
}
int pf_decoded_text(pf_decoded* d, const char** utf8, int32_t* len16) {
  PF_TRY
  NEED(d);
  if (utf8) *utf8 = d->r.Text.c_str();
  if (len16) *len16 = d->r.TextLen;
  return PF_OK;
  PF_CATCH
}
int pf_decoded_num_tokens(pf_decoded* d, int32_t* n) {
  PF_TRY
  NEED(d); NEED(n);
  *n = (int32_t)d->r.Tokens.size();
  return PF_OK;
  PF_CATCH
}
int pf_decoded_token(pf_decoded* d, int32_t j, const char** utf8) {
  PF_TRY
  NEED(d); NEED(utf8);
  PF_CHECK(j >= 0 && j < (int32_t)d->r.Tokens.size(), PF_ERR_INVALID_ARG, "token index out of range");
  *utf8 = d->r.Tokens[(size_t)j].c_str();
  return PF_OK;
  PF_CATCH
}
int pf_decoded_num_timestamps(pf_decoded* d, int32_t* n) {
  PF_TRY
  NEED(d); NEED(n);
  *n = (int32_t)d->r.Timestamps.size();
  return PF_OK;
  PF_CATCH
}
int pf_decoded_timestamp(pf_decoded* d, int32_t j, const int32_t** ints, int32_t* n_ints) {
  PF_TRY
  NEED(d);
  PF_CHECK(j >= 0 && j < (int32_t)d->r.Timestamps.size(), PF_ERR_INVALID_ARG, "timestamp index out of range");
  if (ints) *ints = d->r.Timestamps[(size_t)j].data();
  if (n_ints) *n_ints = (int32_t)d->r.Timestamps[(size_t)j].size();
  return PF_OK;
  PF_CATCH
}
void pf_decoded_free(pf_decoded* d) { delete d; }


int pf_host_wav_read(const char* path, float* out, int64_t cap, int64_t* n_out, int32_t* sample_rate, int32_t* channels,
                     double* duration_ms) {
  PF_TRY
  NEED(path); NEED(n_out);
  double dur = 0;
  int sr = 0, ch = 0;
  std::vector<float> v;
  if (!file_exists(path)) {
    v.assign(1, 0.f);                                   // GetFileSample: new float[1]
  } else {
    WavData w = decode_wav_file(path);
    sr = w.sample_rate; ch = w.channels; dur = w.duration_ms;
    v = w.sample_rate != 16000 ? resample_linear(w.samples, w.sample_rate, 16000, w.channels) : std::move(w.samples);
  }
  *n_out = (int64_t)v.size();
  if (sample_rate) *sample_rate = sr;
  if (channels) *channels = ch;
  if (duration_ms) *duration_ms = dur;
  if (out) {
    PF_CHECK(cap >= (int64_t)v.size(), PF_ERR_CAPACITY, "wav_read: output capacity too small");
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * 4);
  }
  return PF_OK;
  PF_CATCH
}

int pf_host_resample(const float* src, int64_t n, int32_t sr_in, int32_t sr_out, int32_t channels, float* out, int64_t cap,
                     int64_t* n_out) {
  PF_TRY
  NEED(n_out);
  PF_CHECK(n >= 0 && (n == 0 || src), PF_ERR_INVALID_ARG, "resample: bad arguments");
  std::vector<float> in(src, src + n);
  std::vector<float> v = resample_linear(in, sr_in, sr_out, channels);
  *n_out = (int64_t)v.size();
  if (out) {
    PF_CHECK(cap >= (int64_t)v.size(), PF_ERR_CAPACITY, "resample: output capacity too small");
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * 4);
  }
  return PF_OK;
  PF_CATCH
}

int pf_host_is_audio(const char* path, int32_t* is_audio) {
  PF_TRY
  NEED(path); NEED(is_audio);
  *is_audio = is_wav_header(path) ? 1 : 0;
  return PF_OK;
  PF_CATCH
}
}  // extern "C"
