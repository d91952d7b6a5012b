// engine.cpp — see engine.h.  Orchestrates the gfx950 kernels into the graph the reference
// obtains from InferenceSession.Run (AliParaformerAsr/OfflineProjOfParaformer.cs:68,
// OfflineProjOfSenseVoiceSmall.cs:156): SAN-M encoder -> CIF predictor -> parallel SAN-M
// decoder -> log-softmax, followed by the reference's own last-index arg-max
// (AliParaformerAsr/OfflineRecognizer.cs:139-152).
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <set>

#include "hostutil.h"

namespace pf {

static const size_t kAlign = 256;

// live engines' main streams and the streams parked by the hardware-queue probe (Engine::own_hardware_queue below)
static std::mutex g_main_mu;
static std::vector<std::pair<int, hipStream_t>> g_main_streams;      // (device, main stream) of the live engines
static std::vector<hipStream_t> g_parked;

// ------------------------------------------------------------------ construction ----------
Engine::Engine(const pf_engine_config& cfg) {
  device_ = cfg.device;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    throw Error(PF_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
  PF_CHECK(device_ >= 0 && device_ < ndev, PF_ERR_DEVICE, "device ordinal out of range");
  PF_HIP(hipSetDevice(device_));
  hipDeviceProp_t prop;
  PF_HIP(hipGetDeviceProperties(&prop, device_));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    throw Error(PF_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  cus_ = cu_limit(prop.multiProcessorCount);

  fc_.fs = cfg.fs > 0 ? cfg.fs : 16000;
  fc_.n_mels = cfg.n_mels > 0 ? cfg.n_mels : 80;
  fc_.lfr_m = cfg.lfr_m > 0 ? cfg.lfr_m : 7;
  fc_.lfr_n = cfg.lfr_n > 0 ? cfg.lfr_n : 6;
  fc_.snip_edges = cfg.snip_edges != 0;
  fc_.dither = cfg.dither;
  fc_.dither_seed = (uint32_t)cfg.dither_seed;
  fc_.window = cfg.window ? cfg.window : "hamming";
  PF_CHECK(fc_.dither >= 0.f && fc_.dither == fc_.dither, PF_ERR_INVALID_ARG, "dither must be >= 0");
  PF_CHECK(fc_.fs == 16000, PF_ERR_UNSUPPORTED, "only fs = 16000 is supported");
  // the fbank kernel is built for 25 ms / 10 ms frames (400 / 160 samples, 512-point FFT): any other framing in
  // the configuration must be refused, not silently ignored
  PF_CHECK((cfg.frame_length_ms == 0 || cfg.frame_length_ms == 25) && (cfg.frame_shift_ms == 0 || cfg.frame_shift_ms == 10),
           PF_ERR_UNSUPPORTED, "only frame_length = 25 ms and frame_shift = 10 ms are supported");
  PF_CHECK(cfg.math_mode >= 0 && cfg.math_mode <= 3, PF_ERR_INVALID_ARG,
           "math_mode must be 0 (f16 MFMA), 1 (fp32 MFMA), 2 (dynamic int8 as model.int8.onnx, int8 MFMA) or 3 (exact: split-f16 products)");
  fp32_mode_ = cfg.math_mode == 1 || cfg.math_mode == 3;
  x3_mode_ = cfg.math_mode == 3;
  // math_mode 3 keeps the fp32-MFMA attention by default: with x3 operands in the attention too (PF_X3_ATTN=1: 47.0 instead of
  // 50.8 ms per 32 x 30 s) 2 of the 5344 ids of the benchmark batch leave the fp32 oracle's — scores are exponentiated, and a
  // 22-bit product is 4x an fp32 product's error; the Linears tolerate it (identity holds), the softmax does not
  { const char* e = getenv("PF_X3_ATTN"); if (e && e[0]) x3_attn_ = atoi(e); }
  { const char* e = getenv("PF_X3_FUSE"); if (e && e[0] == '0') x3_fuse_ = false; }
  { const char* e = getenv("PF_X3_ONE"); if (e && e[0] == '0') x3_one_ = false; }
  int8_mode_ = cfg.math_mode == 2;
  { const char* e = getenv("PF_NO_RC"); no_rc_ = e && e[0] == '1'; }
  { const char* e = getenv("PF_LSTM_STEPS"); lstm_steps_ = e && e[0] == '1'; }
  { const char* e = getenv("PF_DEC_FUSE"); if (e && e[0]) dec_fuse_ = atoi(e) & 7; }
  { const char* e = getenv("PF_DEC_H32"); dec_h32_ = e && e[0] == '1'; }
  { const char* e = getenv("PF_SMALL_NOFUSE"); no_small_fuse_ = e && e[0] == '1'; }   // A/B: short-input GEMMs without the FSMN epilogue / LayerNorm-in-reduction forms   // A/B switch for tools/: decoder launch fusions
  { const char* e = getenv("PF_QKV_SPLIT"); if (e && e[0]) qkv_split_ = e[0] != '0'; }   // A/B: Q | K blocked + V row-major from the 256 x 192 kernel (k_gemm_qkv.hip)
  { const char* e = getenv("PF_QKV_MIN"); if (e && e[0]) qkv_split_min_tiles_ = atoi(e); }
  { const char* e = getenv("PF_QKV_FILL"); if (e && e[0]) qkv_split_min_fill_ = atoi(e); }
  { const char* e = getenv("PF_RC_FFN2"); if (e && e[0]) rc_ffn2_ = e[0] != '0'; }   // A/B switch for tools/: the unfused encoder sequence
  { const char* e = getenv("PF_FFN_FUSED"); if (e && e[0]) ffn_fused_ = e[0] != '0'; }   // A/B: the whole FFN block in one launch (k_ffn.hip)
  { const char* e = getenv("PF_FFN_MIN"); if (e && e[0]) ffn_fused_min_rows_ = atoi(e); }
  { const char* e = getenv("PF_DEC_FFN"); if (e && e[0]) dec_ffn_fused_ = e[0] != '0'; }
  { const char* e = getenv("PF_DEC_OUT_CHAIN"); if (e && e[0]) dec_out_chain_ = e[0] != '0'; }
  { const char* e = getenv("PF_DEC_MID"); if (e && e[0]) dec_mid_ = e[0] != '0'; }     // A/B: finishing pass + norm2 + FSMN + norm3 + q in one launch (k_decmid.hip)
  { const char* e = getenv("PF_ATTN_FFN"); if (e && e[0]) attn_ffn_ = e[0] != '0'; }
  { const char* e = getenv("PF_QKV_TAIL"); if (e && e[0]) qkv_tail_ = e[0] != '0'; }

  // host-only validation BEFORE anything is uploaded (a bad am.mvn must not cost a 0.9 GB upload per retry)
  std::vector<float> shift, scale;
  if (cfg.mvn_path && cfg.mvn_path[0]) {
    parse_mvn_text(read_text_file(cfg.mvn_path), shift, scale);
  } else if (cfg.cmvn_shift && cfg.cmvn_scale && cfg.cmvn_dim > 0) {
    shift.assign(cfg.cmvn_shift, cfg.cmvn_shift + cfg.cmvn_dim);
    scale.assign(cfg.cmvn_scale, cfg.cmvn_scale + cfg.cmvn_dim);
  }
  if (!shift.empty()) {
    PF_CHECK(shift.size() == scale.size(), PF_ERR_FORMAT, "am.mvn: shift/scale length mismatch");
    PF_CHECK((int)shift.size() == fc_.lfr_m * fc_.n_mels, PF_ERR_FORMAT, "am.mvn: CMVN dim must equal lfr_m * n_mels");
  }

  uid_ = register_uid();
  try {
    PF_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    own_hardware_queue();
    PF_HIP(hipStreamCreateWithFlags(&aux_stream_, hipStreamNonBlocking));
    PF_HIP(hipEventCreateWithFlags(&ev_scan_, hipEventDisableTiming));
    PF_HIP(hipStreamCreateWithFlags(&ts_stream_, hipStreamNonBlocking));
    own_hardware_queue_ts();
    PF_HIP(hipEventCreateWithFlags(&ev_ts_, hipEventDisableTiming));
    PF_HIP(hipEventCreateWithFlags(&ev_enc_, hipEventDisableTiming));
    load_weights(cfg);
    mc_.use_itn = cfg.use_itn != 0 || mc_.use_itn;
    fb_ = fbank_tables_create(fc_.n_mels, fc_.fs, fc_.window.c_str());
    // split partials of the short-input GEMM (k_gemm_small.hip): one per engine = one per stream
    small_ws_ = (float*)dalloc(gemm_small_ws_bytes());
    if (!shift.empty()) {
      cmvn_dim_ = (int)shift.size();
      cmvn_shift_ = (float*)dalloc(sizeof(float) * cmvn_dim_);
      cmvn_scale_ = (float*)dalloc(sizeof(float) * cmvn_dim_);
      PF_HIP(hipMemcpy(cmvn_shift_, shift.data(), sizeof(float) * cmvn_dim_, hipMemcpyHostToDevice));
      PF_HIP(hipMemcpy(cmvn_scale_, scale.data(), sizeof(float) * cmvn_dim_, hipMemcpyHostToDevice));
    }
  } catch (...) {
    release();                                       // ~Engine does not run for a throwing constructor
    throw;
  }
}

Engine::~Engine() { release(); }

void Engine::release() {
  if (uid_) { unregister_uid(uid_); uid_ = 0; }
  hipSetDevice(device_);
  if (stream_) hipStreamSynchronize(stream_);
  if (aux_stream_) { hipStreamSynchronize(aux_stream_); hipStreamDestroy(aux_stream_); aux_stream_ = nullptr; }
  if (plan_host_) { hipHostFree(plan_host_); plan_host_ = nullptr; plan_host_dev_ = nullptr; plan_host_cap_ = 0; }
  if (ev_scan_) { hipEventDestroy(ev_scan_); ev_scan_ = nullptr; }
  if (ts_stream_) { hipStreamSynchronize(ts_stream_); hipStreamDestroy(ts_stream_); ts_stream_ = nullptr; }
  if (ev_ts_) { hipEventDestroy(ev_ts_); ev_ts_ = nullptr; }
  if (ev_enc_) { hipEventDestroy(ev_enc_); ev_enc_ = nullptr; }
  if (lstm_graph_exec_) { hipGraphExecDestroy(lstm_graph_exec_); lstm_graph_exec_ = nullptr; }
  profile_reset();
  if (fb_) { fbank_tables_destroy(fb_); fb_ = nullptr; }
  for (void* p : owned_) hipFree(p);
  owned_.clear();
  DevBuf* bufs[] = {&ws_f32_, &ws_audio_, &ws_meta_, &ws_fbank_, &ws_speech_, &ws_enc_, &ws_dec_, &ws_kv_, &ws_pe_, &ws_tmp_,
                    &ws_ts_, &ws_seaco_, &ws_seaco_in_, &ws_seaco_hw_, &ws_q_, &ws_qf_, &ws_seaco_q_, &ws_x3a_, &ws_x3t_, &ws_x3h_};
  x3_pair_live_ = false; x3a_src_ = nullptr; x3a_pair_only_ = false;
  x3w_.clear();
  ts_whh_x3_ = nullptr;
  seaco_hw_valid_ = false;
  for (DevBuf* b : bufs)
    if (b->p) { hipFree(b->p); b->p = nullptr; b->bytes = 0; }
  if (blob_owned_ && blob_dev_) hipFree(blob_dev_);
  blob_dev_ = nullptr;
  if (stream_) {
    {
      std::lock_guard<std::mutex> lk(g_main_mu);
      for (size_t i = 0; i < g_main_streams.size(); ++i)
        if (g_main_streams[i].second == stream_) { g_main_streams.erase(g_main_streams.begin() + (long)i); break; }
      if (g_main_streams.empty()) {                           // the last engine of the process: the parked streams go too
        for (hipStream_t s : g_parked) hipStreamDestroy(s);
        g_parked.clear();
      }
    }
    hipStreamDestroy(stream_); stream_ = nullptr;
  }
}

// Engines on one device overlap (the recognizer's pool, bench.py's steps in flight) only if their main streams sit on DIFFERENT
// hardware queues: HIP multiplexes its streams onto GPU_MAX_HW_QUEUES (4) queues, and two streams on one queue run their kernels in
// submission order.  Which queue a new stream gets depends on every stream the process has created and destroyed before — measured
// in round 6: the second pair of engines of a process (bench.py's `exact` twin behind the f16 engines) did not overlap at all (38.6 ms
// per step with two in flight against 35.1 in a fresh process; GPU_MAX_HW_QUEUES=8 brought 34.9 back).  So the constructor PROBES: a
// single wave spins ~150 us on a live engine's main stream while an empty kernel goes to the new stream; if the empty kernel does not
// finish well before the spinning one, the two streams share a queue: the new stream is parked (kept alive, so that the runtime's
// allocator moves on) and another one is tried, up to 8 times.  PF_QUEUE_PROBE=0 switches the probe off.

namespace {
struct ProbeEvents {                                          // (the probe's calls may throw: the events go with the scope)
  hipEvent_t a = nullptr, b = nullptr;
  ProbeEvents() { PF_HIP(hipEventCreate(&a)); if (hipEventCreate(&b) != hipSuccess) { hipEventDestroy(a); throw Error(PF_ERR_DEVICE, "hipEventCreate"); } }
  ~ProbeEvents() { hipEventDestroy(a); hipEventDestroy(b); }
};
// a stream parked by the probe has done its work once the next stream exists: keep the last few, destroy the oldest (a process that
// creates and destroys recognizers beside a long-lived one would otherwise collect them for ever)
void park_stream(hipStream_t s) {
  g_parked.push_back(s);
  if (g_parked.size() > 24) {
    for (int i = 0; i < 8; ++i) hipStreamDestroy(g_parked[(size_t)i]);
    g_parked.erase(g_parked.begin(), g_parked.begin() + 8);
  }
}
}  // namespace

void Engine::own_hardware_queue() {
  static const int on = env_int("PF_QUEUE_PROBE", 1);
  std::lock_guard<std::mutex> lk(g_main_mu);
  if (on) {
    ProbeEvents ev;
    hipEvent_t ea = ev.a, eb = ev.b;
    for (int attempt = 0; attempt < 8; ++attempt) {
      bool clash = false;
      int seen = 0;
      for (auto it = g_main_streams.rbegin(); it != g_main_streams.rend() && seen < 3 && !clash; ++it) {
        if (it->first != device_) continue;
        ++seen;
        launch_spin(it->second, 300000ull);                  // ~150 us at ~2 GHz, one wave: every other CU stays free
        launch_nop(stream_);
        PF_HIP(hipEventRecord(eb, stream_));
        PF_HIP(hipEventRecord(ea, it->second));
        PF_HIP(hipEventSynchronize(ea));
        PF_HIP(hipEventSynchronize(eb));
        float ms = 0.f;
        PF_HIP(hipEventElapsedTime(&ms, eb, ea));            // how long before the spinning kernel's end the empty one was done
        clash = ms < 0.05f;
      }
      if (!clash) {
        if (attempt && getenv("PF_QUEUE_PROBE_VERBOSE")) fprintf(stderr, "pf: engine main stream moved to another hardware queue after %d attempt(s)\n", attempt);
        break;
      }
      park_stream(stream_);                                  // stays alive: the next stream gets another queue
      stream_ = nullptr;
      PF_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    }
  }
  g_main_streams.emplace_back(device_, stream_);
}

// the timestamp head runs BESIDE the decoder on ts_stream_: the same requirement against the engine's own main stream
void Engine::own_hardware_queue_ts() {
  static const int on = env_int("PF_QUEUE_PROBE", 1);
  if (!on || !ts_stream_) return;
  std::lock_guard<std::mutex> lk(g_main_mu);
  ProbeEvents ev;
  hipEvent_t ea = ev.a, eb = ev.b;
  for (int attempt = 0; attempt < 8; ++attempt) {
    launch_spin(stream_, 300000ull);
    launch_nop(ts_stream_);
    PF_HIP(hipEventRecord(eb, ts_stream_));
    PF_HIP(hipEventRecord(ea, stream_));
    PF_HIP(hipEventSynchronize(ea));
    PF_HIP(hipEventSynchronize(eb));
    float ms = 0.f;
    PF_HIP(hipEventElapsedTime(&ms, eb, ea));
    if (ms >= 0.05f) {
      if (attempt && getenv("PF_QUEUE_PROBE_VERBOSE")) fprintf(stderr, "pf: timestamp stream moved to another hardware queue after %d attempt(s)\n", attempt);
      break;
    }
    park_stream(ts_stream_);
    ts_stream_ = nullptr;
    PF_HIP(hipStreamCreateWithFlags(&ts_stream_, hipStreamNonBlocking));
  }
}

// pinned host memory the CIF plan's counts are exported into (PF_PLAN_ZERO_COPY=0: the three device-to-host copies of rounds 1-5)
void Engine::ensure_plan_host(int B) {
  static const int on = env_int("PF_PLAN_ZERO_COPY", 1);
  if (!on || plan_host_cap_ >= B) return;
  if (plan_host_) { hipHostFree(plan_host_); plan_host_ = nullptr; plan_host_dev_ = nullptr; plan_host_cap_ = 0; }
  const int cap = std::max(64, (int)round_up(B, 64));
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)(1 + 2 * cap) * 4, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return; }
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); hipHostFree(p); return; }
  plan_host_ = (int32_t*)p; plan_host_dev_ = (int32_t*)d; plan_host_cap_ = cap;
}

// The CIF plan's counts on their way to the host (all three arithmetic modes).  export_plan goes right behind the scan: the counts
// are written into pinned host memory by a kernel and ev_scan_ is recorded; read_back_plan waits for that event alone and fills
// last_.fire_count / last_.token_num.  Without pinned memory (or with PF_PLAN_ZERO_COPY=0): three copies on the side stream.
void Engine::export_plan(int B) {
  ensure_plan_host(B);
  if (plan_host_) launch_export_plan(stream_, plan_.max_count, plan_.fire_count, plan_.token_num, B, plan_host_dev_);
  PF_HIP(hipEventRecord(ev_scan_, stream_));
}

int32_t Engine::read_back_plan(int B) {
  int32_t L = 0;
  last_.fire_count.resize(B);
  last_.token_num.resize(B);
  if (plan_host_) {
    PF_HIP(hipEventSynchronize(ev_scan_));
    L = plan_host_[0];
    std::memcpy(last_.fire_count.data(), plan_host_ + 1, (size_t)B * 4);
    std::memcpy(last_.token_num.data(), plan_host_ + 1 + B, (size_t)B * 4);
  } else {
    PF_HIP(hipStreamWaitEvent(aux_stream_, ev_scan_, 0));
    PF_HIP(hipMemcpyAsync(&L, plan_.max_count, 4, hipMemcpyDeviceToHost, aux_stream_));
    PF_HIP(hipMemcpyAsync(last_.fire_count.data(), plan_.fire_count, (size_t)B * 4, hipMemcpyDeviceToHost, aux_stream_));
    PF_HIP(hipMemcpyAsync(last_.token_num.data(), plan_.token_num, (size_t)B * 4, hipMemcpyDeviceToHost, aux_stream_));
    PF_HIP(hipStreamSynchronize(aux_stream_));
  }
  return L;
}

void* Engine::dalloc(size_t bytes) {
  void* p = nullptr;
  PF_HIP(hipMalloc(&p, std::max<size_t>(bytes, 256)));
  owned_.push_back(p);
  return p;
}

void Engine::ensure(DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return;
  if (b.p) {
    PF_HIP(hipStreamSynchronize(stream_));
    if (ts_stream_) PF_HIP(hipStreamSynchronize(ts_stream_));   // (a call that threw may have left the timestamp head unjoined)
    PF_HIP(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  const size_t want = round_up((int64_t)(bytes + bytes / 8), 1 << 20);
  PF_HIP(hipMalloc(&b.p, want));
  PF_HIP(hipMemsetAsync(b.p, 0, want, stream_));
  b.bytes = want;
}

void Engine::sync() {
  PF_HIP(hipStreamSynchronize(stream_));
  check_async_errors();
}

void Engine::check_async_errors() {
  if (!lstm_err_) return;                            // the persistent recurrence raises this word when a spin timed out
  unsigned flag = 0;
  unsigned* w = lstm_err_;
  lstm_err_ = nullptr;
  PF_HIP(hipMemcpy(&flag, w, 4, hipMemcpyDeviceToHost));
  PF_CHECK(flag == 0, PF_ERR_DEVICE, "timestamp head: the persistent LSTM timed out waiting for a workgroup");
}

// ------------------------------------------------------------------ weights ---------------
const Engine::Tensor& Engine::tensor(const std::string& name) const {
  auto it = tensors_.find(name);
  if (it == tensors_.end()) throw Error(PF_ERR_FORMAT, "weights: missing tensor '" + name + "'");
  if (it->second.u8) throw Error(PF_ERR_FORMAT, "weights: tensor '" + name + "' must be f32");
  return it->second;
}

const Engine::Tensor* Engine::tensor_u8(const std::string& name) const {
  auto it = tensors_.find(name);
  if (it == tensors_.end()) return nullptr;
  if (!it->second.u8) throw Error(PF_ERR_FORMAT, "weights: tensor '" + name + "' must be u8");
  return &it->second;
}

void Engine::load_weights(const pf_engine_config& cfg) {
  std::vector<char> file;
  const char* host = nullptr;
  int64_t nbytes = 0;
  std::vector<char> head;          // header bytes when the image lives on the device
  const char* dev_image = nullptr;
  if (cfg.weights_path && cfg.weights_path[0]) {
    read_binary_file(cfg.weights_path, file);
    host = file.data();
    nbytes = (int64_t)file.size();
  } else if (cfg.weights_host) {
    host = (const char*)cfg.weights_host;
    nbytes = cfg.weights_bytes;
  } else if (cfg.weights_device) {
    dev_image = (const char*)cfg.weights_device;
    nbytes = cfg.weights_bytes;
    PF_CHECK(nbytes >= 16, PF_ERR_FORMAT, "weights: image too small");
    head.resize(16);
    PF_HIP(hipMemcpy(head.data(), dev_image, 16, hipMemcpyDeviceToHost));
  } else {
    throw Error(PF_ERR_INVALID_ARG, "weights: no source given (weights_path / weights_host / weights_device)");
  }
  PF_CHECK(nbytes >= 16, PF_ERR_FORMAT, "weights: image too small");
  const char* h16 = host ? host : head.data();
  PF_CHECK(std::memcmp(h16, "PFW1", 4) == 0, PF_ERR_FORMAT, "weights: bad magic (expected PFW1 container)");
  uint64_t hlen = 0;
  std::memcpy(&hlen, h16 + 8, 8);
  PF_CHECK(hlen <= (uint64_t)nbytes - 16u, PF_ERR_FORMAT, "weights: truncated header");   // unsigned: a huge hlen cannot wrap
  std::string hdr;
  if (host) hdr.assign(host + 16, hlen);
  else {
    hdr.resize(hlen);
    PF_HIP(hipMemcpy(&hdr[0], dev_image + 16, hlen, hipMemcpyDeviceToHost));
  }
  Json j = JsonParser(hdr.data(), hdr.size()).parse();
  const Json* jc = j.get("config");
  const Json* jt = j.get("tensors");
  PF_CHECK(jc && jt && jt->type == Json::Arr, PF_ERR_FORMAT, "weights: header lacks config/tensors");
  mc_.kind = jc->str_or("kind", mc_.kind);
  mc_.feat_dim = (int)jc->int_or("feat_dim", mc_.feat_dim, INT32_MIN, INT32_MAX);
  mc_.d_model = (int)jc->int_or("d_model", mc_.d_model, INT32_MIN, INT32_MAX);
  mc_.heads = (int)jc->int_or("heads", mc_.heads, INT32_MIN, INT32_MAX);
  mc_.ffn = (int)jc->int_or("ffn", mc_.ffn, INT32_MIN, INT32_MAX);
  mc_.enc_layers = (int)jc->int_or("enc_layers", mc_.enc_layers, INT32_MIN, INT32_MAX);
  mc_.tp_layers = (int)jc->int_or("tp_layers", mc_.tp_layers, INT32_MIN, INT32_MAX);
  mc_.kernel = (int)jc->int_or("kernel", mc_.kernel, INT32_MIN, INT32_MAX);
  mc_.dec_layers = (int)jc->int_or("dec_layers", mc_.dec_layers, INT32_MIN, INT32_MAX);
  mc_.vocab = (int)jc->int_or("vocab", mc_.vocab, INT32_MIN, INT32_MAX);
  mc_.cif_threshold = (float)jc->num_or("cif_threshold", mc_.cif_threshold);
  mc_.cif_tail = (float)jc->num_or("cif_tail", mc_.cif_tail);
  mc_.cif_smooth = (float)jc->num_or("cif_smooth", mc_.cif_smooth);
  mc_.cif_noise = (float)jc->num_or("cif_noise", mc_.cif_noise);
  mc_.cif_l_order = (int)jc->int_or("cif_l_order", mc_.cif_l_order, INT32_MIN, INT32_MAX);
  mc_.cif_r_order = (int)jc->int_or("cif_r_order", mc_.cif_r_order, INT32_MIN, INT32_MAX);
  {
    const std::string cv = jc->str_or("cif_variant", "loop");
    PF_CHECK(cv == "loop" || cv == "cumsum", PF_ERR_UNSUPPORTED, "weights: cif_variant must be \"loop\" or \"cumsum\"");
    mc_.cif_cumsum = cv == "cumsum";
    PF_CHECK(!mc_.cif_cumsum || mc_.cif_threshold == 1.0f, PF_ERR_UNSUPPORTED, "cif_variant cumsum is defined for threshold 1.0");
  }
  mc_.timestamp_head = jc->bool_or("timestamp_head", false);
  mc_.seaco = jc->bool_or("seaco", false);
  mc_.use_itn = jc->bool_or("use_itn", false) || cfg.use_itn != 0;
  PF_CHECK(mc_.d_model == 512 && mc_.heads == 4, PF_ERR_UNSUPPORTED,
           "kernels are built for d_model = 512, 4 heads of 128");
  PF_CHECK(mc_.d_model % 8 == 0 && mc_.ffn % 64 == 0 && mc_.feat_dim % 4 == 0, PF_ERR_UNSUPPORTED,
           "unsupported model dimensions");
  mc_.cif_smooth2 = (float)jc->num_or("cif_smooth2", mc_.cif_smooth2);
  mc_.cif_noise2 = (float)jc->num_or("cif_noise2", mc_.cif_noise2);
  mc_.upsample = (int)jc->int_or("upsample", mc_.upsample, INT32_MIN, INT32_MAX);
  mc_.seaco_layers = (int)jc->int_or("seaco_layers", mc_.seaco_layers, INT32_MIN, INT32_MAX);
  mc_.seaco_ffn = (int)jc->int_or("seaco_ffn", mc_.seaco_ffn, INT32_MIN, INT32_MAX);
  mc_.seaco_kernel = (int)jc->int_or("seaco_kernel", mc_.seaco_kernel, INT32_MIN, INT32_MAX);
  mc_.seaco_lstm_layers = (int)jc->int_or("seaco_lstm_layers", mc_.seaco_lstm_layers, INT32_MIN, INT32_MAX);
  mc_.seaco_nobias = (int)jc->int_or("seaco_nobias", mc_.seaco_nobias, INT32_MIN, INT32_MAX);
  if (mc_.kind == "seacoparaformer") mc_.seaco = true;
  PF_CHECK(!mc_.seaco || (mc_.seaco_ffn % 64 == 0 && mc_.seaco_lstm_layers >= 1), PF_ERR_UNSUPPORTED,
           "seaco: unsupported dimensions");
  PF_CHECK(!mc_.timestamp_head || mc_.upsample == 3, PF_ERR_UNSUPPORTED, "timestamp head: only upsample = 3");

  const int64_t data_off = round_up((int64_t)(16 + hlen), (int64_t)kAlign);
  const int64_t data_bytes = nbytes - data_off;
  PF_CHECK(data_bytes >= 0, PF_ERR_FORMAT, "weights: truncated data section");
  const char* data_dev = nullptr;
  if (host) {
    PF_HIP(hipMalloc(&blob_dev_, std::max<int64_t>(data_bytes, 256)));
    blob_owned_ = true;
    PF_HIP(hipMemcpy(blob_dev_, host + data_off, data_bytes, hipMemcpyHostToDevice));
    data_dev = (const char*)blob_dev_;
  } else {
    data_dev = dev_image + data_off;
  }
  for (const Json& t : jt->arr) {
    Tensor tt;
    const std::string name = t.str_or("name", "");
    const std::string dt = t.str_or("dtype", "f32");
    PF_CHECK(dt == "f32" || dt == "u8", PF_ERR_FORMAT, "weights: tensor '" + name + "' has dtype '" + dt + "' (f32 and u8 are supported)");
    tt.u8 = dt == "u8";
    const int64_t off = t.int_or("offset", -1), nb = t.int_or("nbytes", -1);
    PF_CHECK(off >= 0 && nb >= 0 && off + nb <= data_bytes && off % 16 == 0, PF_ERR_FORMAT,
             "weights: bad tensor extent for '" + name + "'");
    tt.dev = (const float*)(data_dev + off);
    tt.numel = 1;
    if (const Json* sh = t.get("shape"))
      for (const Json& d : sh->arr) {
        const double dv = d.type == Json::Num ? d.num : -1.0;
        PF_CHECK(dv >= 1.0 && dv <= 2147483647.0 && dv == (double)(int64_t)dv, PF_ERR_FORMAT,
                 "weights: bad dimension in the shape of '" + name + "'");
        PF_CHECK(tt.numel <= (int64_t)1 << 40, PF_ERR_FORMAT, "weights: shape of '" + name + "' overflows");
        tt.shape.push_back((int64_t)dv);
        tt.numel *= (int64_t)dv;
      }
    PF_CHECK(tt.numel * (tt.u8 ? 1 : 4) == nb, PF_ERR_FORMAT, "weights: shape/nbytes mismatch for '" + name + "'");
    tensors_[name] = tt;
  }
  if (const Json* ex = jc->get("int8_exclude"))
    if (ex->type == Json::Arr)
      for (const Json& e : ex->arr)
        if (e.type == Json::Str && !e.str.empty()) int8_exclude_.push_back(e.str);
  for (const auto& kv : tensors_) {
    const std::string& n = kv.first;
    if (kv.second.u8 && n.size() > 9 && n.compare(n.size() - 9, 9, ".weight_q") == 0) any_stored_q_ = true;
    if (!kv.second.u8 && n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0) lin_names_[kv.second.dev] = n.substr(0, n.size() - 7);
  }

  // ---- bind layers, build f16 GEMM operands
  const int D = mc_.d_model;
  auto enc_layer = [&](const std::string& p, int d_in) {
    EncLayer L;
    L.d_in = d_in;
    L.norm1 = make_ln(p + ".norm1", d_in);
    L.qkv = make_lin(p + ".attn.qkv", true);
    L.fsmn_wT = make_fsmn_wT(p + ".attn.fsmn.weight");
    L.out = make_lin(p + ".attn.out", true);
    L.norm2 = make_ln(p + ".norm2", D);
    L.w1 = make_lin(p + ".ffn.w1", true);
    L.w2 = make_lin(p + ".ffn.w2", true);
    PF_CHECK(L.qkv.K == d_in && L.qkv.N == 3 * D, PF_ERR_FORMAT, "weights: qkv shape mismatch in " + p);
    if (qkv_split_ && D == 512 && mc_.heads == 4 && !fp32_mode_ && !int8_mode_) {
      // the same weight with its rows in the tile order of gemm_qkvp_kernel (k_gemm_qkv.hip)
      L.qkv_p = (half_t*)dalloc((size_t)1536 * L.qkv.Kpad * 2);
      L.qkv_bias_p = (float*)dalloc(1536 * 4);
      launch_qkv_permute(stream_, L.qkv.w, L.qkv.Kpad, L.qkv.bias, L.qkv_p, L.qkv_bias_p);
    }
    PF_CHECK(L.out.N == D && L.out.K == D && L.w1.K == D && L.w1.N == mc_.ffn && L.w2.N == D && L.w2.K == L.w1.N,
             PF_ERR_FORMAT, "weights: attn.out / ffn shape mismatch in " + p);
    if (ffn_fused_ && ffn_fused_applicable(D, mc_.ffn) && !fp32_mode_ && !int8_mode_ && L.w1.w && L.w2.w) {
      // W1 and W2 once more, in the fragment order the fused FFN kernel streams (4 MiB per layer)
      L.ffn_wt = (half_t*)dalloc(ffn_fused_weight_bytes());
      launch_ffn_retile(stream_, L.w1.w, L.w1.Kpad, L.w2.w, L.w2.Kpad, L.ffn_wt);
      if (attn_ffn_ && L.out.w && L.out.bias && mc_.kernel == 11) {
        L.out_wt = (half_t*)dalloc(ffn_outproj_weight_bytes());
        launch_ffn_retile_out(stream_, L.out.w, L.out.Kpad, L.out_wt);
        if (qkv_tail_ && L.qkv_p && L.qkv.Kpad == D && L.qkv.bias) {
          L.qkv_t = (half_t*)dalloc(3 * ffn_outproj_weight_bytes());
          for (int part = 0; part < 3; ++part)
            launch_ffn_retile_out(stream_, L.qkv.w + (size_t)part * D * L.qkv.Kpad, L.qkv.Kpad,
                                  L.qkv_t + (size_t)part * (ffn_outproj_weight_bytes() / 2));
        }
      }
    }
    return L;
  };
  for (int i = 0; i < mc_.enc_layers; ++i)
    enc_.push_back(enc_layer("encoder.layers." + std::to_string(i), i == 0 ? mc_.feat_dim : D));
  enc_after_ = make_ln("encoder.after_norm", D);
  for (int i = 0; i < mc_.tp_layers; ++i) tp_.push_back(enc_layer("encoder.tp_layers." + std::to_string(i), D));
  if (mc_.tp_layers) tp_norm_ = make_ln("encoder.tp_norm", D);

  if (mc_.kind == "sensevoicesmall") {
    ctc_ = make_lin("ctc", true);
    if (has_tensor("embed.weight")) {
      const Tensor& t = tensor("embed.weight");
      embed_host_.resize(t.numel);
      PF_HIP(hipMemcpy(embed_host_.data(), t.dev, t.numel * 4, hipMemcpyDeviceToHost));
      // device copy of the 4 query rows [language, 1, 2, textnorm] with the EFFECTIVE ids of the
      // reference (quirk Q7, OfflineProjOfSenseVoiceSmall.cs:57-74): language = 14 if use_itn else 15,
      // textnorm = 15 — used by the audio-in entry points (pf_recognize / pf_run_staged)
      const int W = mc_.feat_dim;
      if (t.numel >= (int64_t)16 * W) {
        const int order[4] = {mc_.use_itn ? 14 : 15, 1, 2, 15};
        sv_prompt_ = (float*)dalloc((size_t)4 * W * 4);
        for (int r = 0; r < 4; ++r)
          PF_HIP(hipMemcpy(sv_prompt_ + (size_t)r * W, embed_host_.data() + (size_t)order[r] * W, (size_t)W * 4,
                           hipMemcpyHostToDevice));
      }
    }
    return;
  }

  // CIF predictor: conv weight [out, in, k] -> GEMM operand [out][j*D + in]
  {
    const Tensor& cw = tensor("predictor.conv.weight");
    const int taps = mc_.cif_l_order + mc_.cif_r_order + 1;
    PF_CHECK(cw.shape.size() == 3 && cw.shape[0] == D && cw.shape[1] == D && cw.shape[2] == taps, PF_ERR_FORMAT,
             "weights: predictor.conv.weight shape");
    std::vector<float> hostw(cw.numel), re(cw.numel);
    PF_HIP(hipMemcpy(hostw.data(), cw.dev, cw.numel * 4, hipMemcpyDeviceToHost));
    for (int o = 0; o < D; ++o)
      for (int i = 0; i < D; ++i)
        for (int jx = 0; jx < taps; ++jx) re[(size_t)o * taps * D + (size_t)jx * D + i] = hostw[((size_t)o * D + i) * taps + jx];
    float* tmp = nullptr;
    PF_HIP(hipMalloc(&tmp, re.size() * 4));
    PF_HIP(hipMemcpy(tmp, re.data(), re.size() * 4, hipMemcpyHostToDevice));
    cif_conv_.N = D; cif_conv_.K = taps * D; cif_conv_.Kpad = (int)round_up(taps * D, 64);
    const int npad = (int)round_up(D, 128);
    cif_conv_.w = (half_t*)dalloc((size_t)npad * cif_conv_.Kpad * 2);
    PF_HIP(hipMemsetAsync(cif_conv_.w, 0, (size_t)npad * cif_conv_.Kpad * 2, stream_));
    launch_f32_to_f16(stream_, tmp, D, taps * D, taps * D, cif_conv_.w, cif_conv_.Kpad);
    PF_HIP(hipStreamSynchronize(stream_));
    if (fp32_mode_) { cif_conv_w32_ = tmp; owned_.push_back(tmp); }
    else PF_HIP(hipFree(tmp));
    cif_conv_.bias = tensor("predictor.conv.bias").dev;
    PF_CHECK(tensor("predictor.conv.bias").numel == D && tensor("predictor.out.weight").numel == D &&
                 tensor("predictor.out.bias").numel == 1,
             PF_ERR_FORMAT, "weights: predictor.conv.bias / predictor.out.* sizes");
    cif_out_w_ = tensor("predictor.out.weight").dev;
    cif_out_b_ = tensor("predictor.out.bias").dev;
  }
  // BiCIF timestamp head: ConvTranspose1d weight [in, out, j] -> GEMM operand [(j, out)][in];
  // W_ih of both directions stacked [8D, D] with b_ih + b_hh folded into one bias; W_hh f16 [2][4D][D].
  if (mc_.timestamp_head) {
    const int up = mc_.upsample;
    const Tensor& uw = tensor("predictor.upsample.weight");
    const Tensor& ub = tensor("predictor.upsample.bias");
    PF_CHECK(uw.shape.size() == 3 && uw.shape[0] == D && uw.shape[1] == D && uw.shape[2] == up && ub.numel == D,
             PF_ERR_FORMAT, "weights: predictor.upsample shape");
    std::vector<float> hw(uw.numel), re(uw.numel), hb(D), rb((size_t)up * D);
    PF_HIP(hipMemcpy(hw.data(), uw.dev, uw.numel * 4, hipMemcpyDeviceToHost));
    PF_HIP(hipMemcpy(hb.data(), ub.dev, (size_t)D * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < D; ++i)
      for (int o = 0; o < D; ++o)
        for (int jx = 0; jx < up; ++jx) re[((size_t)jx * D + o) * D + i] = hw[((size_t)i * D + o) * up + jx];
    for (int jx = 0; jx < up; ++jx) std::memcpy(&rb[(size_t)jx * D], hb.data(), (size_t)D * 4);
    float* tmp = nullptr;
    PF_HIP(hipMalloc(&tmp, std::max(re.size(), (size_t)8 * D * D) * 4));
    PF_HIP(hipMemcpy(tmp, re.data(), re.size() * 4, hipMemcpyHostToDevice));
    ts_up_.N = up * D; ts_up_.K = D; ts_up_.Kpad = D;
    ts_up_.w = (half_t*)dalloc((size_t)round_up(up * D, 128) * D * 2);
    PF_HIP(hipMemsetAsync(ts_up_.w, 0, (size_t)round_up(up * D, 128) * D * 2, stream_));
    launch_f32_to_f16(stream_, tmp, up * D, D, D, ts_up_.w, D);
    float* ubias = (float*)dalloc(rb.size() * 4);
    PF_HIP(hipMemcpyAsync(ubias, rb.data(), rb.size() * 4, hipMemcpyHostToDevice, stream_));
    ts_up_.bias = ubias;
    PF_HIP(hipStreamSynchronize(stream_));

    ts_ih_.N = 8 * D; ts_ih_.K = D; ts_ih_.Kpad = D;
    ts_ih_.w = (half_t*)dalloc((size_t)8 * D * D * 2);
    ts_whh_ = (half_t*)dalloc((size_t)8 * D * D * 2);
    std::vector<float> bsum((size_t)8 * D), b1((size_t)4 * D), b2((size_t)4 * D);
    const char* sfx[2] = {"", "_reverse"};
    for (int d = 0; d < 2; ++d) {
      const Tensor& wih = tensor(std::string("predictor.blstm.weight_ih") + sfx[d]);
      const Tensor& whh = tensor(std::string("predictor.blstm.weight_hh") + sfx[d]);
      const Tensor& bih = tensor(std::string("predictor.blstm.bias_ih") + sfx[d]);
      const Tensor& bhh = tensor(std::string("predictor.blstm.bias_hh") + sfx[d]);
      PF_CHECK(wih.numel == (int64_t)4 * D * D && whh.numel == (int64_t)4 * D * D && bih.numel == 4 * D && bhh.numel == 4 * D,
               PF_ERR_FORMAT, "weights: predictor.blstm shapes");
      launch_f32_to_f16(stream_, wih.dev, 4 * D, D, D, ts_ih_.w + (size_t)d * 4 * D * D, D);
      launch_f32_to_f16(stream_, whh.dev, 4 * D, D, D, ts_whh_ + (size_t)d * 4 * D * D, D);
      PF_HIP(hipMemcpy(b1.data(), bih.dev, (size_t)4 * D * 4, hipMemcpyDeviceToHost));
      PF_HIP(hipMemcpy(b2.data(), bhh.dev, (size_t)4 * D * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < 4 * D; ++i) bsum[(size_t)d * 4 * D + i] = b1[i] + b2[i];
    }
    float* gb = (float*)dalloc(bsum.size() * 4);
    PF_HIP(hipMemcpyAsync(gb, bsum.data(), bsum.size() * 4, hipMemcpyHostToDevice, stream_));
    ts_ih_.bias = gb;
    const Tensor& ow = tensor("predictor.out2.weight");
    PF_CHECK(ow.numel == 2 * D, PF_ERR_FORMAT, "weights: predictor.out2.weight shape");
    PF_CHECK(tensor("predictor.out2.bias").numel == 1, PF_ERR_FORMAT, "weights: predictor.out2.bias shape");
    ts_out_w_ = ow.dev;
    ts_out_b_ = tensor("predictor.out2.bias").dev;
    PF_HIP(hipStreamSynchronize(stream_));
    if (fp32_mode_) { ts_up_w32_ = tmp; owned_.push_back(tmp); }   // [(j, out)][in] fp32, the operand of the parity mode
    else PF_HIP(hipFree(tmp));
  }
  // decoder
  const int nd = mc_.dec_layers;
  if (nd > 0) {
    dec_kv_all_.N = nd * 2 * D; dec_kv_all_.K = D; dec_kv_all_.Kpad = D;
    dec_kv_all_.w = (half_t*)dalloc((size_t)round_up(dec_kv_all_.N, 128) * D * 2);
    PF_HIP(hipMemsetAsync(dec_kv_all_.w, 0, (size_t)round_up(dec_kv_all_.N, 128) * D * 2, stream_));
    float* kvb = (float*)dalloc((size_t)dec_kv_all_.N * 4);
    dec_kv_all_.bias = kvb;
    for (int i = 0; i < nd; ++i) {
      const std::string p = "decoder.layers." + std::to_string(i);
      DecLayer L;
      L.norm1 = make_ln(p + ".norm1", D);
      L.w1 = make_lin(p + ".ffn.w1", true);
      L.ffn_norm = make_ln(p + ".ffn.norm", L.w1.N);
      L.w2 = make_lin(p + ".ffn.w2", false);
      L.norm2 = make_ln(p + ".norm2", D);
      L.fsmn_wT = make_fsmn_wT(p + ".fsmn.weight");
      L.norm3 = make_ln(p + ".norm3", D);
      L.q = make_lin(p + ".src.q", true);
      L.out = make_lin(p + ".src.out", true);
      PF_CHECK(L.w1.K == D && L.w1.N == mc_.ffn && L.w2.N == D && L.w2.K == L.w1.N && L.q.N == D && L.q.K == D &&
                   L.out.N == D && L.out.K == D,
               PF_ERR_FORMAT, "weights: decoder layer shape mismatch in " + p);
      const Tensor& kvw = tensor(p + ".src.kv.weight");
      const Tensor& kvbias = tensor(p + ".src.kv.bias");
      PF_CHECK(kvw.numel == (int64_t)2 * D * D && kvbias.numel == 2 * D, PF_ERR_FORMAT, "weights: src.kv shape in " + p);
      launch_f32_to_f16(stream_, kvw.dev, 2 * D, D, D, dec_kv_all_.w + (size_t)i * 2 * D * D, D);
      PF_HIP(hipMemcpyAsync(kvb + (size_t)i * 2 * D, kvbias.dev, 2 * D * 4, hipMemcpyDeviceToDevice, stream_));
      L.kv32.w32 = kvw.dev; L.kv32.bias = kvbias.dev; L.kv32.N = 2 * D; L.kv32.K = D;
      L.kv32.w = dec_kv_all_.w + (size_t)i * 2 * D * D; L.kv32.Kpad = D;
      L.ffn_img = make_dec_ffn_image(L.w1, L.ffn_norm, L.w2);
      if (L.ffn_img && dec_out_chain_ && L.out.w && L.out.bias && L.out.Kpad == D) {
        L.out_wt = (half_t*)dalloc(ffn_outproj_weight_bytes());
        launch_ffn_retile_out(stream_, L.out.w, L.out.Kpad, L.out_wt);
      }
      if (L.ffn_img && dec_mid_ && L.q.w && L.q.bias && L.q.Kpad == D && L.q.N == D) {
        L.q_wt = (half_t*)dalloc(ffn_outproj_weight_bytes());
        launch_ffn_retile_out(stream_, L.q.w, L.q.Kpad, L.q_wt);
      }
      dec_.push_back(L);
    }
  }
  dec_final_norm1_ = make_ln("decoder.final.norm1", D);
  dec_final_w1_ = make_lin("decoder.final.ffn.w1", true);
  dec_final_ffn_norm_ = make_ln("decoder.final.ffn.norm", dec_final_w1_.N);
  dec_final_w2_ = make_lin("decoder.final.ffn.w2", false);
  dec_after_ = make_ln("decoder.after_norm", D);
  dec_final_img_ = make_dec_ffn_image(dec_final_w1_, dec_final_ffn_norm_, dec_final_w2_);
  dec_out_ = make_lin("decoder.output", true);
  PF_CHECK(dec_out_.N == mc_.vocab && dec_out_.K == D && dec_final_w1_.K == D && dec_final_w1_.N == mc_.ffn &&
               dec_final_w2_.N == D && dec_final_w2_.K == dec_final_w1_.N,
           PF_ERR_FORMAT, "weights: decoder.final / decoder.output shapes");
  // SeACo: hotword embedder (Embedding + LSTM stack) and the bias decoder
  if (mc_.seaco) {
    const int ns = mc_.seaco_layers;
    const Tensor& ew = tensor("seaco.embed.weight");
    PF_CHECK(ew.shape.size() == 2 && ew.shape[1] == D, PF_ERR_FORMAT, "weights: seaco.embed.weight shape");
    seaco_embed_w_ = ew.dev;
    std::vector<float> b1((size_t)4 * D), b2((size_t)4 * D);
    for (int l = 0; l < mc_.seaco_lstm_layers; ++l) {
      const std::string p = "seaco.lstm.l" + std::to_string(l);
      const Tensor& wih = tensor(p + ".weight_ih");
      const Tensor& whh = tensor(p + ".weight_hh");
      const Tensor& bih = tensor(p + ".bias_ih");
      const Tensor& bhh = tensor(p + ".bias_hh");
      PF_CHECK(wih.numel == (int64_t)4 * D * D && whh.numel == (int64_t)4 * D * D && bih.numel == 4 * D && bhh.numel == 4 * D,
               PF_ERR_FORMAT, "weights: seaco.lstm shapes");
      LstmLayer L;
      L.ih.N = 4 * D; L.ih.K = D; L.ih.Kpad = D;
      L.ih.w = (half_t*)dalloc((size_t)4 * D * D * 2);
      L.whh = (half_t*)dalloc((size_t)4 * D * D * 2);
      launch_f32_to_f16(stream_, wih.dev, 4 * D, D, D, L.ih.w, D);
      launch_f32_to_f16(stream_, whh.dev, 4 * D, D, D, L.whh, D);
      PF_HIP(hipMemcpy(b1.data(), bih.dev, (size_t)4 * D * 4, hipMemcpyDeviceToHost));
      PF_HIP(hipMemcpy(b2.data(), bhh.dev, (size_t)4 * D * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < 4 * D; ++i) b1[i] += b2[i];
      float* gb = (float*)dalloc((size_t)4 * D * 4);
      PF_HIP(hipMemcpy(gb, b1.data(), (size_t)4 * D * 4, hipMemcpyHostToDevice));
      L.ih.bias = gb;
      seaco_lstm_.push_back(L);
    }
    if (ns > 0) {
      seaco_kv_all_.N = ns * 2 * D; seaco_kv_all_.K = D; seaco_kv_all_.Kpad = D;
      seaco_kv_all_.w = (half_t*)dalloc((size_t)round_up(seaco_kv_all_.N, 128) * D * 2);
      PF_HIP(hipMemsetAsync(seaco_kv_all_.w, 0, (size_t)round_up(seaco_kv_all_.N, 128) * D * 2, stream_));
      float* kvb = (float*)dalloc((size_t)seaco_kv_all_.N * 4);
      seaco_kv_all_.bias = kvb;
      for (int i = 0; i < ns; ++i) {
        const std::string p = "seaco.decoder.layers." + std::to_string(i);
        DecLayer L;
        L.norm1 = make_ln(p + ".norm1", D);
        L.w1 = make_lin(p + ".ffn.w1", true);
        L.ffn_norm = make_ln(p + ".ffn.norm", L.w1.N);
        L.w2 = make_lin(p + ".ffn.w2", false);
        L.norm2 = make_ln(p + ".norm2", D);
        L.fsmn_wT = make_fsmn_wT(p + ".fsmn.weight", mc_.seaco_kernel);
        L.norm3 = make_ln(p + ".norm3", D);
        L.q = make_lin(p + ".src.q", true);
        L.out = make_lin(p + ".src.out", true);
        PF_CHECK(L.w1.K == D && L.w2.N == D && L.w2.K == L.w1.N && L.q.N == D && L.q.K == D && L.out.N == D && L.out.K == D,
                 PF_ERR_FORMAT, "weights: seaco decoder layer shape mismatch in " + p);
        const Tensor& kvw = tensor(p + ".src.kv.weight");
        const Tensor& kvbias = tensor(p + ".src.kv.bias");
        PF_CHECK(kvw.numel == (int64_t)2 * D * D && kvbias.numel == 2 * D, PF_ERR_FORMAT, "weights: src.kv shape in " + p);
        PF_CHECK(L.w1.N == mc_.seaco_ffn, PF_ERR_FORMAT, "weights: seaco ffn width != seaco_ffn");
        launch_f32_to_f16(stream_, kvw.dev, 2 * D, D, D, seaco_kv_all_.w + (size_t)i * 2 * D * D, D);
        PF_HIP(hipMemcpyAsync(kvb + (size_t)i * 2 * D, kvbias.dev, 2 * D * 4, hipMemcpyDeviceToDevice, stream_));
        L.kv32.w32 = kvw.dev; L.kv32.bias = kvbias.dev; L.kv32.N = 2 * D; L.kv32.K = D;
        L.kv32.w = seaco_kv_all_.w + (size_t)i * 2 * D * D; L.kv32.Kpad = D;
        sdec_.push_back(L);
      }
    }
    seaco_final_norm1_ = make_ln("seaco.decoder.final.norm1", D);
    seaco_final_w1_ = make_lin("seaco.decoder.final.ffn.w1", true);
    seaco_final_ffn_norm_ = make_ln("seaco.decoder.final.ffn.norm", seaco_final_w1_.N);
    seaco_final_w2_ = make_lin("seaco.decoder.final.ffn.w2", false);
    seaco_after_ = make_ln("seaco.decoder.after_norm", D);
    seaco_out_ = make_lin("seaco.output", true);
    PF_CHECK(seaco_out_.N == mc_.vocab, PF_ERR_FORMAT, "weights: seaco.output rows != vocab");
  }
  PF_HIP(hipStreamSynchronize(stream_));
}

// The decoder's FFN block for the split form of ffn_fused_kernel (k_ffn.hip): W1, gamma_F (.) W2 (from the fp32 tensor, one
// rounding), b1, colsum(gamma_F (.) W2), W2 beta_F.  Null when the form does not apply (mode, shape).
half_t* Engine::make_dec_ffn_image(const Lin& w1, const LNp& fn, const Lin& w2) {
  if (!dec_ffn_fused_ || !ffn_fused_ || fp32_mode_ || int8_mode_ || !ffn_fused_applicable(w1.K, w1.N) || w2.N != w1.K || w2.K != w1.N ||
      !w1.w || !w1.bias || !w2.w32 || w2.bias || w1.Kpad != w1.K || fn.D != w1.N)
    return nullptr;
  half_t* img = (half_t*)dalloc(ffn_dec_image_bytes());
  launch_ffn_dec_retile(stream_, w1.w, w1.Kpad, w2.w32, fn.g, fn.b, w1.bias, img);
  return img;
}

Lin Engine::make_lin(const std::string& prefix, bool bias) {
  const Tensor& w = tensor(prefix + ".weight");
  PF_CHECK(w.shape.size() == 2, PF_ERR_FORMAT, "weights: '" + prefix + ".weight' must be 2-D [out,in]");
  Lin L;
  L.N = (int)w.shape[0];
  L.K = (int)w.shape[1];
  L.Kpad = (int)round_up(L.K, 64);
  const size_t npad = (size_t)round_up(L.N, 128);
  L.w = (half_t*)dalloc(npad * L.Kpad * 2);
  PF_HIP(hipMemsetAsync(L.w, 0, npad * L.Kpad * 2, stream_));
  launch_f32_to_f16(stream_, w.dev, L.N, L.K, L.K, L.w, L.Kpad);
  L.w32 = w.dev;
  if (bias) {
    const Tensor& b = tensor(prefix + ".bias");
    PF_CHECK(b.numel == L.N, PF_ERR_FORMAT, "weights: bias length mismatch for " + prefix);
    L.bias = b.dev;
  }
  return L;
}

LNp Engine::make_ln(const std::string& prefix, int width) {
  LNp p;
  const Tensor& g = tensor(prefix + ".weight");
  const Tensor& b = tensor(prefix + ".bias");
  PF_CHECK(g.numel == b.numel, PF_ERR_FORMAT, "weights: LayerNorm size mismatch for " + prefix);
  // the kernel normalises rows of `width` floats: a shorter gamma / beta would be read out of bounds
  PF_CHECK(g.numel == width, PF_ERR_FORMAT, "weights: '" + prefix + "' has " + std::to_string(g.numel) +
                                                " elements, the layer is " + std::to_string(width) + " wide");
  p.g = g.dev; p.b = b.dev; p.D = (int)g.numel;
  return p;
}

float* Engine::make_fsmn_wT(const std::string& name, int Kopt) {
  const Tensor& w = tensor(name);
  const int D = mc_.d_model, K = Kopt > 0 ? Kopt : mc_.kernel;
  PF_CHECK(w.shape.size() == 2 && w.shape[0] == D && w.shape[1] == K, PF_ERR_FORMAT,
           "weights: '" + name + "' must be [d_model, kernel]");
  std::vector<float> h(w.numel), t(w.numel);
  PF_HIP(hipMemcpy(h.data(), w.dev, w.numel * 4, hipMemcpyDeviceToHost));
  for (int c = 0; c < D; ++c)
    for (int jx = 0; jx < K; ++jx) t[(size_t)jx * D + c] = h[(size_t)c * K + jx];
  float* d = (float*)dalloc(w.numel * 4);
  PF_HIP(hipMemcpy(d, t.data(), w.numel * 4, hipMemcpyHostToDevice));
  return d;
}

// ------------------------------------------------------------------ profiling -------------
void Engine::profile_reset() {
  for (auto& kv : prof_)
    for (auto& pr : kv.second.ev) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  prof_.clear();
}
void Engine::prof_begin(const char* cls, double flops) {
  if (!prof_on_) return;
  if (!prof_only_.empty() && prof_only_ != cls) return;
  ProfClass& pc = prof_[cls];
  hipEvent_t a, b;
  PF_HIP(hipEventCreate(&a));
  PF_HIP(hipEventCreate(&b));
  pc.ev.emplace_back(a, b);
  pc.flops += flops;
  pc.n += 1;
  PF_HIP(hipEventRecord(a, stream_));
}
void Engine::prof_end(const char* cls) {
  if (!prof_on_) return;
  if (!prof_only_.empty() && prof_only_ != cls) return;
  ProfClass& pc = prof_[cls];
  PF_HIP(hipEventRecord(pc.ev.back().second, stream_));
  if (cls[0] == 'g' && cls[1] == 'e') pc.kernel = last_gemm_kernel();   // "gemm_*" classes
}
std::string Engine::profile_kernel(const std::string& cls) const {
  auto it = prof_.find(cls);
  return it == prof_.end() ? std::string() : it->second.kernel;
}
bool Engine::profile_get(const std::string& cls, double* ms, int64_t* launches, double* flops_per_launch) {
  auto it = prof_.find(cls);
  if (it == prof_.end()) return false;
  PF_HIP(hipStreamSynchronize(stream_));
  double total = 0;
  for (auto& pr : it->second.ev) {
    float t = 0;
    PF_HIP(hipEventElapsedTime(&t, pr.first, pr.second));
    total += t;
  }
  if (ms) *ms = total;
  if (launches) *launches = it->second.n;
  if (flops_per_launch) *flops_per_launch = it->second.n ? it->second.flops / it->second.n : 0;
  return true;
}

void Engine::gemm(const char* cls, const Lin& w, const half_t* A, int lda, int M, float* out32, int ld32,
                  half_t* out16, int ld16, const float* resid, int ldr, const float* add2, int ld2, bool relu,
                  int scale_cols, float scale, bool bias, int blocked) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = w.w; g.ldw = w.Kpad; g.bias = bias ? w.bias : nullptr;
  g.M = M; g.N = w.N; g.K = w.Kpad;
  g.out_f32 = out32; g.ldc32 = ld32; g.out_f16 = out16; g.ldc16 = ld16;
  g.resid = resid; g.ldr = ldr; g.add2 = add2; g.ld2 = ld2;
  g.relu = relu ? 1 : 0; g.scale_cols = scale_cols; g.scale = scale;
  g.out_padded = 1;   // every pipeline buffer is carved with round_up(rows,128)+128 rows
  g.out_blocked = blocked == 1; g.a_blocked = blocked == 2;
  g.small_ws = small_ws_;
  prof_begin(cls, 2.0 * M * (double)w.N * w.K);
  launch_gemm(stream_, g);
  prof_end(cls);
}

// short-input form of a projection (k_gemm_small.hip): the caller fills operand / epilogue fields of `g`, this adds
// the weight, shape, scratch and the profile class
void Engine::gemm_small_call(const char* cls, const Lin& w, GemmSmallArgs g, bool bias) {
  g.W = w.w; g.ldw = w.Kpad; g.bias = bias ? w.bias : nullptr; g.N = w.N; g.K = w.Kpad; g.ws = small_ws_;
  prof_begin(cls, 2.0 * g.M * (double)w.N * w.K);
  launch_gemm_small(stream_, g);
  prof_end(cls);
}

// ------------------------------------------------------------------ front-end -------------
int Engine::num_fbank_frames(int64_t n) const {
  if (fc_.snip_edges) return n < 400 ? 0 : (int)(1 + (n - 400) / 160);
  return (int)((n + 80) / 160);
}
int Engine::num_lfr_frames(int64_t n) const {
  const int t80 = num_fbank_frames(n);
  if (fc_.lfr_m == 1 && fc_.lfr_n == 1) return t80;
  return t80 / fc_.lfr_n;
}

void Engine::stage_audio(const float* const* samples, const int64_t* n, int B, int force_T) {
  PF_CHECK(B >= 0, PF_ERR_INVALID_ARG, "negative batch");
  PF_HIP(hipSetDevice(device_));
  st_audio_ext_ = nullptr;
  st_B_ = B;
  st_n_.assign(n, n + B);
  st_t80_.resize(B);
  std::vector<int64_t> meta(3 * (size_t)(B + 1), 0);   // audio_off | n_samples | frame_off
  int64_t tot = 0, frames = 0;
  int tmax = 0;
  for (int b = 0; b < B; ++b) {
    if (!samples[b] && n[b] > 0) throw Error(PF_ERR_NULL_SAMPLES, "source");
    PF_CHECK(n[b] >= 0, PF_ERR_INVALID_ARG, "negative sample count");
    meta[b] = tot;
    meta[(B + 1) + b] = n[b];
    meta[2 * (B + 1) + b] = frames;
    st_t80_[b] = num_fbank_frames(n[b]);
    tot += round_up(n[b], 4);
    frames += st_t80_[b];
    tmax = std::max(tmax, num_lfr_frames(n[b]));
  }
  meta[2 * (B + 1) + B] = frames;
  st_total_frames_ = frames;
  st_T_ = std::max(tmax, force_T);
  ensure(ws_audio_, (size_t)std::max<int64_t>(tot, 1) * 4);
  ensure(ws_meta_, meta.size() * 8 + (size_t)B * 4 + 64);
  for (int b = 0; b < B; ++b)
    if (n[b] > 0)
      PF_HIP(hipMemcpyAsync((float*)ws_audio_.p + meta[b], samples[b], n[b] * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(ws_meta_.p, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync((char*)ws_meta_.p + meta.size() * 8, st_t80_.data(), (size_t)B * 4, hipMemcpyHostToDevice,
                        stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::stage_device_audio(const float* const* samples_dev, const int64_t* n, int B, int force_T) {
  PF_CHECK(B >= 0, PF_ERR_INVALID_ARG, "negative batch");
  PF_HIP(hipSetDevice(device_));
  st_B_ = B;
  st_n_.assign(n, n + B);
  st_t80_.resize(B);
  // offsets are relative to the LOWEST of the buffers (element offsets, non-negative): the fbank kernel adds them to one base
  uintptr_t lo = UINTPTR_MAX;
  for (int b = 0; b < B; ++b) {
    if (!samples_dev[b] && n[b] > 0) throw Error(PF_ERR_NULL_SAMPLES, "source");
    PF_CHECK(n[b] >= 0 && ((uintptr_t)samples_dev[b] & 3) == 0, PF_ERR_INVALID_ARG, "device audio must be 4-byte aligned");
    if (n[b] > 0) lo = std::min(lo, (uintptr_t)samples_dev[b]);
  }
  if (lo == UINTPTR_MAX) lo = 0;
  std::vector<int64_t> meta(3 * (size_t)(B + 1), 0);   // audio_off | n_samples | frame_off
  int64_t frames = 0;
  int tmax = 0;
  for (int b = 0; b < B; ++b) {
    meta[b] = n[b] > 0 ? (int64_t)(((uintptr_t)samples_dev[b] - lo) / 4) : 0;
    meta[(B + 1) + b] = n[b];
    meta[2 * (B + 1) + b] = frames;
    st_t80_[b] = num_fbank_frames(n[b]);
    frames += st_t80_[b];
    tmax = std::max(tmax, num_lfr_frames(n[b]));
  }
  meta[2 * (B + 1) + B] = frames;
  st_total_frames_ = frames;
  st_T_ = std::max(tmax, force_T);
  st_audio_ext_ = (const float*)lo;
  ensure(ws_meta_, meta.size() * 8 + (size_t)B * 4 + 64);
  PF_HIP(hipMemcpyAsync(ws_meta_.p, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync((char*)ws_meta_.p + meta.size() * 8, st_t80_.data(), (size_t)B * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipStreamSynchronize(stream_));               // (meta is a stack vector; 1 KB)
}

void Engine::run_staged(bool want_logits) {
  PF_HIP(hipSetDevice(device_));
  const int B = st_B_, T = st_T_;
  if (B == 0) { last_ = HostBatchOut(); return; }
  PF_CHECK(T > 0, PF_ERR_INVALID_ARG, "audio too short: no LFR frame");
  const int W = fc_.lfr_m * fc_.n_mels;
  PF_CHECK(W == mc_.feat_dim, PF_ERR_INVALID_ARG, "lfr_m * n_mels != model feature dim");
  const int64_t* meta = (const int64_t*)ws_meta_.p;
  const int32_t* t80d = (const int32_t*)((const char*)ws_meta_.p + 3 * (size_t)(B + 1) * 8);
  ensure(ws_fbank_, (size_t)std::max<int64_t>(st_total_frames_, 1) * fc_.n_mels * 4);
  const int P = sv_prompt_ ? 4 : 0;                 // SenseVoice: query rows prepended on the device
  ensure(ws_speech_, (size_t)B * (T + P) * W * 4);
  prof_begin("fbank", 0);
  launch_fbank(stream_, fb_, st_audio_ext_ ? st_audio_ext_ : (const float*)ws_audio_.p, meta, meta + (B + 1), meta + 2 * (B + 1), B,
               st_total_frames_, fc_.snip_edges ? 1 : 0, (float*)ws_fbank_.p, fc_.dither, next_dither_seed());
  prof_end("fbank");
  prof_begin("lfr_cmvn_pad", 0);
  launch_lfr_cmvn_pad(stream_, (const float*)ws_fbank_.p, meta + 2 * (B + 1), t80d, B, T, fc_.lfr_m, fc_.lfr_n,
                      fc_.n_mels, cmvn_shift_, cmvn_scale_, cmvn_shift_ ? 1 : 0, 1, (float*)ws_speech_.p, sv_prompt_, P);
  prof_end("lfr_cmvn_pad");
  forward_device((const float*)ws_speech_.p, B, T + P, want_logits);
}

void Engine::fbank_host(const float* samples, int64_t n, std::vector<float>& out, int& t80) {
  if (!samples) throw Error(PF_ERR_NULL_SAMPLES, "source");
  const float* arr[1] = {samples};
  const int64_t nn[1] = {n};
  stage_audio(arr, nn, 1);
  t80 = st_t80_[0];
  out.assign((size_t)t80 * fc_.n_mels, 0.f);
  if (t80 == 0) return;
  const int64_t* meta = (const int64_t*)ws_meta_.p;
  ensure(ws_fbank_, (size_t)t80 * fc_.n_mels * 4);
  launch_fbank(stream_, fb_, (const float*)ws_audio_.p, meta, meta + 2, meta + 4, 1, t80, fc_.snip_edges ? 1 : 0,
               (float*)ws_fbank_.p, fc_.dither, next_dither_seed());
  PF_HIP(hipMemcpyAsync(out.data(), ws_fbank_.p, out.size() * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::frontend_host(const float* samples, int64_t n, std::vector<float>& feats, int& t_lfr) {
  if (!samples) throw Error(PF_ERR_NULL_SAMPLES, "source");
  const float* arr[1] = {samples};
  const int64_t nn[1] = {n};
  stage_audio(arr, nn, 1);
  frontend_staged_one(feats, t_lfr);
}

void Engine::frontend_from_device(const float* samples_dev, int64_t n, std::vector<float>& feats, int& t_lfr) {
  const float* arr[1] = {samples_dev};
  const int64_t nn[1] = {n};
  stage_device_audio(arr, nn, 1);
  frontend_staged_one(feats, t_lfr);
}

// fbank + LFR + CMVN of the ONE staged utterance -> host features (no padding, no sentinel)
void Engine::frontend_staged_one(std::vector<float>& feats, int& t_lfr) {
  const int t80 = st_t80_[0];
  const bool lfr = fc_.lfr_m != 1 || fc_.lfr_n != 1;
  const int m = lfr ? fc_.lfr_m : 1, nn_ = lfr ? fc_.lfr_n : 1;
  t_lfr = t80 / nn_;
  const int W = m * fc_.n_mels;
  feats.assign((size_t)t_lfr * W, 0.f);
  if (t_lfr == 0) return;
  const int64_t* meta = (const int64_t*)ws_meta_.p;
  const int32_t* t80d = (const int32_t*)((const char*)ws_meta_.p + 3 * 2 * 8);
  ensure(ws_fbank_, (size_t)t80 * fc_.n_mels * 4);
  ensure(ws_speech_, feats.size() * 4);
  launch_fbank(stream_, fb_, st_audio_ext_ ? st_audio_ext_ : (const float*)ws_audio_.p, meta, meta + 2, meta + 4, 1, t80, fc_.snip_edges ? 1 : 0,
               (float*)ws_fbank_.p, fc_.dither, next_dither_seed());
  const bool cm = cmvn_shift_ && cmvn_dim_ == W;
  launch_lfr_cmvn_pad(stream_, (const float*)ws_fbank_.p, meta + 4, t80d, 1, t_lfr, m, nn_, fc_.n_mels, cmvn_shift_,
                      cmvn_scale_, cm ? 1 : 0, 0, (float*)ws_speech_.p);
  PF_HIP(hipMemcpyAsync(feats.data(), ws_speech_.p, feats.size() * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

// ------------------------------------------------------------------ encoder ---------------
void Engine::build_pe(int T) {
  if (T <= pe_T_) return;
  const int F = mc_.feat_dim, half = F / 2;
  const int Tn = (int)round_up(T, 256);
  std::vector<float> pe((size_t)Tn * F);
  // SinusoidalPositionEncoder.encode in float32: inv_timescales = exp(i * -(ln 1e4 / (half-1)))
  const float inc = (float)(std::log(10000.0) / (double)(half - 1));
  std::vector<float> inv(half);
  for (int i = 0; i < half; ++i) inv[i] = std::exp((float)i * (-inc));
  for (int t = 0; t < Tn; ++t) {
    const float pos = (float)(t + 1);
    for (int i = 0; i < half; ++i) {
      const float st = pos * inv[i];
      pe[(size_t)t * F + i] = (float)std::sin((double)st);
      pe[(size_t)t * F + half + i] = (float)std::cos((double)st);
    }
  }
  ensure(ws_pe_, pe.size() * 4);
  PF_HIP(hipMemcpyAsync(ws_pe_.p, pe.data(), pe.size() * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
  pe_T_ = Tn;
}

// One SAN-M encoder layer = 5 launches: [LayerNorm fused into the previous launch] QKV GEMM -> attention ->
// row-complete out-projection (+ bias + residual + FSMN memory from the V slice + LayerNorm norm2 -> xn16) ->
// FFN-up (blocked hidden) -> row-complete FFN-down (+ bias + residual + the NEXT LayerNorm `nx` -> its outputs).
// Entry: xn16_ holds LayerNorm norm1 of the residual stream (written by the previous layer's FFN-down, by the
// position-encoding kernel for layer 0, or by the stand-alone LayerNorm for the first tp layer).
void Engine::enc_layer(const EncLayer& L, int first, const float* speech_dev, int B, int T, const EncNext& nx) {
  const int D = mc_.d_model, M = B * T, F = mc_.ffn;
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));
  const int lda = first ? L.qkv.Kpad : D;
  if (first == 1) {
    prof_begin("layernorm", 0);
    launch_posenc_ln_tab(stream_, speech_dev, B, T, mc_.feat_dim, std::sqrt((float)D), (const float*)ws_pe_.p,
                         L.norm1.g, L.norm1.b, xn16_, lda);
    prof_end("layernorm");
  } else if (first == 2) {
    // streaming path: the caller scaled and position-encoded the chunk already (OnlineStream.cs GetDecodeChunk)
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, speech_dev, M, mc_.feat_dim, L.norm1.g, L.norm1.b, xn16_, lda, nullptr, 0);
    prof_end("layernorm");
  }
  AttnArgs a{};
  a.q = qkv16_; a.k = qkv16_ + D; a.v = qkv16_ + 2 * D; a.o = ctx16_;
  a.q_bstride = a.k_bstride = a.v_bstride = (int64_t)T * 3 * D;
  a.q_rstride = a.k_rstride = a.v_rstride = 3 * D;
  a.o_bstride = (int64_t)T * D; a.o_rstride = D;
  a.B = B; a.H = mc_.heads; a.Lq = T; a.Lk = T;
  // short inputs (M <= 512 rows): every GEMM goes to k_gemm_small.hip, which has no row-complete epilogue and no blocked
  // layout but takes the FSMN memory as an epilogue term and the LayerNorm behind FFN-down in its reduction
  const bool small = M <= gemm_small_max_rows() && small_ws_;
  // the split FFN-down form must really apply to (M, F) — e.g. F = 4096 needs 16 splits, whose partials outgrow the
  // scratch beyond 256 rows: then the regular kernels run instead of failing the utterance
  auto split_ok = [&](int rows, const Lin& w2, const half_t* Aop) {
    GemmSmallArgs t{};
    t.M = rows; t.N = w2.N; t.K = w2.Kpad; t.A = Aop; t.lda = w2.Kpad; t.W = w2.w; t.ldw = w2.Kpad; t.ws = small_ws_;
    t.post_ln_g = nx.ln.g; t.post_ln_b = nx.ln.b;
    return gemm_small_applicable(t);
  };
  if (small && D == 512 && F % 64 == 0 && F > 576 && mc_.kernel == 11 && !no_small_fuse_ && split_ok(M, L.w2, h16_)) {
    // 7 launches: QKV | attention | out-projection + FSMN memory + residual | norm2 | FFN-up | FFN-down partials | their
    // sum + bias + residual + the LayerNorm behind the block (next norm1 / after_norm)
    gemm("gemm_qkv", L.qkv, xn16_, lda, M, nullptr, 0, qkv16_, 3 * D, nullptr, 0, nullptr, 0, false, D, qscale);
    prof_begin("attn_self", 4.0 * B * (double)T * T * D);
    launch_attention(stream_, a);
    prof_end("attn_self");
    GemmSmallArgs o{};
    o.M = M; o.A = ctx16_; o.lda = D; o.out_f32 = x_; o.ldc32 = D;
    if (!first) { o.resid = x_; o.ldr = D; }
    o.fsmn_v = qkv16_ + 2 * D; o.ldv = 3 * D; o.fsmn_wT = L.fsmn_wT; o.fsmn_k = 11; o.T = T;
    gemm_small_call("gemm_out", L.out, o);
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, x_, M, D, L.norm2.g, L.norm2.b, xn16_, D, nullptr, 0);
    prof_end("layernorm");
    gemm("gemm_ffn1", L.w1, xn16_, D, M, nullptr, 0, h16_, F, nullptr, 0, nullptr, 0, true, 0, 1.f);
    GemmSmallArgs dn{};
    dn.M = M; dn.A = h16_; dn.lda = F; dn.resid = x_; dn.ldr = D;
    if (nx.keep_x) { dn.out_f32 = x_; dn.ldc32 = D; }
    dn.post_ln_g = nx.ln.g; dn.post_ln_b = nx.ln.b; dn.post_n16 = nx.n16; dn.ldn16 = D; dn.post_n32 = nx.n32; dn.ldn32 = D;
    gemm_small_call("gemm_ffn2", L.w2, dn);
    return;
  }
  const bool rc = mc_.kernel == 11 && T >= 8 && !no_rc_ && !small;
  // Q | K blocked + V row-major from the persistent 256 x 192 kernel when the row-complete out-projection (the reader of
  // the row-major V) runs and the launch has at least a round of tiles
  const half_t* v16 = qkv16_ + 2 * D;
  int ldv = 3 * D;
  // (chosen by idle rounds: 32 x 500 rows are 504 tiles = 1.97 rounds on 256 CUs — 1.88 vs 2.06 ms per step for the 50 layers;
  // SenseVoice's 10 944 rows are 344 tiles = 1.34 rounds: 2.45 vs 2.30 ms alone, but 10.19 vs 10.73 ms per step with two steps in
  // flight — the second engine's kernels take the CUs its short second round leaves — so the fill bar is 60 %)
  const int qt = 8 * cdiv(M, 256);
  const bool split = rc && L.qkv_p && qt >= qkv_split_min_tiles_ && qt * 100 >= qkv_split_min_fill_ * (int)round_up(qt, cus_) &&
                     gemm_qkvp_applicable(M, L.qkv.Kpad, lda, L.qkv.Kpad, D);
  const bool pre = qkv_done_;                                  // written by the previous layer's launch (its Q | K | V tail)
  qkv_done_ = false;
  PF_CHECK(!pre || split, PF_ERR_DEVICE, "encoder: a projected Q | K | V without its consumer");
  const int64_t Mp = round_up((int64_t)M, 128) + 128;          // rows of every encoder buffer (>= round_up(M, 256))
  // two V buffers: the fused launch of layer l reads V_l (FSMN window: rows of NEIGHBOURING tiles too) and writes V_{l+1}
  half_t* const vbufs[2] = {qkv16_ + (size_t)Mp * 2 * D, h16_};
  if (split) {
    half_t* vbuf = vbufs[v_pp_];
    if (!pre) {
      prof_begin("gemm_qkv", 2.0 * M * (double)L.qkv.N * L.qkv.K);
      launch_gemm_qkvp(stream_, xn16_, lda, L.qkv_p, L.qkv.Kpad, L.qkv_bias_p, M, L.qkv.Kpad, qscale, qkv16_, vbuf, D);
      prof_end("gemm_qkv");
    }
    a.q = qkv16_; a.k = qkv16_; a.qk_blocked = 1; a.blk_groups = 2 * D / 8; a.blk_brows = T; a.blk_kgrp = D / 8;
    a.v = vbuf; a.v_bstride = (int64_t)T * D; a.v_rstride = D;
    v16 = vbuf; ldv = D;
  } else {
    gemm("gemm_qkv", L.qkv, xn16_, lda, M, nullptr, 0, qkv16_, 3 * D, nullptr, 0, nullptr, 0, false, D, qscale);
  }
  prof_begin("attn_self", 4.0 * B * (double)T * T * D);
  launch_attention(stream_, a);
  prof_end("attn_self");
  if (rc && L.ffn_wt && L.out_wt && M >= ffn_fused_min_rows_) {
    // out-projection + bias + residual + FSMN memory + norm2 + the whole FFN block + the NEXT LayerNorm: ONE launch per 64-row
    // tile (k_ffn.hip, OP = 1): norm2's result never leaves LDS, x_mid goes through a scratch the same lanes read back
    FfnFusedArgs f{};
    f.ctx = ctx16_; f.lda_c = D; f.Wot = L.out_wt; f.bo = L.out.bias; f.fsmn_v = v16; f.ldv = ldv; f.fsmn_wT = L.fsmn_wT; f.T = T;
    f.ln2_g = L.norm2.g; f.ln2_b = L.norm2.b;
    f.Wt = L.ffn_wt; f.b1 = L.w1.bias; f.b2 = L.w2.bias; f.M = M;
    f.resid = first ? nullptr : x_; f.ldr = D; f.out_x = nx.keep_x ? x_ : nullptr; f.ldx = D;
    f.ln_g = nx.ln.g; f.ln_b = nx.ln.b; f.eps = 1e-12f; f.out_n16 = nx.n16; f.ldn16 = D; f.out_n32 = nx.n32; f.ldn32 = D;
    // the NEXT layer's Q | K | V projection behind the block when that layer would take the same (blocked-layout) path on
    // the same rows: LayerNorm_next(x) then never visits HBM either
    const bool tail = split && nx.next && nx.next->qkv_t && nx.next->ffn_wt && nx.next->out_wt && nx.n16 == xn16_ && !nx.n32;
    double flops = 2.0 * M * (double)D * D + 4.0 * M * (double)D * F;
    if (tail) {
      f.Wqt = nx.next->qkv_t; f.bq = nx.next->qkv.bias; f.out_qk = qkv16_; f.out_v = vbufs[v_pp_ ^ 1]; f.ldvo = D; f.qscale = qscale;
      f.out_n16 = nullptr;
      flops += 2.0 * M * 3.0 * D * D;
    }
    prof_begin("gemm_outffn", flops);
    launch_ffn_fused(stream_, f);
    prof_end("gemm_outffn");
    if (tail) { qkv_done_ = true; v_pp_ ^= 1; }
    return;
  }
  if (rc) {
    GemmRcArgs g{};
    g.A = ctx16_; g.lda = D; g.W = L.out.w; g.ldw = L.out.Kpad; g.bias = L.out.bias; g.M = M; g.K = L.out.Kpad;
    g.resid = first ? nullptr : x_; g.ldr = D; g.out_x = x_; g.ldx = D;
    g.fsmn_v = v16; g.ldv = ldv; g.fsmn_wT = L.fsmn_wT; g.fsmn_k = mc_.kernel; g.T = T;
    g.ln_g = L.norm2.g; g.ln_b = L.norm2.b; g.eps = 1e-12f; g.out_n16 = xn16_; g.ldn16 = D;
    prof_begin("gemm_out", 2.0 * M * (double)D * D);
    launch_gemm_rc(stream_, g);
    prof_end("gemm_out");
  } else {
    // fallback (FSMN kernel size other than 11, utterances shorter than 8 frames, PF_NO_RC=1): the unfused sequence
    prof_begin("fsmn", 0);
    launch_fsmn_enc(stream_, qkv16_ + 2 * D, 3 * D, L.fsmn_wT, B, T, D, mc_.kernel, fsm_);
    prof_end("fsmn");
    gemm("gemm_out", L.out, ctx16_, D, M, x_, D, nullptr, 0, first ? nullptr : x_, D, fsm_, D, false, 0, 1.f);
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, x_, M, D, L.norm2.g, L.norm2.b, xn16_, D, nullptr, 0);
    prof_end("layernorm");
  }
  if (rc && L.ffn_wt && M >= ffn_fused_min_rows_) {
    // the whole feed-forward block + the NEXT LayerNorm in one launch (k_ffn.hip, round 5): the [M, 2048] hidden stays in
    // LDS.  Measured against gemm_bigp_kernel + the row-complete FFN-down: see DESIGN.md 4.1g; PF_FFN_FUSED=0 restores them
    FfnFusedArgs f{};
    f.A = xn16_; f.lda = D; f.Wt = L.ffn_wt; f.b1 = L.w1.bias; f.b2 = L.w2.bias; f.M = M;
    f.resid = x_; f.ldr = D; f.out_x = nx.keep_x ? x_ : nullptr; f.ldx = D;
    f.ln_g = nx.ln.g; f.ln_b = nx.ln.b; f.eps = 1e-12f; f.out_n16 = nx.n16; f.ldn16 = D; f.out_n32 = nx.n32; f.ldn32 = D;
    prof_begin("gemm_ffn", 4.0 * M * (double)D * F);
    launch_ffn_fused(stream_, f);
    prof_end("gemm_ffn");
    return;
  }
  // the FFN hidden lives in the blocked activation layout (kernels.h): FFN-up stores its fragments as whole
  // lines without the LDS transposition, FFN-down's LDS-DMA reads 1 KiB contiguous pieces
  const int blk = (F % 64 == 0 && !small) ? 1 : 0;
  gemm("gemm_ffn1", L.w1, xn16_, D, M, nullptr, 0, h16_, F, nullptr, 0, nullptr, 0, true, 0, 1.f, true, blk);
  if (rc && rc_ffn2_ && blk) {
    // row-complete FFN-down (+ the next LayerNorm), default since round 4: 53.6 us against 47.6 + 9-11 us for the persistent
    // 256 x 128 kernel + its LayerNorm launch (same box, same session: 12.86 -> 12.67 ms per step with one step in flight,
    // 11.32 -> 11.06 with two; round 3 had measured the opposite, 67.7 vs 57.5 us).  PF_RC_FFN2=0 restores the two launches.
    GemmRcArgs f{};
    f.A = h16_; f.lda = F; f.a_blocked = 1; f.W = L.w2.w; f.ldw = L.w2.Kpad; f.bias = L.w2.bias; f.M = M; f.K = L.w2.Kpad;
    f.resid = x_; f.ldr = D; f.out_x = nx.keep_x ? x_ : nullptr; f.ldx = D;
    f.ln_g = nx.ln.g; f.ln_b = nx.ln.b; f.eps = 1e-12f; f.out_n16 = nx.n16; f.ldn16 = D; f.out_n32 = nx.n32; f.ldn32 = D;
    prof_begin("gemm_ffn2", 2.0 * M * (double)D * F);
    launch_gemm_rc(stream_, f);
    prof_end("gemm_ffn2");
    return;
  }
  gemm("gemm_ffn2", L.w2, h16_, F, M, x_, D, nullptr, 0, x_, D, nullptr, 0, false, 0, 1.f, true, blk ? 2 : 0);
  prof_begin("layernorm", 0);
  launch_layernorm(stream_, x_, M, D, nx.ln.g, nx.ln.b, nx.n16, D, nx.n32, D);
  prof_end("layernorm");
}

void Engine::encoder(const float* speech_dev, int B, int T, bool pre_encoded) {
  const int D = mc_.d_model, F = mc_.ffn;
  const int64_t M = (int64_t)B * T;
  const int64_t Mp = round_up(M, 128) + 128;
  PF_CHECK(M < (1ll << 31) / (3 * D), PF_ERR_INVALID_ARG, "batch too large for 32-bit row indexing");
  if (!pre_encoded) build_pe(T);
  const int k0 = enc_.empty() ? D : enc_[0].qkv.Kpad;
  // carve the encoder arena
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_x = carve(Mp * D * 4), o_xn = carve(Mp * std::max(k0, D) * 2), o_qkv = carve(Mp * 3 * D * 2);
  const size_t o_ctx = carve(Mp * D * 2), o_fsm = carve(Mp * D * 4), o_h = carve(Mp * std::max(F, 3 * D) * 2);
  const size_t o_H32 = carve(Mp * D * 4), o_H16 = carve(Mp * D * 2), o_al = carve((size_t)B * (T + 1) * 4);
  const size_t o_fc = carve((size_t)B * 4), o_tn = carve((size_t)B * 4), o_ff = carve((size_t)B * (T + 1) * 4);
  const size_t o_wc = carve((size_t)B * (T + 1) * 4), o_wr = carve((size_t)B * (T + 1) * 4), o_mx = carve(256);
  ensure(ws_enc_, off);
  char* base = (char*)ws_enc_.p;
  x_ = (float*)(base + o_x); xn16_ = (half_t*)(base + o_xn); qkv16_ = (half_t*)(base + o_qkv);
  ctx16_ = (half_t*)(base + o_ctx); fsm_ = (float*)(base + o_fsm); h16_ = (half_t*)(base + o_h);
  H32_ = (float*)(base + o_H32); H16_ = (half_t*)(base + o_H16); alphas_ = (float*)(base + o_al);
  plan_.fire_count = (int32_t*)(base + o_fc); plan_.token_num = (int32_t*)(base + o_tn);
  plan_.fire_frame = (int32_t*)(base + o_ff); plan_.w_cur = (float*)(base + o_wc);
  plan_.w_rem = (float*)(base + o_wr); plan_.max_count = (int32_t*)(base + o_mx);

  // the LayerNorm that FOLLOWS layer i's FFN-down is the next layer's norm1, or after_norm behind the last one
  const bool has_tp = !tp_.empty();
  qkv_done_ = false; v_pp_ = 0;
  for (size_t i = 0; i < enc_.size(); ++i) {
    EncNext nx;
    if (i + 1 < enc_.size()) { nx.ln = enc_[i + 1].norm1; nx.n16 = xn16_; nx.keep_x = true; nx.next = &enc_[i + 1]; }
    else if (has_tp) { nx.ln = enc_after_; nx.n32 = x_; nx.keep_x = false; }     // after_norm output = the tp residual stream
    else { nx.ln = enc_after_; nx.n16 = H16_; nx.n32 = H32_; nx.keep_x = false; }
    enc_layer(enc_[i], i == 0 ? (pre_encoded ? 2 : 1) : 0, speech_dev, B, T, nx);
  }
  if (has_tp) {
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, x_, M, D, tp_[0].norm1.g, tp_[0].norm1.b, xn16_, D, nullptr, 0);
    prof_end("layernorm");
    for (size_t i = 0; i < tp_.size(); ++i) {
      EncNext nx;
      if (i + 1 < tp_.size()) { nx.ln = tp_[i + 1].norm1; nx.n16 = xn16_; nx.keep_x = true; nx.next = &tp_[i + 1]; }
      else { nx.ln = tp_norm_; nx.n16 = H16_; nx.n32 = H32_; nx.keep_x = false; }
      enc_layer(tp_[i], 0, nullptr, B, T, nx);
    }
  }
}

// ------------------------------------------------------------------ CIF + decoder ---------
// FFN-up of a decoder block and the LayerNorm(F) behind it: leaves the normalised hidden in h16.  FFN-up writes the
// hidden as f16 (ReLU output, one rounding — the oracle's 16-bit mode rounds at the same point) and the LayerNorm runs
// in place on it: 40 -> 18 us for the GEMM (no fp32 row-segment epilogue) and a third of the LayerNorm's bytes at
// M = 5344.  PF_DEC_H32=1 keeps the fp32 round trip (h32 must then be non-null).
void Engine::dec_ffn_hidden(const char* cls, const Lin& w1, const LNp& fn, const half_t* xn16, int lda, int rows, float* h32,
                            half_t* h16) {
  const int F = w1.N;
  if (dec_h32_ && h32) {
    gemm(cls, w1, xn16, lda, rows, h32, F, nullptr, 0, nullptr, 0, nullptr, 0, true, 0, 1.f);
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, h32, rows, F, fn.g, fn.b, h16, F, nullptr, 0);
    prof_end("layernorm");
  } else {
    gemm(cls, w1, xn16, lda, rows, nullptr, 0, h16, F, nullptr, 0, nullptr, 0, true, 0, 1.f);
    prof_begin("layernorm", 0);
    launch_layernorm_f16(stream_, h16, rows, F, fn.g, fn.b, h16);
    prof_end("layernorm");
  }
}

void Engine::predictor_and_decoder(int B, int T, bool want_logits) {
  const int D = mc_.d_model, F = mc_.ffn, V = mc_.vocab;
  const int M = B * T, T1 = T + 1;
  const int taps = mc_.cif_l_order + mc_.cif_r_order + 1;
  if (ev_enc_) PF_HIP(hipEventRecord(ev_enc_, stream_));     // the encoder output exists: all the timestamp head's GEMMs and its recurrence need
  // conv1d(k=3) as im2col GEMM; reuse the FFN hidden buffer for the [M, 3D] operand and the
  // FSMN buffer for the fp32 conv output.
  half_t* col16 = h16_;
  float* conv32 = fsm_;
  prof_begin("cif_misc", 0);
  launch_cif_im2col(stream_, H16_, B, T, D, mc_.cif_l_order, mc_.cif_r_order, col16);
  prof_end("cif_misc");
  gemm("gemm_cif", cif_conv_, col16, taps * D, M, conv32, D, nullptr, 0, nullptr, 0, nullptr, 0, true, 0, 1.f);
  prof_begin("cif_misc", 0);
  launch_cif_alpha(stream_, conv32, B, T, D, cif_out_w_, cif_out_b_, mc_.cif_smooth, mc_.cif_noise, mc_.cif_tail,
                   alphas_);
  if (mc_.cif_cumsum) launch_cif_scan_cumsum(stream_, alphas_, B, T1, plan_);
  else launch_cif_scan(stream_, alphas_, B, T1, mc_.cif_threshold, plan_);
  prof_end("cif_misc");
  export_plan(B);
  last_.peak_len = 0;
  last_.cif_peak.clear();
  if (mc_.timestamp_head) start_timestamp_head(B, T);
  // The cross-attention K/V projections of all decoder layers depend on the encoder output only, not on the decoder
  // length: they go out BEFORE the length is read back and keep the device busy during the host round trip (and,
  // in a multi-device group, during the rendez-vous that agrees on the batch-wide length).
  const int nd = (int)dec_.size();
  const int64_t Mp = round_up(M, 128) + 128;
  const int ldkv = std::max(nd, 1) * 2 * D;
  ensure(ws_kv_, (size_t)Mp * ldkv * 2);
  half_t* kv16 = (half_t*)ws_kv_.p;
  if (nd > 0)
    gemm("gemm_dec_kv", dec_kv_all_, H16_, D, M, nullptr, 0, kv16, ldkv, nullptr, 0, nullptr, 0, false, 0, 1.f);
  // the path's only host sync: the decoder length L is data dependent.  The read-back rides a side stream that waits
  // for the CIF scan alone.
  int32_t L = read_back_plan(B);       // (the K / V GEMM above is queued BEHIND the event this waits for: nothing waits for the GEMM)
  if (l_hook_) L = l_hook_(L);                       // shard of a multi-device batch: the batch-wide maximum
  last_.B = B; last_.L = L; last_.V = V; last_.T = T;
  last_.ids.assign((size_t)B * L, 0);
  if (L == 0) { join_ts(); return; }

  const int Md = B * L;
  const int64_t Mdp = round_up(Md, 128) + 128;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_x = carve(Mdp * D * 4), o_xn = carve(Mdp * D * 2);
  const size_t o_h32 = carve(Mdp * F * 4), o_h16 = carve(Mdp * F * 2), o_t = carve(Mdp * D * 4), o_tn = carve(Mdp * D * 4);
  const size_t o_q = carve(Mdp * D * 2), o_ctx = carve(Mdp * D * 2), o_lg = carve((size_t)Mdp * round_up(V, 4) * 4), o_ids = carve((size_t)Md * 8);
  const size_t o_x2 = carve(Mdp * D * 4);
  ensure(ws_dec_, off);
  char* base = (char*)ws_dec_.p;
  float* xd = (float*)(base + o_x); half_t* xdn16 = (half_t*)(base + o_xn);
  float* xd_alt = (float*)(base + o_x2);               // the residual stream ping-pongs when the out-projection rides in front of the next FFN launch
  float* hd32 = (float*)(base + o_h32); half_t* hd16 = (half_t*)(base + o_h16);
  float* t32 = (float*)(base + o_t); float* tn32 = (float*)(base + o_tn);
  half_t* qd16 = (half_t*)(base + o_q); half_t* ctxd16 = (half_t*)(base + o_ctx);
  logits_ = (float*)(base + o_lg); ids_dev_ = (int64_t*)(base + o_ids);
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));

  prof_begin("cif_misc", 0);
  if (mc_.cif_cumsum) launch_cif_gather_cumsum(stream_, H32_, alphas_, B, T, D, T1, plan_, L, xd);
  else launch_cif_gather(stream_, H32_, B, T, D, T1, plan_, L, xd);
  prof_end("cif_misc");
  const bool bias_branch = mc_.seaco && n_hotwords_ > 0;
  float* e0 = nullptr;                                 // SeACo: the bias decoder also starts from the CIF embeds
  float* hid32 = nullptr;
  if (bias_branch) {
    ensure(ws_seaco_in_, (size_t)2 * Mdp * D * 4);
    e0 = (float*)ws_seaco_in_.p;
    hid32 = e0 + (size_t)Mdp * D;
    PF_HIP(hipMemcpyAsync(e0, xd, (size_t)Md * D * 4, hipMemcpyDeviceToDevice, stream_));
  }

  // Decoder layer = 9 launches: norm1 | FFN-up (f16 hidden) | LayerNorm(2048) in place | FFN-down | norm2 | FSMN memory +
  // norm3 (one kernel, dec_fuse_ bit 1) | q | cross attention | out-projection + residual.  The row-complete GEMM can
  // take norm2 (bit 4) and the next block's norm1 (bit 2) as epilogues, but at M = B*L = 5344 it runs on 84 CUs only:
  // measured 47 vs 29.5 + 5.6 us (FFN-down) and 21 vs 16 + 5.6 us (out-projection), so both stay off (PF_DEC_FUSE=7
  // enables them for experiments).
  bool dsmall = Md <= gemm_small_max_rows() && small_ws_ && D == 512 && F % 64 == 0 && F > 576 && !no_small_fuse_;
  if (dsmall) {                                        // the split FFN-down form must apply to (Md, F), else the regular kernels
    GemmSmallArgs t{};
    t.M = Md; t.N = D; t.K = dec_final_w2_.Kpad; t.A = hd16; t.lda = F; t.W = dec_final_w2_.w; t.ldw = dec_final_w2_.Kpad;
    t.ws = small_ws_; t.post_ln_g = dec_after_.g; t.post_ln_b = dec_after_.b;
    dsmall = gemm_small_applicable(t);
  }
  const bool f_fsmn = (dec_fuse_ & 1) != 0, f_out = (dec_fuse_ & 2) != 0 && !dsmall, f_ffn2 = (dec_fuse_ & 4) != 0;
  bool have_n1 = false;                                // xdn16 already holds norm1(xd) of the coming block
  // out-projection chain (k_ffn.hip, OP = 2): layer i's cross-attention out-projection + residual + the next norm1 run in
  // front of the NEXT block's split FFN launch; `pend` = the layer whose context (ctxd16) still waits for its projection
  bool chain = dec_out_chain_ && !dsmall && dec_final_img_ != nullptr;
  for (int i = 0; i < nd && chain; ++i) chain = dec_[i].ffn_img && dec_[i].out_wt;
  const DecLayer* pend = nullptr;
  // ffn_dec: norm1 -> w_1 + ReLU -> LayerNorm(2048) -> w_2 (no bias) [-> LayerNorm `post`]; leaves t32 (unfused) or
  // post(t) in n32 / n16
  bool mid_next = false;                                // the coming ffn_dec call leaves its shares to launch_dec_mid
  auto ffn_dec = [&](const LNp& n1, const Lin& w1, const LNp& fn, const Lin& w2, const half_t* img, const LNp& post, float* n32, half_t* n16) {
    if (img && !dsmall) {
      // norm1 | the whole block in the split form of the fused FFN kernel + its finishing pass (LayerNorm over the hidden
      // applied from row statistics, then `post`): 2 launches for FFN-up | LayerNorm(2048) | FFN-down | LayerNorm
      if (!have_n1 && !pend) {
        prof_begin("layernorm", 0);
        launch_layernorm(stream_, xd, Md, D, n1.g, n1.b, xdn16, D, nullptr, 0);
        prof_end("layernorm");
      }
      have_n1 = false;
      ensure(ws_decffn_, ffn_dec_workspace_bytes(Md));
      FfnDecArgs f{};
      f.A = xdn16; f.lda = D; f.img = img; f.ws = ws_decffn_.p; f.M = Md; f.eps_hidden = 1e-12f;
      f.ln_g = post.g; f.ln_b = post.b; f.eps = 1e-12f; f.n32 = n32; f.ldn32 = D; f.n16 = n16; f.ldn16 = D;
      f.no_finish = mid_next;
      if (pend) {
        f.A = nullptr; f.ctx = ctxd16; f.lda_c = D; f.Wot = pend->out_wt; f.bo = pend->out.bias;
        f.resid = xd; f.ldr = D; f.out_x = xd_alt; f.ldx = D; f.ln1_g = n1.g; f.ln1_b = n1.b; f.eps1 = 1e-12f;
      }
      prof_begin("gemm_dec_ffn", 4.0 * Md * (double)D * F + (pend ? 2.0 * Md * (double)D * D : 0.0));
      launch_ffn_dec(stream_, f);
      prof_end("gemm_dec_ffn");
      if (pend) { std::swap(xd, xd_alt); pend = nullptr; }
      return;
    }
    if (dsmall) {
      // short inputs: norm1 | FFN-up | LayerNorm(2048) in place | FFN-down partials | their sum + the LayerNorm behind it
      prof_begin("layernorm", 0);
      launch_layernorm(stream_, xd, Md, D, n1.g, n1.b, xdn16, D, nullptr, 0);
      prof_end("layernorm");
      dec_ffn_hidden("gemm_dec_ffn1", w1, fn, xdn16, D, Md, hd32, hd16);
      GemmSmallArgs dn{};
      dn.M = Md; dn.A = hd16; dn.lda = F;
      dn.post_ln_g = post.g; dn.post_ln_b = post.b; dn.post_n16 = n16; dn.ldn16 = D; dn.post_n32 = n32; dn.ldn32 = D;
      gemm_small_call("gemm_dec_ffn2", w2, dn, false);
      have_n1 = false;
      return;
    }
    if (!have_n1) {
      prof_begin("layernorm", 0);
      launch_layernorm(stream_, xd, Md, D, n1.g, n1.b, xdn16, D, nullptr, 0);
      prof_end("layernorm");
    }
    have_n1 = false;
    dec_ffn_hidden("gemm_dec_ffn1", w1, fn, xdn16, D, Md, hd32, hd16);
    if (f_ffn2) {
      GemmRcArgs g{};
      g.A = hd16; g.lda = F; g.W = w2.w; g.ldw = w2.Kpad; g.bias = nullptr; g.M = Md; g.K = w2.Kpad;
      g.ln_g = post.g; g.ln_b = post.b; g.eps = 1e-12f;
      g.out_n32 = n32; g.ldn32 = D; g.out_n16 = n16; g.ldn16 = D;
      prof_begin("gemm_dec_ffn2", 2.0 * Md * (double)D * F);
      launch_gemm_rc(stream_, g);
      prof_end("gemm_dec_ffn2");
    } else {
      gemm("gemm_dec_ffn2", w2, hd16, F, Md, t32, D, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f, false);
      prof_begin("layernorm", 0);
      launch_layernorm(stream_, t32, Md, D, post.g, post.b, n16, D, n32, n32 ? D : 0);
      prof_end("layernorm");
    }
  };

  for (int i = 0; i < nd; ++i) {
    const DecLayer& Lr = dec_[i];
    // round 6: the finishing pass of the split FFN form, norm2, the FSMN memory, the residual, norm3 and the q-projection in ONE
    // launch (k_decmid.hip): a decoder layer = split FFN | middle | cross-attention
    const bool mid = dec_mid_ && Lr.ffn_img && Lr.q_wt && !dsmall && D == 512 && mc_.kernel == 11;
    mid_next = mid;
    ffn_dec(Lr.norm1, Lr.w1, Lr.ffn_norm, Lr.w2, Lr.ffn_img, Lr.norm2, tn32, nullptr);
    mid_next = false;
    bool mid_done = false;
    if (mid) {
      DecMidArgs m{};
      m.ws = ws_decffn_.p; m.img = Lr.ffn_img; m.splits = 0; m.eps_hidden = 1e-12f;
      m.n2_g = Lr.norm2.g; m.n2_b = Lr.norm2.b; m.eps2 = 1e-12f;
      m.fsmn_wT = Lr.fsmn_wT; m.k = mc_.kernel; m.token_num = plan_.token_num; m.B = B; m.L = L;
      m.x = xd; m.n3_g = Lr.norm3.g; m.n3_b = Lr.norm3.b;
      m.Wqt = Lr.q_wt; m.bq = Lr.q.bias; m.qscale = qscale; m.q16 = qd16; m.ldq = D;
      prof_begin("dec_mid", 2.0 * Md * (double)D * D);
      mid_done = launch_dec_mid(stream_, m);
      prof_end("dec_mid");
      PF_CHECK(mid_done, PF_ERR_UNSUPPORTED, "decoder: the fused middle launch does not cover this geometry");
    }
    bool fused = mid_done;
    if (f_fsmn && !mid_done) {
      prof_begin("fsmn", 0);
      fused = launch_fsmn_dec_ln(stream_, tn32, Lr.fsmn_wT, plan_.token_num, B, L, D, mc_.kernel, xd, Lr.norm3.g, Lr.norm3.b, xdn16);
      prof_end("fsmn");
    }
    if (!fused) {
      prof_begin("fsmn", 0);
      launch_fsmn_dec(stream_, tn32, Lr.fsmn_wT, plan_.token_num, B, L, D, mc_.kernel, xd);
      prof_end("fsmn");
      prof_begin("layernorm", 0);
      launch_layernorm(stream_, xd, Md, D, Lr.norm3.g, Lr.norm3.b, xdn16, D, nullptr, 0);
      prof_end("layernorm");
    }
    if (!mid_done) gemm("gemm_dec_q", Lr.q, xdn16, D, Md, nullptr, 0, qd16, D, nullptr, 0, nullptr, 0, false, D, qscale);
    AttnArgs a{};
    a.q = qd16; a.q_bstride = (int64_t)L * D; a.q_rstride = D;
    a.k = kv16 + (size_t)i * 2 * D; a.v = kv16 + (size_t)i * 2 * D + D;
    a.k_bstride = a.v_bstride = (int64_t)T * ldkv; a.k_rstride = a.v_rstride = ldkv;
    a.o = ctxd16; a.o_bstride = (int64_t)L * D; a.o_rstride = D;
    a.B = B; a.H = mc_.heads; a.Lq = L; a.Lk = T;
    prof_begin("attn_cross", 4.0 * B * (double)L * T * D);
    launch_attention(stream_, a);
    prof_end("attn_cross");
    if (chain) {
      pend = &Lr;
    } else if (f_out) {
      const LNp& nxt = i + 1 < nd ? dec_[i + 1].norm1 : dec_final_norm1_;
      GemmRcArgs g{};
      g.A = ctxd16; g.lda = D; g.W = Lr.out.w; g.ldw = Lr.out.Kpad; g.bias = Lr.out.bias; g.M = Md; g.K = Lr.out.Kpad;
      g.resid = xd; g.ldr = D; g.out_x = xd; g.ldx = D;
      g.ln_g = nxt.g; g.ln_b = nxt.b; g.eps = 1e-12f; g.out_n16 = xdn16; g.ldn16 = D;
      prof_begin("gemm_dec_out", 2.0 * Md * (double)D * D);
      launch_gemm_rc(stream_, g);
      prof_end("gemm_dec_out");
      have_n1 = true;
    } else {
      gemm("gemm_dec_out", Lr.out, ctxd16, D, Md, xd, D, nullptr, 0, xd, D, nullptr, 0, false, 0, 1.f);
    }
  }
  ffn_dec(dec_final_norm1_, dec_final_w1_, dec_final_ffn_norm_, dec_final_w2_, dec_final_img_, dec_after_, hid32, xdn16);
  logits_ld_ = (int)round_up(V, 4);                 // fp32 rows stay 16-byte aligned for any vocabulary size
  gemm("gemm_vocab", dec_out_, xdn16, D, Md, logits_, logits_ld_, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
  prof_begin("argmax", 0);
  launch_argmax(stream_, logits_, Md, V, logits_ld_, want_logits ? 2 : 1, ids_dev_);
  prof_end("argmax");
  if (bias_branch) seaco_head(B, L, e0, hid32, want_logits);
  join_ts();
  PF_HIP(hipMemcpyAsync(last_.ids.data(), ids_dev_, (size_t)Md * 8, hipMemcpyDeviceToHost, stream_));
}

// Starts the BiCIF timestamp head of the current forward: beside the decoder on its own stream (default), or in line.
// Needs ev_enc_ (encoder output) and ev_scan_ (token_num) recorded on stream_.
void Engine::start_timestamp_head(int B, int T) {
  {
    static const int ts_side = env_int("PF_TS_STREAM", 1);
    if (ts_side && !lstm_steps_) {
      // beside the decoder, on its own stream: everything timestamp_head enqueues (two GEMMs, the persistent recurrence,
      // the peaks and their copy to the host) goes to ts_stream_, which waits for the CIF scan (= the encoder too)
      PF_HIP(hipStreamWaitEvent(ts_stream_, ev_enc_, 0));    // (token_num — the CIF scan — is waited for in front of the peaks only)
      std::swap(stream_, ts_stream_);
      ts_defer_copy_ = true;
      try {
        timestamp_head(B, T);
      } catch (...) {
        std::swap(stream_, ts_stream_);
        ts_defer_copy_ = false;
        throw;
      }
      ts_defer_copy_ = false;
      std::swap(stream_, ts_stream_);
      PF_HIP(hipEventRecord(ev_ts_, ts_stream_));
      ts_pending_ = true;
    } else {
      timestamp_head(B, T);
    }
  }
}

void Engine::join_ts() {
  if (!ts_pending_) return;
  PF_HIP(hipStreamWaitEvent(stream_, ev_ts_, 0));      // the caller's sync of stream_ then covers the timestamp head
  if (ts_copy_floats_)
    PF_HIP(hipMemcpyAsync(last_.cif_peak.data(), us_peak_, ts_copy_floats_ * 4, hipMemcpyDeviceToHost, stream_));
  ts_copy_floats_ = 0;
  ts_pending_ = false;
}

// ------------------------------------------------------------------ streaming seams -------
// The two ONNX graphs of the reference's streaming path (AliParaformerAsr/OnlineRecognizer.cs:49-124 EncoderProj,
// :233-334 DecoderProj; sessions built in OnlineModel.cs:23-31).  Their arithmetic is external (FunASR
// paraformer-online export) and restated in oracle/online.py — parity unpinned like every model graph here:
//   encoder: speech [B,Tc,560] (already x sqrt(512) + position-encoded by the caller, OnlineStream.cs:214-221)
//            -> SAN-M encoder WITHOUT the embed stage -> enc [B,Tc,512]; alphas [B,Tc] = CIF weights (no tail frame)
//   decoder: enc, acoustic_embeds [B,L,512] (+ lengths), 16 FSMN caches [B,512,10] -> log-probs [B,L,V], new caches;
//            FSMN memory = conv over cat(cache, x) (no padding), cache_out = its last 10 columns.
void Engine::online_encoder(const float* speech, int B, int Tc, float* enc_out, float* alphas_out) {
  PF_CHECK(speech && enc_out && alphas_out && B > 0 && Tc > 0, PF_ERR_INVALID_ARG, "online_encoder: bad arguments");
  PF_CHECK(mc_.kind != "sensevoicesmall", PF_ERR_UNSUPPORTED, "online_encoder: paraformer models only");
  PF_HIP(hipSetDevice(device_));
  const int D = mc_.d_model, M = B * Tc;
  const size_t n = (size_t)M * mc_.feat_dim;
  ensure(ws_speech_, n * 4);
  PF_HIP(hipMemcpyAsync(ws_speech_.p, speech, n * 4, hipMemcpyHostToDevice, stream_));
  encoder((const float*)ws_speech_.p, B, Tc, true);
  const int taps = mc_.cif_l_order + mc_.cif_r_order + 1;
  launch_cif_im2col(stream_, H16_, B, Tc, D, mc_.cif_l_order, mc_.cif_r_order, h16_);
  gemm("gemm_cif", cif_conv_, h16_, taps * D, M, fsm_, D, nullptr, 0, nullptr, 0, nullptr, 0, true, 0, 1.f);
  launch_cif_alpha(stream_, fsm_, B, Tc, D, cif_out_w_, cif_out_b_, mc_.cif_smooth, mc_.cif_noise, mc_.cif_tail, alphas_);
  PF_HIP(hipMemcpyAsync(enc_out, H32_, (size_t)M * D * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipMemcpy2DAsync(alphas_out, (size_t)Tc * 4, alphas_, (size_t)(Tc + 1) * 4, (size_t)Tc * 4, B, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::online_decoder(const float* enc, int B, int Tc, const float* embeds, int L, const int32_t* embeds_len,
                            const float* caches_in, float* logits_out, int64_t* ids_out, float* caches_out) {
  PF_CHECK(enc && embeds && embeds_len && caches_in && ids_out && caches_out && B > 0 && Tc > 0 && L > 0, PF_ERR_INVALID_ARG,
           "online_decoder: bad arguments");
  PF_CHECK(mc_.kernel == 11, PF_ERR_UNSUPPORTED, "online_decoder: FSMN kernel 11 (cache of 10 columns) only");
  PF_HIP(hipSetDevice(device_));
  const int D = mc_.d_model, F = mc_.ffn, V = mc_.vocab, nd = (int)dec_.size(), CW = mc_.kernel - 1;
  const int M = B * Tc, Md = B * L;
  const int64_t Mp = round_up(M, 128) + 128, Mdp = round_up(Md, 128) + 128;
  const int ldV = (int)round_up(V, 4);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_e32 = carve((size_t)M * D * 4), o_e16 = carve((size_t)Mp * D * 2), o_kv = carve((size_t)Mp * std::max(nd, 1) * 2 * D * 2);
  const size_t o_x = carve((size_t)Mdp * D * 4), o_xn = carve((size_t)Mdp * D * 2), o_h32 = carve((size_t)Mdp * F * 4);
  const size_t o_h16 = carve((size_t)Mdp * F * 2), o_t = carve((size_t)Mdp * D * 4), o_tn = carve((size_t)Mdp * D * 4);
  const size_t o_q = carve((size_t)Mdp * D * 2), o_ctx = carve((size_t)Mdp * D * 2), o_lg = carve((size_t)Mdp * ldV * 4);
  const size_t o_ids = carve((size_t)Md * 8), o_len = carve((size_t)B * 4);
  const size_t o_ci = carve((size_t)nd * B * D * CW * 4), o_co = carve((size_t)nd * B * D * CW * 4);
  ensure(ws_dec_, off);
  char* base = (char*)ws_dec_.p;
  float* e32 = (float*)(base + o_e32); half_t* e16 = (half_t*)(base + o_e16); half_t* kv16 = (half_t*)(base + o_kv);
  float* xd = (float*)(base + o_x); half_t* xdn16 = (half_t*)(base + o_xn);
  float* hd32 = (float*)(base + o_h32); half_t* hd16 = (half_t*)(base + o_h16);
  float* t32 = (float*)(base + o_t); float* tn32 = (float*)(base + o_tn);
  half_t* qd16 = (half_t*)(base + o_q); half_t* ctxd16 = (half_t*)(base + o_ctx);
  float* lg = (float*)(base + o_lg); int64_t* ids = (int64_t*)(base + o_ids); int32_t* lens = (int32_t*)(base + o_len);
  float* ci = (float*)(base + o_ci); float* co = (float*)(base + o_co);
  PF_HIP(hipMemsetAsync(e16, 0, (size_t)Mp * D * 2, stream_));
  PF_HIP(hipMemcpyAsync(e32, enc, (size_t)M * D * 4, hipMemcpyHostToDevice, stream_));
  launch_f32_to_f16(stream_, e32, M, D, D, e16, D);
  PF_HIP(hipMemcpyAsync(xd, embeds, (size_t)Md * D * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(lens, embeds_len, (size_t)B * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(ci, caches_in, (size_t)nd * B * D * CW * 4, hipMemcpyHostToDevice, stream_));
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));
  const int ldkv = nd * 2 * D;
  if (nd > 0) gemm("gemm_dec_kv", dec_kv_all_, e16, D, M, nullptr, 0, kv16, ldkv, nullptr, 0, nullptr, 0, false, 0, 1.f);
  auto ffn_dec = [&](const LNp& n1, const Lin& w1, const LNp& fn, const Lin& w2) {
    launch_layernorm(stream_, xd, Md, D, n1.g, n1.b, xdn16, D, nullptr, 0);
    dec_ffn_hidden("gemm_dec_ffn1", w1, fn, xdn16, D, Md, hd32, hd16);
    gemm("gemm_dec_ffn2", w2, hd16, F, Md, t32, D, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f, false);
  };
  for (int i = 0; i < nd; ++i) {
    const DecLayer& Lr = dec_[i];
    ffn_dec(Lr.norm1, Lr.w1, Lr.ffn_norm, Lr.w2);
    launch_layernorm(stream_, t32, Md, D, Lr.norm2.g, Lr.norm2.b, nullptr, 0, tn32, D);
    launch_fsmn_dec_stream(stream_, tn32, Lr.fsmn_wT, lens, ci + (size_t)i * B * D * CW, B, L, D, mc_.kernel, xd,
                           co + (size_t)i * B * D * CW);
    launch_layernorm(stream_, xd, Md, D, Lr.norm3.g, Lr.norm3.b, xdn16, D, nullptr, 0);
    gemm("gemm_dec_q", Lr.q, xdn16, D, Md, nullptr, 0, qd16, D, nullptr, 0, nullptr, 0, false, D, qscale);
    AttnArgs a{};
    a.q = qd16; a.q_bstride = (int64_t)L * D; a.q_rstride = D;
    a.k = kv16 + (size_t)i * 2 * D; a.v = kv16 + (size_t)i * 2 * D + D;
    a.k_bstride = a.v_bstride = (int64_t)Tc * ldkv; a.k_rstride = a.v_rstride = ldkv;
    a.o = ctxd16; a.o_bstride = (int64_t)L * D; a.o_rstride = D;
    a.B = B; a.H = mc_.heads; a.Lq = L; a.Lk = Tc;
    launch_attention(stream_, a);
    gemm("gemm_dec_out", Lr.out, ctxd16, D, Md, xd, D, nullptr, 0, xd, D, nullptr, 0, false, 0, 1.f);
  }
  ffn_dec(dec_final_norm1_, dec_final_w1_, dec_final_ffn_norm_, dec_final_w2_);
  launch_layernorm(stream_, t32, Md, D, dec_after_.g, dec_after_.b, xdn16, D, nullptr, 0);
  gemm("gemm_vocab", dec_out_, xdn16, D, Md, lg, ldV, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
  launch_argmax(stream_, lg, Md, V, ldV, logits_out ? 2 : 1, ids);
  if (logits_out) PF_HIP(hipMemcpy2DAsync(logits_out, (size_t)V * 4, lg, (size_t)ldV * 4, (size_t)V * 4, Md, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipMemcpyAsync(ids_out, ids, (size_t)Md * 8, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipMemcpyAsync(caches_out, co, (size_t)nd * B * D * CW * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::set_hotwords(const int32_t* hw, int n) {
  PF_CHECK(n >= 0 && (n == 0 || hw), PF_ERR_INVALID_ARG, "set_hotwords: bad arguments");
  if (n == n_hotwords_ && hotwords_.size() == (size_t)n * 10 && (n == 0 || std::memcmp(hotwords_.data(), hw, (size_t)n * 40) == 0)) return;
  hotwords_.assign(hw, hw + (size_t)n * 10);
  n_hotwords_ = n;
  seaco_hw_valid_ = false;                            // the embedder output / K,V rows of the old list are stale
}

// SeACo bias branch (export_forward of the FunASR SeACo export; reference call site
// OfflineProjOfSeacoParaformer.cs:48-135).  hotwords [N,10] -> Embedding -> LSTM stack -> hw_embed; bias_embed
// row n*10+j = hw_embed[j,n] is the same for every utterance (OfflineProjOfSeacoParaformer.cs:83-111 tiles it
// over B), so its K/V projections are computed once and every (utterance, head) attends the same rows.  The
// bias decoder runs once on 2*B*L rows: [CIF embeds ; ASR decoder hidden].
void Engine::seaco_head(int B, int L, const float* e0, const float* hid32, bool want_logits) {
  const int D = mc_.d_model, V = mc_.vocab, Fs = mc_.seaco_ffn, ns = (int)sdec_.size();
  const int N = n_hotwords_, J = 10, NJ = N * J;
  const int Md = B * L, R = 2 * Md;
  const int64_t NJp = round_up(NJ, 128) + 128, Rp = round_up(R, 128) + 128, Mdp = round_up(Md, 128) + 128;
  const int ldV = (int)round_up(V, 4);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  // the hot-word side (ids, embedder, its K / V rows) lives in its own buffer and is computed once per hot-word LIST, not
  // once per call: it is a function of the weights and the list alone (the reference re-runs model_eb every call,
  // OfflineProjOfSeacoParaformer.cs:83-111, with the same result)
  size_t hoff = 0;
  auto hcarve = [&](size_t bytes) { size_t o = hoff; hoff += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_ids = hcarve((size_t)NJ * 4), o_e32 = hcarve((size_t)NJp * D * 4), o_in16 = hcarve((size_t)NJp * D * 2);
  const size_t o_xg = hcarve((size_t)NJp * 4 * D * 4), o_ho = hcarve((size_t)NJp * D * 4), o_hs = hcarve((size_t)2 * N * D * 2);
  const size_t o_cs = hcarve((size_t)N * D * 4), o_kv = hcarve((size_t)NJp * std::max(ns, 1) * 2 * D * 2);
  if (!ws_seaco_hw_.p || ws_seaco_hw_.bytes < hoff) seaco_hw_valid_ = false;
  ensure(ws_seaco_hw_, hoff);
  const size_t o_x = carve((size_t)Rp * D * 4), o_xn = carve((size_t)Rp * D * 2), o_h32 = carve((size_t)Rp * Fs * 4);
  const size_t o_h16 = carve((size_t)Rp * Fs * 2), o_t = carve((size_t)Rp * D * 4), o_tn = carve((size_t)Rp * D * 4);
  const size_t o_q = carve((size_t)Rp * D * 2), o_ctx = carve((size_t)Rp * D * 2), o_hid = carve((size_t)Rp * D * 4);
  const size_t o_m16 = carve((size_t)Mdp * D * 2), o_dha = carve((size_t)Mdp * ldV * 4), o_did = carve((size_t)Md * 8);
  const size_t o_tn2 = carve((size_t)2 * B * 4);
  ensure(ws_seaco_, off);
  char* base = (char*)ws_seaco_.p;
  char* hbase = (char*)ws_seaco_hw_.p;
  int32_t* ids = (int32_t*)(hbase + o_ids);
  float* e32 = (float*)(hbase + o_e32); half_t* in16 = (half_t*)(hbase + o_in16);
  float* xg = (float*)(hbase + o_xg); float* hout = (float*)(hbase + o_ho);
  half_t* hs = (half_t*)(hbase + o_hs); float* cs = (float*)(hbase + o_cs);
  float* xs = (float*)(base + o_x); half_t* xn16 = (half_t*)(base + o_xn);
  float* h32 = (float*)(base + o_h32); half_t* h16 = (half_t*)(base + o_h16);
  float* t32 = (float*)(base + o_t); float* tn32 = (float*)(base + o_tn);
  half_t* q16 = (half_t*)(base + o_q); half_t* ctx16 = (half_t*)(base + o_ctx);
  float* hid = (float*)(base + o_hid); half_t* m16 = (half_t*)(base + o_m16);
  float* dha = (float*)(base + o_dha); int64_t* dha_ids = (int64_t*)(base + o_did);
  int32_t* tn2 = (int32_t*)(base + o_tn2);
  half_t* kv16 = (half_t*)(hbase + o_kv);
  const int ldkv = ns * 2 * D;

  if (!seaco_hw_valid_) {
  // ---- hotword embedder: Embedding -> LSTM stack (all J outputs kept), batch-major rows n*J + j
  prof_begin("seaco_embed", 0);
  PF_HIP(hipMemcpyAsync(ids, hotwords_.data(), (size_t)NJ * 4, hipMemcpyHostToDevice, stream_));
  launch_embed_gather(stream_, seaco_embed_w_, ids, NJ, D, (int)tensor("seaco.embed.weight").shape[0], e32, in16);
  prof_end("seaco_embed");
  for (size_t l = 0; l < seaco_lstm_.size(); ++l) {
    gemm("gemm_seaco", seaco_lstm_[l].ih, in16, D, NJ, xg, 4 * D, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
    prof_begin("seaco_embed", 0);
    PF_HIP(hipMemsetAsync(hs, 0, (size_t)2 * N * D * 2, stream_));
    PF_HIP(hipMemsetAsync(cs, 0, (size_t)N * D * 4, stream_));
    LstmArgs a{};
    a.whh = seaco_lstm_[l].whh; a.xg = xg; a.hstate = hs; a.cstate = cs; a.hout = hout; a.B = N; a.T3 = J; a.D = D; a.ndir = 1;
    for (int st = 0; st < J; ++st) { a.step = st; launch_lstm_step(stream_, a); }
    launch_f32_to_f16(stream_, hout, NJ, D, D, in16, D);
    prof_end("seaco_embed");
  }
  if (ns > 0 && !int8_mode_)
    gemm("gemm_seaco", seaco_kv_all_, in16, D, NJ, nullptr, 0, kv16, ldkv, nullptr, 0, nullptr, 0, false, 0, 1.f);
  if (ns > 0 && int8_mode_) seaco_kv_int8(hout, in16, NJ, kv16, ldkv);
  seaco_hw_valid_ = true;
  }

  // ---- bias decoder on [CIF embeds ; decoder hidden]
  PF_HIP(hipMemcpyAsync(xs, e0, (size_t)Md * D * 4, hipMemcpyDeviceToDevice, stream_));
  PF_HIP(hipMemcpyAsync(xs + (size_t)Md * D, hid32, (size_t)Md * D * 4, hipMemcpyDeviceToDevice, stream_));
  PF_HIP(hipMemcpyAsync(tn2, plan_.token_num, (size_t)B * 4, hipMemcpyDeviceToDevice, stream_));
  PF_HIP(hipMemcpyAsync(tn2 + B, plan_.token_num, (size_t)B * 4, hipMemcpyDeviceToDevice, stream_));
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));
  if (int8_mode_) {
    // math_mode 2: the bias decoder's Linears are MatMulInteger pairs in model.int8.onnx like the ASR decoder's
    seaco_decoder_int8(B, L, NJ, xs, h32, t32, tn32, q16, ctx16, kv16, ldkv, tn2, hid);
    seaco_decoder_int8(B, L, NJ, xs + (size_t)Md * D, h32, t32, tn32, q16, ctx16, kv16, ldkv, tn2, hid + (size_t)Md * D);
    prof_begin("seaco_merge", 0);
    launch_add_f32(stream_, hid, hid + (size_t)Md * D, (int64_t)Md * D);
    prof_end("seaco_merge");
    qgemm("gemm_seaco", seaco_out_, true, hid, nullptr, D, Md, dha, ldV, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
    prof_begin("seaco_merge", 0);
    launch_argmax(stream_, dha, Md, V, ldV, 2, dha_ids);
    launch_seaco_merge(stream_, dha, ldV, dha_ids, Md, V, mc_.seaco_nobias, want_logits ? 1 : 0, logits_, logits_ld_, ids_dev_);
    prof_end("seaco_merge");
    return;
  }
  auto ffn_dec = [&](const LNp& n1, const Lin& w1, const LNp& fn, const Lin& w2) {
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, xs, R, D, n1.g, n1.b, xn16, D, nullptr, 0);
    prof_end("layernorm");
    dec_ffn_hidden("gemm_seaco", w1, fn, xn16, D, R, h32, h16);
    gemm("gemm_seaco", w2, h16, Fs, R, t32, D, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f, false);
  };
  for (int i = 0; i < ns; ++i) {
    const DecLayer& Lr = sdec_[i];
    ffn_dec(Lr.norm1, Lr.w1, Lr.ffn_norm, Lr.w2);
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, t32, R, D, Lr.norm2.g, Lr.norm2.b, nullptr, 0, tn32, D);
    prof_end("layernorm");
    prof_begin("fsmn", 0);
    launch_fsmn_dec(stream_, tn32, Lr.fsmn_wT, tn2, 2 * B, L, D, mc_.seaco_kernel, xs);
    prof_end("fsmn");
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, xs, R, D, Lr.norm3.g, Lr.norm3.b, xn16, D, nullptr, 0);
    prof_end("layernorm");
    gemm("gemm_seaco", Lr.q, xn16, D, R, nullptr, 0, q16, D, nullptr, 0, nullptr, 0, false, D, qscale);
    AttnArgs a{};
    a.q = q16; a.q_bstride = (int64_t)L * D; a.q_rstride = D;
    a.k = kv16 + (size_t)i * 2 * D; a.v = kv16 + (size_t)i * 2 * D + D;
    a.k_bstride = a.v_bstride = 0; a.k_rstride = a.v_rstride = ldkv;      // one bias_embed for every utterance
    a.o = ctx16; a.o_bstride = (int64_t)L * D; a.o_rstride = D;
    a.B = 2 * B; a.H = mc_.heads; a.Lq = L; a.Lk = NJ;
    prof_begin("attn_seaco", 4.0 * 2 * B * (double)L * NJ * D);
    launch_attention(stream_, a);
    prof_end("attn_seaco");
    gemm("gemm_seaco", Lr.out, ctx16, D, R, xs, D, nullptr, 0, xs, D, nullptr, 0, false, 0, 1.f);
  }
  ffn_dec(seaco_final_norm1_, seaco_final_w1_, seaco_final_ffn_norm_, seaco_final_w2_);
  prof_begin("layernorm", 0);
  launch_layernorm(stream_, t32, R, D, seaco_after_.g, seaco_after_.b, nullptr, 0, hid, D);
  prof_end("layernorm");
  // ---- merged = cif_attended + dec_attended -> hotword_output_layer -> NO-BIAS merge with the ASR rows
  prof_begin("seaco_merge", 0);
  launch_add_to_f16(stream_, hid, hid + (size_t)Md * D, Md, D, m16);
  prof_end("seaco_merge");
  gemm("gemm_seaco", seaco_out_, m16, D, Md, dha, ldV, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
  prof_begin("seaco_merge", 0);
  launch_argmax(stream_, dha, Md, V, ldV, 2, dha_ids);   // the NO-BIAS decision below is taken on the log-probs too
  launch_seaco_merge(stream_, dha, ldV, dha_ids, Md, V, mc_.seaco_nobias, want_logits ? 1 : 0, logits_, logits_ld_, ids_dev_);
  prof_end("seaco_merge");
}

// BiCIF timestamp head (k_bicif.hip): us_cif_peak [B, 3T]; needs token_num (device) only.
void Engine::timestamp_head(int B, int T) {
  const int D = mc_.d_model, up = mc_.upsample;
  const int M = B * T, T3 = up * T;
  const int64_t M3 = (int64_t)B * T3;
  const int64_t Mp = round_up(M, 128) + 128, M3p = round_up(M3, 128) + 128;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_up = carve((size_t)std::max<int64_t>(Mp * up, M3p) * D * 2), o_xg = carve((size_t)M3p * 8 * D * 4);
  const size_t o_ho = carve((size_t)M3 * 2 * D * 4), o_hs = carve((size_t)8 * B * D * 2), o_cs = carve((size_t)2 * B * D * 4);
  const size_t o_al = carve((size_t)M3 * 4), o_pk = carve((size_t)M3 * 4), o_sw = carve(256);
  ensure(ws_ts_, off);
  char* base = (char*)ws_ts_.p;
  half_t* up16 = (half_t*)(base + o_up);
  float* xg = (float*)(base + o_xg);
  float* hout = (float*)(base + o_ho);
  half_t* hs = (half_t*)(base + o_hs);
  float* cs = (float*)(base + o_cs);
  float* al = (float*)(base + o_al);
  us_peak_ = (float*)(base + o_pk);
  // [M, 3D] row-major IS [3M, D]: output row m of the transposed conv holds frames 3t, 3t+1, 3t+2
  gemm("gemm_ts", ts_up_, H16_, D, M, nullptr, 0, up16, up * D, nullptr, 0, nullptr, 0, false, 0, 1.f);
  gemm("gemm_ts", ts_ih_, up16, D, (int)M3, xg, 8 * D, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
  prof_begin("lstm", 2.0 * 2 * M3 * 4.0 * D * D);
  PF_HIP(hipMemsetAsync(hs, 0, (size_t)4 * B * D * 2, stream_));
  PF_HIP(hipMemsetAsync(cs, 0, (size_t)2 * B * D * 4, stream_));
  LstmArgs a{};
  a.whh = ts_whh_; a.xg = xg; a.hstate = hs; a.cstate = cs; a.hout = hout; a.B = B; a.T3 = T3; a.D = D; a.ndir = 2;
  // the recurrence: ONE persistent launch (W_hh resident in registers, h exchanged through write-through stores and
  // an arrival counter, k_bicif.hip) when every workgroup fits on the device at once; otherwise (B > 64, or
  // PF_LSTM_STEPS=1) the 3T dependent launches, captured once per (shape, workspace) into a hipGraph and replayed
  bool persistent = false;
  if (!lstm_steps_) {
    unsigned* sw = (unsigned*)(base + o_sw);
    persistent = launch_lstm_persistent(stream_, a, sw);
    if (persistent) lstm_err_ = sw + 63;
  }
  if (!persistent) {
    if (!lstm_graph_exec_ || lstm_graph_key_.xg != xg || lstm_graph_key_.B != B || lstm_graph_key_.T3 != T3) {
      if (lstm_graph_exec_) { hipGraphExecDestroy(lstm_graph_exec_); lstm_graph_exec_ = nullptr; }
      hipGraph_t g = nullptr;
      PF_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < T3; ++s) { a.step = s; launch_lstm_step(stream_, a); }
      PF_HIP(hipStreamEndCapture(stream_, &g));
      PF_HIP(hipGraphInstantiate(&lstm_graph_exec_, g, nullptr, nullptr, 0));
      hipGraphDestroy(g);
      lstm_graph_key_.xg = xg; lstm_graph_key_.B = B; lstm_graph_key_.T3 = T3;
    }
    PF_HIP(hipGraphLaunch(lstm_graph_exec_, stream_));
  }
  prof_end("lstm");
  if (ts_defer_copy_) PF_HIP(hipStreamWaitEvent(stream_, ev_scan_, 0));   // on the side stream: token_num comes from the CIF scan
  prof_begin("ts_misc", 0);
  launch_us_alpha(stream_, hout, M3, 2 * D, ts_out_w_, ts_out_b_, mc_.cif_smooth2, mc_.cif_noise2, al);
  launch_us_peak(stream_, al, plan_.token_num, B, T3, mc_.cif_threshold - 1e-4f, us_peak_);
  prof_end("ts_misc");
  last_.peak_len = T3;
  last_.cif_peak.resize((size_t)M3);
  // the copy to (pageable) host memory blocks the calling thread until the recurrence has finished: when the head runs
  // beside the decoder it is issued at the join instead (join_ts), after the decoder has been enqueued
  if (ts_defer_copy_) ts_copy_floats_ = (size_t)M3;
  else PF_HIP(hipMemcpyAsync(last_.cif_peak.data(), us_peak_, (size_t)M3 * 4, hipMemcpyDeviceToHost, stream_));
}

void Engine::sensevoice_head(int B, int T, bool want_logits) {
  const int D = mc_.d_model, V = mc_.vocab;
  const int M = B * T;
  const int64_t Mp = round_up(M, 128) + 128;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_lg = carve((size_t)Mp * round_up(V, 4) * 4), o_ids = carve((size_t)M * 8);
  ensure(ws_dec_, off);
  logits_ = (float*)((char*)ws_dec_.p + o_lg);
  ids_dev_ = (int64_t*)((char*)ws_dec_.p + o_ids);
  logits_ld_ = (int)round_up(V, 4);
  gemm("gemm_vocab", ctc_, H16_, D, M, logits_, logits_ld_, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
  prof_begin("argmax", 0);
  launch_argmax(stream_, logits_, M, V, logits_ld_, want_logits ? 2 : 1, ids_dev_);
  prof_end("argmax");
  last_.B = B; last_.L = T; last_.V = V; last_.T = T;
  last_.ids.assign((size_t)M, 0);
  last_.token_num.assign(B, T);
  last_.fire_count.assign(B, T);
  PF_HIP(hipMemcpyAsync(last_.ids.data(), ids_dev_, (size_t)M * 8, hipMemcpyDeviceToHost, stream_));
}

void Engine::forward_device(const float* speech_dev, int B, int T, bool want_logits) {
  PF_HIP(hipSetDevice(device_));
  PF_CHECK(B > 0 && T > 0, PF_ERR_INVALID_ARG, "forward: empty batch");
  last_logits_ = want_logits;
  if (fp32_mode_) { forward_fp32(speech_dev, B, T, want_logits); return; }
  if (int8_mode_) { forward_int8(speech_dev, B, T, want_logits); return; }
  encoder(speech_dev, B, T);
  if (mc_.kind == "sensevoicesmall") sensevoice_head(B, T, want_logits);
  else predictor_and_decoder(B, T, want_logits);
  // algorithmic FLOPs
  const double D = mc_.d_model, F = mc_.ffn, Td = T, Ld = last_.L, K = mc_.kernel, V = mc_.vocab;
  double enc = 2 * Td * mc_.feat_dim * 3 * D + (mc_.enc_layers - 1 + mc_.tp_layers) * 2 * Td * D * 3 * D +
               (mc_.enc_layers + mc_.tp_layers) * (4 * Td * Td * D + 2 * Td * D * D + 4 * Td * D * F + 2 * Td * D * K);
  double fl = enc;
  if (mc_.kind == "sensevoicesmall") fl += 2 * Td * D * V;
  else {
    fl += 2 * Td * D * D * 3 + 2 * Td * D;
    if (mc_.timestamp_head) fl += 2 * Td * D * 3 * D + 2 * (2 * 3 * Td * D * 8 * D);
    fl += mc_.dec_layers * (4 * Ld * D * F + 2 * Ld * D * K + 2 * Ld * D * D + 4 * Td * D * D + 4 * Ld * Td * D + 2 * Ld * D * D) +
          4 * Ld * D * F + 2 * Ld * D * V;
  }
  last_flops_ = fl * B;
}


// Per-thread result slots live in thread-local storage keyed by the engine's uid: a slot dies with its thread
// (no growth under thread-pool churn; a new thread that happens to re-use an OS thread id starts empty), is
// released by the pf_fetch that delivers token_ids, and slots of engines that no longer exist are purged here.
static std::mutex g_live_mu;
static std::set<uint64_t> g_live;
static uint64_t g_next_uid = 1;
static thread_local std::map<uint64_t, HostBatchOut> t_slots;

uint64_t Engine::register_uid() {
  std::lock_guard<std::mutex> lk(g_live_mu);
  const uint64_t id = g_next_uid++;
  g_live.insert(id);
  return id;
}
void Engine::unregister_uid(uint64_t id) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  g_live.erase(id);
}

void Engine::publish_thread_result() {
  PF_HIP(hipStreamSynchronize(stream_));
  check_async_errors();
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (auto it = t_slots.begin(); it != t_slots.end();)
      it = g_live.count(it->first) ? std::next(it) : t_slots.erase(it);
  }
  HostBatchOut& sl = t_slots[uid_];
  sl = last_;
  sl.has_logits = last_logits_;
  sl.logits.clear();
  const int64_t need = (int64_t)last_.B * last_.L * last_.V;
  if (last_logits_ && need > 0) {
    sl.logits.resize((size_t)need);
    PF_HIP(hipMemcpy2D(sl.logits.data(), (size_t)last_.V * 4, logits_, (size_t)logits_ld_ * 4, (size_t)last_.V * 4,
                       (size_t)last_.B * last_.L, hipMemcpyDeviceToHost));
  }
}

void Engine::drop_thread_result() { t_slots.erase(uid_); }

void Engine::copy_logits(HostBatchOut& r) {
  const int64_t need = (int64_t)last_.B * last_.L * last_.V;
  r.logits.assign((size_t)std::max<int64_t>(need, 0), 0.f);
  r.has_logits = last_logits_;
  if (!last_logits_ || need <= 0) return;
  PF_HIP(hipMemcpy2D(r.logits.data(), (size_t)last_.V * 4, logits_, (size_t)logits_ld_ * 4, (size_t)last_.V * 4,
                     (size_t)last_.B * last_.L, hipMemcpyDeviceToHost));
}

void Engine::fetch_ids_device(int64_t* ids_dev, int l_cap, int32_t* L_out) {
  PF_HIP(hipSetDevice(device_));
  const int B = last_.B, L = last_.L;
  PF_CHECK(ids_dev && l_cap > 0, PF_ERR_INVALID_ARG, "fetch_ids_device: null buffer");
  PF_CHECK(l_cap >= L, PF_ERR_CAPACITY, "fetch_ids_device: capacity " + std::to_string(l_cap) + " < L = " + std::to_string(L));
  if (L_out) *L_out = L;
  if (B > 0) {
    PF_HIP(hipMemsetAsync(ids_dev, 0xFF, (size_t)B * l_cap * 8, stream_));
    if (L > 0)
      PF_HIP(hipMemcpy2DAsync(ids_dev, (size_t)l_cap * 8, ids_dev_, (size_t)L * 8, (size_t)L * 8, B, hipMemcpyDeviceToDevice, stream_));
  }
  sync();
}

void Engine::fetch(pf_batch_out* out) {
  PF_CHECK(out, PF_ERR_INVALID_ARG, "fetch: null out");
  PF_HIP(hipStreamSynchronize(stream_));
  check_async_errors();
  auto it = t_slots.find(uid_);
  const bool slot = it != t_slots.end();
  const HostBatchOut& r = slot ? it->second : last_;
  const int B = r.B, L = r.L, V = r.V;
  out->L = L; out->V = V; out->cif_peak_len = r.peak_len;
  if (out->cif_peak && out->cif_peak_cap > 0 && r.peak_len > 0) {
    PF_CHECK(out->cif_peak_cap >= (int64_t)r.cif_peak.size(), PF_ERR_CAPACITY, "cif_peak capacity < B*3T");
    std::memcpy(out->cif_peak, r.cif_peak.data(), r.cif_peak.size() * 4);
  }
  if (out->token_ids) {
    PF_CHECK(out->l_cap >= L, PF_ERR_CAPACITY, "token_ids capacity " + std::to_string(out->l_cap) + " < L = " + std::to_string(L));
    for (int b = 0; b < B; ++b) {
      std::memcpy(out->token_ids + (size_t)b * out->l_cap, r.ids.data() + (size_t)b * L, (size_t)L * 8);
    }
  }
  if (out->token_num) std::memcpy(out->token_num, r.token_num.data(), (size_t)B * 4);
  if (out->logits && out->logits_cap > 0) {
    PF_CHECK(slot ? r.has_logits : last_logits_, PF_ERR_INVALID_ARG, "logits were not requested for the last forward");
    const int64_t need = (int64_t)B * L * V;
    PF_CHECK(out->logits_cap >= need, PF_ERR_CAPACITY, "logits capacity < B*L*V = " + std::to_string(need));
    if (need > 0) {
      if (slot) std::memcpy(out->logits, r.logits.data(), (size_t)need * 4);
      else
        PF_HIP(hipMemcpy2D(out->logits, (size_t)V * 4, logits_, (size_t)logits_ld_ * 4, (size_t)V * 4, (size_t)B * L,
                           hipMemcpyDeviceToHost));
    }
  }
  // the call that receives the ids completes the learn-L-then-fetch protocol: release the slot (a B*L*V host
  // copy of the log-probs may hang off it)
  if (slot && out->token_ids) t_slots.erase(it);
}

}  // namespace pf
