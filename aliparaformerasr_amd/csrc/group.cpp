// group.cpp — see group.h.
#include "group.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <set>

#include <rccl/rccl.h>   // types and enums only: every entry point is resolved with dlsym

#include "hostutil.h"

namespace pf {

// ------------------------------------------------------------------ RCCL (dlopen) ---------
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  ~Rccl() { if (lib) dlclose(lib); }
  static std::unique_ptr<Rccl> open() {
    std::unique_ptr<Rccl> r(new Rccl());
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
      r->lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (r->lib) break;
    }
    if (!r->lib) throw Error(PF_ERR_UNSUPPORTED, std::string("pf_group: librccl.so not found (") + dlerror() + ")");
#define PF_SYM(field, name)                                                                        \
  r->field = reinterpret_cast<decltype(r->field)>(dlsym(r->lib, name));                            \
  if (!r->field) throw Error(PF_ERR_UNSUPPORTED, std::string("pf_group: librccl lacks ") + name)
    PF_SYM(CommInitAll, "ncclCommInitAll");
    PF_SYM(CommDestroy, "ncclCommDestroy");
    PF_SYM(Broadcast, "ncclBroadcast");
    PF_SYM(AllGather, "ncclAllGather");
    PF_SYM(GroupStart, "ncclGroupStart");
    PF_SYM(GroupEnd, "ncclGroupEnd");
    PF_SYM(GetErrorString, "ncclGetErrorString");
#undef PF_SYM
    return r;
  }
  void check(ncclResult_t rc, const char* what) const {
    if (rc != ncclSuccess) throw Error(PF_ERR_DEVICE, std::string("rccl: ") + what + ": " + GetErrorString(rc));
  }
};

// ------------------------------------------------------------------ Group ------------------
Group::Group(const pf_engine_config& cfg, const int32_t* devices, int n) {
  PF_CHECK(devices && n > 0 && n <= 64, PF_ERR_INVALID_ARG, "pf_group: device list must name 1..64 devices");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    throw Error(PF_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
  for (int i = 0; i < n; ++i) {
    PF_CHECK(devices[i] >= 0 && devices[i] < ndev, PF_ERR_DEVICE, "pf_group: device ordinal out of range");
    devs_.push_back(devices[i]);
  }
  const std::set<int> uniq(devs_.begin(), devs_.end());
  const bool distinct = (int)uniq.size() == n;

  // ---- the weight image: host -> devices[0] once, then GPU -> GPU
  std::vector<char> file;
  const char* host = nullptr;
  int64_t nbytes = 0;
  if (cfg.weights_path && cfg.weights_path[0]) {
    read_binary_file(cfg.weights_path, file);
    host = file.data();
    nbytes = (int64_t)file.size();
  } else if (cfg.weights_host) {
    host = (const char*)cfg.weights_host;
    nbytes = cfg.weights_bytes;
  }
  PF_CHECK(host || cfg.weights_device, PF_ERR_INVALID_ARG, "pf_group: no weight source");
  if (!host) nbytes = cfg.weights_bytes;
  PF_CHECK(nbytes >= 16, PF_ERR_FORMAT, "weights: image too small");
  try {
    images_.assign((size_t)n, nullptr);
    image_owned_.assign((size_t)n, false);
    for (int i = 0; i < n; ++i) {
      int first = i;                                  // engines on one device share one image
      for (int j = 0; j < i; ++j)
        if (devs_[(size_t)j] == devs_[(size_t)i]) { first = j; break; }
      if (first != i) { images_[(size_t)i] = images_[(size_t)first]; continue; }
      if (i == 0 && !host) { images_[0] = const_cast<void*>(cfg.weights_device); continue; }
      PF_HIP(hipSetDevice(devs_[(size_t)i]));
      PF_HIP(hipMalloc(&images_[(size_t)i], (size_t)nbytes));
      image_owned_[(size_t)i] = true;
    }
    PF_HIP(hipSetDevice(devs_[0]));
    if (host) PF_HIP(hipMemcpy(images_[0], host, (size_t)nbytes, hipMemcpyHostToDevice));
    if (distinct) {
      // one communicator over the device list (n == 1: a one-rank communicator — the same calls, no traffic)
      rccl_ = Rccl::open();
      std::vector<ncclComm_t> cs((size_t)n);
      rccl_->check(rccl_->CommInitAll(cs.data(), n, devs_.data()), "ncclCommInitAll");
      for (ncclComm_t c : cs) comms_.push_back((void*)c);
      comms_ready_ = true;
      if (n > 1) {
        std::vector<hipStream_t> ss((size_t)n);
        for (int i = 0; i < n; ++i) {
          PF_HIP(hipSetDevice(devs_[(size_t)i]));
          PF_HIP(hipStreamCreateWithFlags(&ss[(size_t)i], hipStreamNonBlocking));
        }
        rccl_->check(rccl_->GroupStart(), "ncclGroupStart");
        for (int i = 0; i < n; ++i)
          rccl_->check(rccl_->Broadcast(images_[0], images_[(size_t)i], (size_t)nbytes, ncclUint8, 0, (ncclComm_t)comms_[(size_t)i],
                                        ss[(size_t)i]),
                       "ncclBroadcast");
        rccl_->check(rccl_->GroupEnd(), "ncclGroupEnd");
        for (int i = 0; i < n; ++i) {
          PF_HIP(hipSetDevice(devs_[(size_t)i]));
          PF_HIP(hipStreamSynchronize(ss[(size_t)i]));
          PF_HIP(hipStreamDestroy(ss[(size_t)i]));
        }
      }
    } else {
      for (int i = 1; i < n; ++i)
        if (image_owned_[(size_t)i])
          PF_HIP(hipMemcpyPeer(images_[(size_t)i], devs_[(size_t)i], images_[0], devs_[0], (size_t)nbytes));
    }
    // ---- one engine per entry, each adopting its device image in place
    for (int i = 0; i < n; ++i) {
      pf_engine_config ec = cfg;
      ec.device = devs_[(size_t)i];
      ec.weights_path = nullptr;
      ec.weights_host = nullptr;
      ec.weights_device = images_[(size_t)i];
      ec.weights_bytes = nbytes;
      eng_.push_back(std::make_shared<Engine>(ec));
    }
    gsend_.assign((size_t)n, nullptr); grecv_.assign((size_t)n, nullptr);
    gsend_bytes_.assign((size_t)n, 0); grecv_bytes_.assign((size_t)n, 0);
    gsrc_.assign((size_t)n, nullptr); gsrc_bytes_.assign((size_t)n, 0); gsrc_rows_.assign((size_t)n, 0); gsrc_L_.assign((size_t)n, 0);
    runner_.reset(new ShardRunner(n));
    runner_->run_on_all([this](int g) { PF_HIP(hipSetDevice(devs_[(size_t)g])); });   // each worker thread binds its device once
  } catch (...) {
    release();
    throw;
  }
}

Group::~Group() { release(); }

void Group::release() {
  runner_.reset();                                    // joins the worker threads
  eng_.clear();                                       // engines first: they borrow the images
  for (size_t i = 0; i < gsend_.size(); ++i) {
    hipSetDevice(devs_[i]);
    if (gsend_[i]) hipFree(gsend_[i]);
    if (grecv_[i]) hipFree(grecv_[i]);
    if (i < gsrc_.size() && gsrc_[i]) hipFree(gsrc_[i]);
  }
  gsend_.clear(); grecv_.clear(); gsrc_.clear();
  if (rccl_)
    for (void* c : comms_) rccl_->CommDestroy((ncclComm_t)c);
  comms_.clear();
  comms_ready_ = false;
  for (size_t i = 0; i < images_.size(); ++i)
    if (image_owned_[i] && images_[i]) { hipSetDevice(devs_[i]); hipFree(images_[i]); }
  images_.clear();
  rccl_.reset();
}

void Group::recognize(const float* const* samples, const int64_t* n, int B, const int32_t* hotwords, int n_hotwords,
                      bool want_logits) {
  PF_CHECK(B >= 0 && (B == 0 || (samples && n)), PF_ERR_INVALID_ARG, "pf_group_recognize: bad arguments");
  merged_ = HostBatchOut();
  merged_logits_ = want_logits;
  if (B == 0) return;
  int Tg = 0;                                         // the reference pads to the BATCH maximum (PadHelper.cs:25)
  for (int b = 0; b < B; ++b) {
    if (!samples[b] && n[b] > 0) throw Error(PF_ERR_NULL_SAMPLES, "source");
    Tg = std::max(Tg, eng_[0]->num_lfr_frames(n[b]));
  }
  cur_samples_ = samples; cur_n_ = n; cur_hotwords_ = hotwords; cur_n_hotwords_ = hotwords ? n_hotwords : 0;
  cur_has_cif_ = eng_[0]->model().kind != "sensevoicesmall";
  runner_->recognize(*this, B, Tg, cur_has_cif_, eng_[0]->model().vocab, want_logits, merged_);
  cur_samples_ = nullptr; cur_n_ = nullptr; cur_hotwords_ = nullptr;
}

// ---- ShardBackend: one shard on its engine (worker thread g, device bound at construction) -------------------
void Group::run(int g, int lo, int hi, int Tg, bool want_logits, const std::function<int(int)>& l_sync, HostBatchOut& r) {
  Engine& e = *eng_[(size_t)g];
  const int Bg = hi - lo;
  std::lock_guard<std::mutex> lk(e.mutex());
  if (e.model().seaco) e.set_hotwords(cur_hotwords_, cur_n_hotwords_);
  if (cur_has_cif_) e.set_l_hook(l_sync);
  try {
    e.stage_audio(cur_samples_ + lo, cur_n_ + lo, Bg, Tg);
    e.run_staged(want_logits);
  } catch (...) {
    e.set_l_hook(nullptr);
    throw;
  }
  e.set_l_hook(nullptr);
  e.sync();
  r = e.last_result();
  if (want_logits && (int64_t)r.B * r.L * r.V > 0) e.copy_logits(r);
  // The gather runs after two rendez-vous, without this engine's mutex: another thread holding a pf_group_engine view may
  // run a forward in between, re-carve (or free) the decoder arena and overwrite the ids.  What the gather sends is
  // therefore copied NOW, under the lock, into a buffer only the group's threads touch.
  if (comms_ready_) {
    const size_t idb = (size_t)r.B * r.L * 8, tnb = (size_t)r.B * 4;
    void*& p = gsrc_[(size_t)g];
    size_t& have = gsrc_bytes_[(size_t)g];
    if (have < idb + tnb + 16) {
      if (p) { PF_HIP(hipFree(p)); p = nullptr; have = 0; }
      PF_HIP(hipMalloc(&p, idb + tnb + 16));
      have = idb + tnb + 16;
    }
    if (idb) PF_HIP(hipMemcpyAsync(p, e.ids_device(), idb, hipMemcpyDeviceToDevice, e.stream()));
    if (cur_has_cif_ && tnb) PF_HIP(hipMemcpyAsync((char*)p + idb, e.token_num_device(), tnb, hipMemcpyDeviceToDevice, e.stream()));
    PF_HIP(hipStreamSynchronize(e.stream()));
    gsrc_rows_[(size_t)g] = r.B; gsrc_L_[(size_t)g] = r.L;
  }
}

// fixed-shape [per, L] int64 ids + [per] int32 token_num per shard; `lay` is the same on every rank (it is a function
// of the shard size and the AGREED decoder length, never of a rank's own result)
void Group::prepare_gather(int g, int count, int L, const GatherLayout& lay, int G) {
  Engine& e = *eng_[(size_t)g];
  auto grow = [&](void*& p, size_t& have, size_t want) {
    if (have >= want) return;
    if (p) { PF_HIP(hipStreamSynchronize(e.stream())); PF_HIP(hipFree(p)); p = nullptr; have = 0; }
    PF_HIP(hipMalloc(&p, want));
    have = want;
  };
  grow(gsend_[(size_t)g], gsend_bytes_[(size_t)g], lay.block_bytes);
  grow(grecv_[(size_t)g], grecv_bytes_[(size_t)g], lay.block_bytes * (size_t)G);
  char* sb = (char*)gsend_[(size_t)g];
  PF_HIP(hipMemsetAsync(sb, 0xFF, lay.block_bytes, e.stream()));           // absent rows: id -1, token_num -1
  if (count > 0 && L > 0) {
    PF_CHECK(gsrc_[(size_t)g] && gsrc_rows_[(size_t)g] == count && gsrc_L_[(size_t)g] == L, PF_ERR_DEVICE,
             "group: the shard's ids were not kept for the gather (rows / length differ from the agreed layout)");
    const char* src = (const char*)gsrc_[(size_t)g];
    PF_HIP(hipMemcpyAsync(sb, src, (size_t)count * L * 8, hipMemcpyDeviceToDevice, e.stream()));
    if (cur_has_cif_)
      PF_HIP(hipMemcpyAsync(sb + lay.ids_bytes, src + (size_t)count * L * 8, (size_t)count * 4, hipMemcpyDeviceToDevice, e.stream()));
  }
}

void Group::gather(int g, const GatherLayout& lay, int /*G*/) {
  Engine& e = *eng_[(size_t)g];
  rccl_->check(rccl_->AllGather(gsend_[(size_t)g], grecv_[(size_t)g], lay.block_bytes, ncclUint8, (ncclComm_t)comms_[(size_t)g],
                                e.stream()),
               "ncclAllGather");
  PF_HIP(hipStreamSynchronize(e.stream()));
}

void Group::read_gathered(std::vector<char>& host, size_t bytes) {
  host.resize(bytes);
  PF_HIP(hipSetDevice(devs_[0]));
  PF_HIP(hipMemcpy(host.data(), grecv_[0], bytes, hipMemcpyDeviceToHost));
}

void Group::fetch(pf_batch_out* out) {
  PF_CHECK(out, PF_ERR_INVALID_ARG, "fetch: null out");
  const HostBatchOut& r = merged_;
  const int B = r.B, L = r.L, V = r.V;
  out->L = L; out->V = V; out->cif_peak_len = r.peak_len;
  if (out->cif_peak && out->cif_peak_cap > 0 && r.peak_len > 0) {
    PF_CHECK(out->cif_peak_cap >= (int64_t)r.cif_peak.size(), PF_ERR_CAPACITY, "cif_peak capacity < B*3T");
    std::memcpy(out->cif_peak, r.cif_peak.data(), r.cif_peak.size() * 4);
  }
  if (out->token_ids) {
    PF_CHECK(out->l_cap >= L, PF_ERR_CAPACITY, "token_ids capacity " + std::to_string(out->l_cap) + " < L = " + std::to_string(L));
    for (int b = 0; b < B; ++b)
      std::memcpy(out->token_ids + (size_t)b * out->l_cap, r.ids.data() + (size_t)b * L, (size_t)L * 8);
  }
  if (out->token_num && B > 0) std::memcpy(out->token_num, r.token_num.data(), (size_t)B * 4);
  if (out->logits && out->logits_cap > 0) {
    PF_CHECK(merged_logits_, PF_ERR_INVALID_ARG, "logits were not requested for the last pf_group_recognize");
    const int64_t need = (int64_t)B * L * V;
    PF_CHECK(out->logits_cap >= need, PF_ERR_CAPACITY, "logits capacity < B*L*V = " + std::to_string(need));
    if (need > 0) std::memcpy(out->logits, r.logits.data(), (size_t)need * 4);
  }
}

}  // namespace pf
