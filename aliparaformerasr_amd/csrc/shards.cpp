// shards.cpp — see shards.h.
#include "shards.h"

#include <algorithm>
#include <climits>
#include <cstring>

namespace pf {

// ------------------------------------------------------------------ MaxBarrier ------------
int MaxBarrier::arrive_and_max(int v) {
  std::unique_lock<std::mutex> lk(mu_);
  if (broken_) throw Error(PF_ERR_RECOGNITION, "another device of the group failed");
  const int gen = gen_;
  cur_ = std::max(cur_, v);
  if (++count_ == n_) {
    result_ = cur_;
    cur_ = 0;
    count_ = 0;
    ++gen_;
    cv_.notify_all();
    return result_;
  }
  cv_.wait(lk, [&] { return gen_ != gen || broken_; });
  if (gen_ == gen) throw Error(PF_ERR_RECOGNITION, "another device of the group failed");
  return result_;
}
void MaxBarrier::abort() {
  std::lock_guard<std::mutex> lk(mu_);
  broken_ = true;
  cv_.notify_all();
}
void MaxBarrier::reset() {
  std::lock_guard<std::mutex> lk(mu_);
  broken_ = false;
  count_ = 0;
  cur_ = 0;
}

// ------------------------------------------------------------------ ShardRunner ------------
ShardRunner::ShardRunner(int G) : lbar_(G), agree_(G), ready_(G) {
  // every Worker exists before the first thread starts: worker_loop indexes workers_, which must not grow under it
  // (a thread started inside the emplace_back loop read the vector while it re-allocated — a crash once in a few runs)
  workers_.reserve((size_t)G);
  for (int i = 0; i < G; ++i) workers_.emplace_back(new Worker());
  for (int i = 0; i < G; ++i) workers_[(size_t)i]->th = std::thread([this, i] { worker_loop(i); });
}

ShardRunner::~ShardRunner() {
  for (auto& w : workers_) {
    { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
  }
}

void ShardRunner::worker_loop(int i) {
  Worker& w = *workers_[(size_t)i];
  for (;;) {
    std::function<void()> job;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      w.cv.wait(lk, [&] { return w.has_job || w.stop; });
      if (w.stop) return;
      job = std::move(w.job);
      w.has_job = false;
    }
    int code = 0;
    std::string err;
    try {
      job();
    } catch (const Error& ex) {
      code = ex.code; err = ex.what();
    } catch (const std::exception& ex) {
      code = PF_ERR_DEVICE; err = ex.what();
    }
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.code = code; w.error = err; w.done = true;
    }
    w.cv.notify_all();
  }
}

void ShardRunner::run_on_all(const std::function<void(int)>& fn) {
  for (size_t i = 0; i < workers_.size(); ++i) {
    Worker& w = *workers_[i];
    std::lock_guard<std::mutex> lk(w.mu);
    const int gi = (int)i;
    w.job = [fn, gi] { fn(gi); };
    w.has_job = true; w.done = false; w.code = 0; w.error.clear();
    w.cv.notify_all();
  }
  int code = 0;
  std::string err;
  for (auto& wp : workers_) {
    std::unique_lock<std::mutex> lk(wp->mu);
    wp->cv.wait(lk, [&] { return wp->done; });
    // report the root cause, not the "another device failed" echoes
    if (wp->code && (!code || (code == PF_ERR_RECOGNITION && err.find("another device") != std::string::npos))) {
      code = wp->code; err = wp->error;
    }
  }
  if (code) throw Error(code, err);
}

void ShardRunner::recognize(ShardBackend& be, int B, int Tg, bool has_cif, int V, bool want_logits, HostBatchOut& merged) {
  const int G = size();
  merged = HostBatchOut();
  if (B == 0) return;
  const ShardPlan plan(G, B);
  std::vector<HostBatchOut> part((size_t)G);
  const bool collective = be.has_collective();
  lbar_.reset(); agree_.reset(); ready_.reset();
  int L_all = 0;
  run_on_all([&](int g) {
    const int lo = plan.lo(g), hi = plan.hi(g), Bg = hi - lo;
    HostBatchOut& r = part[(size_t)g];
    // ---- 1. the forward; CIF models meet once inside it (decoder length).  A shard that throws releases the
    //         waiters of THAT rendez-vous only; every worker, failed or not, goes on to rendez-vous 2.
    bool failed = false;
    int code = 0;
    std::string err;
    try {
      if (Bg == 0) {
        if (has_cif) lbar_.arrive_and_max(0);
      } else {
        be.run(g, lo, hi, Tg, want_logits, [this](int L) { return lbar_.arrive_and_max(L); }, r);
      }
    } catch (const Error& ex) {
      failed = true; code = ex.code; err = ex.what();
    } catch (const std::exception& ex) {
      failed = true; code = PF_ERR_DEVICE; err = ex.what();
    }
    if (failed) lbar_.abort();
    // ---- 2. one decoder length for everybody (an empty shard has none of its own; SenseVoice has no rendez-vous 1)
    //         and one verdict: INT_MAX = somebody failed, nobody enters the collective
    const int agreed = agree_.arrive_and_max(failed ? INT_MAX : (Bg ? r.L : 0));
    if (failed) throw Error(code, err);
    if (agreed == INT_MAX) throw Error(PF_ERR_RECOGNITION, "another device of the group failed");
    PF_CHECK(Bg == 0 || r.L == agreed, PF_ERR_DEVICE, "pf_group: shards disagree on the decoder length");
    if (g == 0) L_all = agreed;
    if (!collective) return;
    // ---- 3. gather of the hypotheses: the part that can fail first, then all shards or none
    const GatherLayout lay(plan.per, agreed);
    try {
      be.prepare_gather(g, Bg, agreed, lay, G);
    } catch (const Error& ex) {
      failed = true; code = ex.code; err = ex.what();
    } catch (const std::exception& ex) {
      failed = true; code = PF_ERR_DEVICE; err = ex.what();
    }
    const int bad = ready_.arrive_and_max(failed ? 1 : 0);
    if (failed) throw Error(code, err);
    if (bad) throw Error(PF_ERR_RECOGNITION, "another device of the group failed");
    be.gather(g, lay, G);
  });
  // ---- merge in the caller's order
  const int L = L_all;
  int P = 0;
  for (auto& r : part) P = std::max(P, r.peak_len);
  merged.B = B; merged.L = L; merged.V = V; merged.T = Tg; merged.peak_len = P;
  merged.ids.assign((size_t)B * L, 0);
  merged.token_num.assign((size_t)B, 0);
  merged.fire_count.assign((size_t)B, 0);
  if (P > 0) merged.cif_peak.assign((size_t)B * P, 0.f);
  if (want_logits) merged.logits.assign((size_t)B * L * V, 0.f);
  merged.has_logits = want_logits;
  std::vector<char> gathered;
  const GatherLayout lay(plan.per, L);
  if (collective && L > 0) be.read_gathered(gathered, lay.block_bytes * (size_t)G);   // ONE read-back
  for (int g = 0; g < G; ++g) {
    const HostBatchOut& r = part[(size_t)g];
    const int lo = plan.lo(g), Bg = plan.count(g);
    for (int b = 0; b < Bg; ++b) {
      if (!gathered.empty()) {
        const char* blk = gathered.data() + lay.block_bytes * (size_t)g;
        std::memcpy(&merged.ids[(size_t)(lo + b) * L], blk + (size_t)b * L * 8, (size_t)L * 8);
        if (has_cif) std::memcpy(&merged.token_num[(size_t)(lo + b)], blk + lay.ids_bytes + (size_t)b * 4, 4);
        else merged.token_num[(size_t)(lo + b)] = r.token_num[(size_t)b];
      } else {
        if (L > 0) std::memcpy(&merged.ids[(size_t)(lo + b) * L], &r.ids[(size_t)b * r.L], (size_t)L * 8);
        merged.token_num[(size_t)(lo + b)] = r.token_num[(size_t)b];
      }
      merged.fire_count[(size_t)(lo + b)] = r.fire_count[(size_t)b];
      if (P > 0 && r.peak_len == P) std::memcpy(&merged.cif_peak[(size_t)(lo + b) * P], &r.cif_peak[(size_t)b * P], (size_t)P * 4);
      if (want_logits && L > 0)
        std::memcpy(&merged.logits[(size_t)(lo + b) * L * V], &r.logits[(size_t)b * L * V], (size_t)L * V * 4);
    }
  }
}

}  // namespace pf
