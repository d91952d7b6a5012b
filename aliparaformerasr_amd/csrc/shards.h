// shards.h — the device-independent half of pf_group (group.h): how a batch is cut into utterance shards, how the
// G shard workers rendez-vous, and how the per-shard results are merged back into the caller's order.
//
// What shards is the utterance list handed to GetResults (AliParaformerAsr/OfflineRecognizer.cs:110-116; the batch is
// dim 0 of every tensor, OfflineProjOfParaformer.cs:49).  Contiguous blocks of ceil(B / G); every
// shard pads to the BATCH maximum length (PadHelper.cs:25) and decodes the BATCH maximum token count, so the merged id
// matrix equals the single-device one position by position.
//
// The work of a shard is behind `ShardBackend`: group.cpp implements it with one Engine per GPU and an RCCL
// all-gather; `pf_host_group_sim` (c_api.cpp) implements it with arithmetic stand-ins so that the plan, the three
// rendez-vous, the gather layout, the failure paths and the merge run in the CPU test-suite (tests/test_shards_cpu.py).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace pf {

struct HostBatchOut { // results of a forward, host side
  int B = 0, L = 0, V = 0, T = 0;
  std::vector<int64_t> ids;        // [B, L]
  std::vector<int32_t> token_num;  // [B]
  std::vector<int32_t> fire_count; // [B]
  std::vector<float> cif_peak;     // [B, peak_len] us_cif_peak (timestamp models), else empty
  int peak_len = 0;
  std::vector<float> logits;       // [B, L, V] host copy (per-thread result slots only)
  bool has_logits = false;
};

// rendezvous of the G worker threads with a max-reduction; abort() releases every waiter with an Error
class MaxBarrier {
 public:
  explicit MaxBarrier(int n) : n_(n) {}
  int arrive_and_max(int v);                 // throws Error(PF_ERR_RECOGNITION) when aborted
  void abort();
  void reset();

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, count_ = 0, gen_ = 0, cur_ = 0, result_ = 0;
  bool broken_ = false;
};

// contiguous blocks of ceil(B / G); trailing shards may be empty ((G-1) * per >= B, e.g. G = 8, B = 9)
struct ShardPlan {
  int G = 1, B = 0, per = 0;
  ShardPlan(int G_, int B_) : G(G_), B(B_), per(G_ > 0 ? (B_ + G_ - 1) / G_ : 0) {}
  int lo(int g) const { return std::min(g * per, B); }
  int hi(int g) const { return std::min(lo(g) + per, B); }
  int count(int g) const { return hi(g) - lo(g); }
};

// the fixed-shape block every shard contributes to the all-gather: [per][L] int64 ids, then [per] int32 token_num,
// rounded up to 256 bytes; rows a shard does not own are 0xFF (id -1, token_num -1).  A function of (per, L) only —
// every rank of a collective must pass the same count.
struct GatherLayout {
  size_t ids_bytes = 0, block_bytes = 0;
  GatherLayout() = default;
  GatherLayout(int per, int L) {
    ids_bytes = (size_t)per * (size_t)std::max(L, 1) * 8;
    block_bytes = (size_t)round_up((int64_t)(ids_bytes + (size_t)per * 4), 256);
  }
};

class ShardBackend {
 public:
  virtual ~ShardBackend() = default;
  // Recognise utterances [lo, hi) of the batch, padded to Tg LFR frames.  For CIF models the backend calls
  // l_sync(own decoder length) exactly once and decodes the length it returns (the batch-wide maximum); an EMPTY
  // shard (lo == hi) is not run at all — the runner arrives at the rendez-vous on its behalf.
  virtual void run(int g, int lo, int hi, int Tg, bool want_logits, const std::function<int(int)>& l_sync, HostBatchOut& out) = 0;
  virtual bool has_collective() const = 0;
  // collective, in two halves: everything that can fail (allocations, packing) happens in prepare_gather; gather is
  // entered by all shards or by none (the runner agrees on that in between)
  virtual void prepare_gather(int g, int count, int L, const GatherLayout& lay, int G) = 0;
  virtual void gather(int g, const GatherLayout& lay, int G) = 0;
  virtual void read_gathered(std::vector<char>& host, size_t bytes) = 0;   // shard 0's receive buffer
};

class ShardRunner {
 public:
  explicit ShardRunner(int G);
  ~ShardRunner();
  int size() const { return (int)workers_.size(); }
  // fn(g) on the worker thread of every shard; rethrows the root-cause error of the first failing shard
  void run_on_all(const std::function<void(int)>& fn);
  // One batch: B utterances over the shards of `be`; Tg = batch-wide LFR frame count; has_cif = the model has a
  // data-dependent decoder length (paraformer / SeACo), else every shard reports the same L on its own.
  void recognize(ShardBackend& be, int B, int Tg, bool has_cif, int V, bool want_logits, HostBatchOut& merged);

 private:
  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, stop = false;
    std::string error;
    int code = 0;
  };
  void worker_loop(int i);
  std::vector<std::unique_ptr<Worker>> workers_;
  MaxBarrier lbar_, agree_, ready_;          // decoder length | shared L + failure flag | prepare_gather outcome
};

}  // namespace pf
