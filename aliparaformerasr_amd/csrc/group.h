// group.h — several engines, one per listed GPU, inside ONE process ("single process, one host
// thread + stream per GPU, RCCL broadcast / gather").  The reference has no multi-device code at all
// (AliParaformerAsr/OfflineRecognizer.cs:23 builds one ORT session); what shards is the utterance list handed to
// GetResults (OfflineRecognizer.cs:110-116): the batch is dim 0 of every tensor (OfflineProjOfParaformer.cs:49).
//
//   * weights: read once, uploaded to devices[0], sent to the other GPUs with ncclBroadcast over xGMI
//   * recognise: contiguous shards of ceil(B/G) utterances (original order restored on return); every shard is
//     padded to the GLOBAL maximum length (PadHelper.cs:25 pads to the batch maximum) and decodes the GLOBAL
//     maximum token count, so the id matrix equals the single-device one position by position
//   * hypotheses: fixed-shape [per, L] ids + [per] token_num gathered to every GPU with ncclAllGather, read back once
//
// librccl is dlopen'ed at pf_group_create (no link-time dependency).  When the device list repeats a device (tests:
// two engines on one GPU) or names a single device, no communicator is needed and copies are plain HIP copies.
#pragma once
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "engine.h"

namespace pf {

// rendezvous of the G worker threads with a max-reduction (the decoder length L); abort() releases every waiter
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

struct Rccl;   // dlopen'ed entry points

class Group {
 public:
  Group(const pf_engine_config& cfg, const int32_t* devices, int n);
  ~Group();
  int size() const { return (int)eng_.size(); }
  std::shared_ptr<Engine> engine(int i) { return eng_[(size_t)i]; }
  bool uses_rccl() const { return comms_ready_; }
  // B utterances in, merged result (original order) kept for fetch()
  void recognize(const float* const* samples, const int64_t* n, int B, const int32_t* hotwords, int n_hotwords,
                 bool want_logits);
  void fetch(pf_batch_out* out);
  std::mutex& mutex() { return mu_; }

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
  void run_on_all(const std::function<void(int)>& fn);
  void release();

  std::mutex mu_;
  std::vector<int> devs_;
  std::vector<std::shared_ptr<Engine>> eng_;
  std::vector<void*> images_;                // device weight image per engine (shared when devices repeat)
  std::vector<bool> image_owned_;
  std::vector<std::unique_ptr<Worker>> workers_;
  std::unique_ptr<MaxBarrier> lbar_;
  std::unique_ptr<Rccl> rccl_;
  std::vector<void*> comms_;                 // ncclComm_t per engine
  bool comms_ready_ = false;
  std::vector<void*> gsend_, grecv_;         // gather buffers per engine
  std::vector<size_t> gsend_bytes_, grecv_bytes_;
  HostBatchOut merged_;
  bool merged_logits_ = false;
};

}  // namespace pf
