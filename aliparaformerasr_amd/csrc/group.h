// group.h — several engines, one per listed GPU, inside ONE process ("single process, one host
// thread + stream per GPU, RCCL broadcast / gather").  The reference has no multi-device code at all
// (AliParaformerAsr/OfflineRecognizer.cs:23 builds one ORT session); what shards is the utterance list handed to
// GetResults (OfflineRecognizer.cs:110-116): the batch is dim 0 of every tensor (OfflineProjOfParaformer.cs:49).
//
//   * weights: read once, uploaded to devices[0], sent to the other GPUs with ncclBroadcast over xGMI
//   * recognise: the shard plan, the rendez-vous and the merge live in shards.h (device independent, CPU-tested);
//     this class is the ShardBackend that runs a shard on its Engine and gathers the hypotheses
//   * hypotheses: fixed-shape [per, L] ids + [per] token_num gathered to every GPU with ncclAllGather, read back once
//
// librccl is dlopen'ed at pf_group_create (no link-time dependency).  When the device list repeats a device (tests:
// two engines on one GPU) no communicator is possible and the merge uses each engine's own host copy.
#pragma once
#include <memory>
#include <mutex>
#include <vector>

#include "engine.h"
#include "shards.h"

namespace pf {

struct Rccl;   // dlopen'ed entry points

class Group : private ShardBackend {
 public:
  Group(const pf_engine_config& cfg, const int32_t* devices, int n);
  ~Group() override;
  int size() const { return (int)eng_.size(); }
  std::shared_ptr<Engine> engine(int i) { return eng_[(size_t)i]; }
  bool uses_rccl() const { return comms_ready_; }
  // B utterances in, merged result (original order) kept for fetch()
  void recognize(const float* const* samples, const int64_t* n, int B, const int32_t* hotwords, int n_hotwords,
                 bool want_logits);
  void fetch(pf_batch_out* out);
  std::mutex& mutex() { return mu_; }

 private:
  // ShardBackend
  void run(int g, int lo, int hi, int Tg, bool want_logits, const std::function<int(int)>& l_sync, HostBatchOut& out) override;
  bool has_collective() const override { return comms_ready_; }
  void prepare_gather(int g, int count, int L, const GatherLayout& lay, int G) override;
  void gather(int g, const GatherLayout& lay, int G) override;
  void read_gathered(std::vector<char>& host, size_t bytes) override;
  void release();

  std::mutex mu_;
  std::vector<int> devs_;
  std::vector<std::shared_ptr<Engine>> eng_;
  std::vector<void*> images_;                // device weight image per engine (shared when devices repeat)
  std::vector<bool> image_owned_;
  std::unique_ptr<ShardRunner> runner_;
  std::unique_ptr<Rccl> rccl_;
  std::vector<void*> comms_;                 // ncclComm_t per engine
  bool comms_ready_ = false;
  std::vector<void*> gsend_, grecv_;         // gather buffers per engine
  std::vector<size_t> gsend_bytes_, grecv_bytes_;
  std::vector<void*> gsrc_;                  // per engine: ids + token_num of the shard, copied under the engine lock in run()
  std::vector<size_t> gsrc_bytes_;
  std::vector<int> gsrc_rows_, gsrc_L_;
  // the call in flight (valid inside recognize())
  const float* const* cur_samples_ = nullptr;
  const int64_t* cur_n_ = nullptr;
  const int32_t* cur_hotwords_ = nullptr;
  int cur_n_hotwords_ = 0;
  bool cur_has_cif_ = true;
  HostBatchOut merged_;
  bool merged_logits_ = false;
};

}  // namespace pf
