// shards_sim.h — arithmetic stand-ins for the devices of a pf_group: the ShardBackend behind `pf_host_group_sim`
// (c_api.cpp; tests/test_shards_cpu.py) and behind the sanitizer harness tests/native/shards_sanitize.cpp.
// Utterance u "decodes" to ids[u][l] = u * 100000 + l; the stand-in collective refuses what RCCL would hang on.
#pragma once
#include <cstring>

#include "shards.h"

namespace pf {
struct SimBackend : ShardBackend {
  int G, B, has_cif, fixed_L, collective, fail_shard, fail_stage;
  const int32_t* fire;
  std::vector<std::vector<char>> send;     // one block per shard
  std::vector<size_t> sizes;               // the count each shard would hand to ncclAllGather
  std::vector<char> recv0;                 // shard 0's receive buffer
  MaxBarrier bar;
  std::vector<HostBatchOut*> outs;
  SimBackend(int G_) : G(G_), send((size_t)G_), sizes((size_t)G_, 0), bar(G_), outs((size_t)G_, nullptr) {}
  void run(int g, int lo, int hi, int /*Tg*/, bool /*want_logits*/, const std::function<int(int)>& l_sync, HostBatchOut& r) override {
    if (g == fail_shard && fail_stage == 0) throw Error(PF_ERR_DEVICE, "simulated failure before the decoder-length rendez-vous");
    const int Bg = hi - lo;
    int own = 0;
    for (int b = lo; b < hi; ++b) own = std::max(own, fire[b]);
    const int L = has_cif ? l_sync(own) : fixed_L;
    if (g == fail_shard && fail_stage == 1) throw Error(PF_ERR_DEVICE, "simulated failure after the decoder-length rendez-vous");
    r = HostBatchOut();
    r.B = Bg; r.L = L; r.V = 1;
    r.ids.resize((size_t)Bg * L);
    r.token_num.resize((size_t)Bg); r.fire_count.resize((size_t)Bg);
    for (int b = 0; b < Bg; ++b) {
      for (int l = 0; l < L; ++l) r.ids[(size_t)b * L + l] = (int64_t)(lo + b) * 100000 + l;
      r.token_num[(size_t)b] = has_cif ? fire[lo + b] : L;
      r.fire_count[(size_t)b] = fire[lo + b];
    }
    outs[(size_t)g] = &r;
  }
  bool has_collective() const override { return collective != 0; }
  void prepare_gather(int g, int count, int L, const GatherLayout& lay, int) override {
    if (g == fail_shard && fail_stage == 2) throw Error(PF_ERR_DEVICE, "simulated failure while preparing the gather");
    sizes[(size_t)g] = lay.block_bytes;
    send[(size_t)g].assign(lay.block_bytes, (char)0xFF);
    if (count > 0 && L > 0) {
      const HostBatchOut& r = *outs[(size_t)g];
      std::memcpy(send[(size_t)g].data(), r.ids.data(), (size_t)count * L * 8);
      if (has_cif) std::memcpy(send[(size_t)g].data() + lay.ids_bytes, r.token_num.data(), (size_t)count * 4);
    }
  }
  void gather(int g, const GatherLayout& lay, int) override {
    bar.arrive_and_max(0);                                       // every block is packed
    if (g == 0) {
      for (int i = 0; i < G; ++i)                                // what RCCL requires of its callers
        PF_CHECK(sizes[(size_t)i] == lay.block_bytes, PF_ERR_DEVICE, "all-gather entered with different counts on different ranks");
      recv0.resize(lay.block_bytes * (size_t)G);
      for (int i = 0; i < G; ++i) std::memcpy(recv0.data() + lay.block_bytes * (size_t)i, send[(size_t)i].data(), lay.block_bytes);
    }
    bar.arrive_and_max(0);
  }
  void read_gathered(std::vector<char>& host, size_t bytes) override {
    PF_CHECK(bytes == recv0.size(), PF_ERR_DEVICE, "merge expects a different gather size than the collective produced");
    host = recv0;
  }
};
}  // namespace pf
