// Sanitizer harness for the shard runner of pf_group (csrc/shards.cpp): start / run / fail / stop, thousands of times,
// under -fsanitize=thread or -fsanitize=address (tests/test_shards_cpu.py builds and runs it).  Found in round 3: worker
// threads started inside the loop that was still growing `workers_` (a crash once in ~5 runs of the CPU suite).
//   usage: shards_sanitize <iterations>
#include <cstdio>
#include <cstdlib>

#include "shards_sim.h"

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 200;
  int fire[64];
  for (int i = 0; i < 64; ++i) fire[i] = 1 + (i * 7) % 23;
  long ok = 0, failed = 0;
  for (int it = 0; it < iters; ++it) {
    const int G = 2 + it % 7, B = 1 + (it * 5) % 64;
    for (int stage = -1; stage < 3; ++stage) {                       // -1: no fault
      pf::SimBackend be(G);
      be.B = B; be.has_cif = it % 3 != 0; be.fixed_L = 17; be.collective = it % 2; be.fire = fire;
      be.fail_shard = stage < 0 ? -1 : it % G; be.fail_stage = stage;
      pf::ShardRunner runner(G);
      pf::HostBatchOut m;
      try {
        runner.recognize(be, B, 1, be.has_cif != 0, 1, false, m);
        ++ok;
      } catch (const pf::Error&) {
        ++failed;
      }
      be.fail_shard = -1;                                             // the same runner again: barriers re-armed, no fault
      try {
        runner.recognize(be, B, 1, be.has_cif != 0, 1, false, m);
      } catch (const pf::Error& ex) {
        std::printf("a call after a failed one failed: %s\n", ex.what());
        return 2;
      }
      if (m.B != B) { std::printf("bad merge\n"); return 2; }
    }
  }
  std::printf("ok %ld failed %ld\n", ok, failed);
  return 0;
}
